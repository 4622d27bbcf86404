"""DR-SPAAM forward (BASELINE config 3 shape): cutout -> SpatialDROW on the device.  Under rocprofv3
--kernel-trace --stats this shows how the time splits between the MIOpen trunks and the HIP kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import ops, synth
from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "hip"      # hip | torch | torch-find
if mode == "torch-find":
    torch.backends.cudnn.benchmark = True               # MIOpen find mode instead of the immediate fallback
torch.manual_seed(3)
m = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
if mode == "hip":
    m.fuse_for_inference()
sb = synth.make_batch(seed=3, B=B, T=5)
scans = torch.from_numpy(sb.scans).cuda()
tab = ops.phi_table()
kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=56, padding_val=29.99,
          area_mode=True)
def step():
    x = ops.cutout(scans, tab, **kw)
    with torch.no_grad():
        return m(x)
if mode.startswith("train"):
    step = lambda: None
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("DR-SPAAM forward B=%d [%s]: %.2f ms/step  %.0f scans/s" % (B, mode, dt * 1e3, B / dt), flush=True)
if mode in ("train", "train-miopen", "train-modules", "train-libconv"):
    # one optimisation-style step: forward in training mode (BatchNorm batch statistics) + backward
    m.train()
    m.fused_train_tail = mode != "train-modules"   # train-modules: the framework's own BatchNorm / LeakyReLU / pool
    m.hip_train_conv = mode != "train-libconv"       # train-libconv: MIOpen convolutions + the fused tail
    x = ops.cutout(scans, tab, **kw)
    def tstep():
        for p in m.parameters(): p.grad = None
        pc, pr, ff = m(x)
        (pc.sum() + pr.sum() + ff.sum()).backward()
    for _ in range(2): tstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): tstep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("DR-SPAAM train step (fwd+bwd) B=%d [%s]: %.1f ms  peak mem %.1f GB" % (B, mode, dt * 1e3, torch.cuda.max_memory_allocated() / 1e9), flush=True)
