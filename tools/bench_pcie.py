"""Headline step with the inputs coming from pinned host memory (what a host-buffer boundary would cost):
H2D copies of the scans / odometry / detections of batch i+1 on a copy stream overlap the kernel of batch i."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import ops, synth

B, N, steps = 4096, 450, 300
sb = synth.make_batch(seed=2, B=B, T=2, N=N)
offs, rphi, cls = sb.det_csr()
dev = torch.device("cuda")
tab = ops.phi_table(device=dev)
host = {"scans": torch.from_numpy(sb.scans).pin_memory(), "o0": torch.from_numpy(sb.odom0).pin_memory(),
        "o1": torch.from_numpy(sb.odom1).pin_memory(), "offs": torch.from_numpy(offs.astype(np.int32)).pin_memory(),
        "rphi": torch.from_numpy(rphi).pin_memory(), "cls": torch.from_numpy(np.full(len(rphi), 2, np.uint8)).pin_memory()}
in_bytes = sum(t.numel() * t.element_size() for t in host.values())
slots = []
for _ in range(2):
    d = {k: torch.empty_like(v, device=dev) for k, v in host.items()}
    outs = {"flow": torch.empty((B, N, 2), dtype=torch.float32, device=dev),
            "target_cls": torch.empty((B, N), dtype=torch.int64, device=dev),
            "target_reg": torch.empty((B, N, 2), dtype=torch.float32, device=dev),
            "exclude_mask": torch.empty((B, N), dtype=torch.float32, device=dev)}
    ws = torch.empty(ops.scan_preprocess_workspace_bytes(B, len(rphi)), dtype=torch.uint8, device=dev)
    slots.append((d, outs, ws, torch.cuda.Event(), torch.cuda.Event()))
copy_stream = torch.cuda.Stream()
want = ("flow", "target_cls", "target_reg", "exclude_mask")

def upload(i):
    d, _, _, ready, free = slots[i % 2]
    with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(free)                 # the kernel that last read this slot has finished
        for k, v in host.items():
            d[k].copy_(v, non_blocking=True)
        ready.record(copy_stream)

def compute(i):
    d, outs, ws, ready, free = slots[i % 2]
    cur = torch.cuda.current_stream()
    cur.wait_event(ready)
    det = ops.DetCSR(d["offs"], d["rphi"], d["cls"])
    ops.scan_preprocess(d["scans"], tab, d["o0"], d["o1"], det, want=want, out=outs, workspace=ws)
    free.record(cur)

for s in slots:
    s[4].record(torch.cuda.current_stream())
upload(0)
for i in range(20):
    upload(i + 1); compute(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20, 20 + steps):
    upload(i + 1); compute(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("host-fed headline step: %.1f us / step, %.2f M scans/s, H2D %.1f GB/s (%d input bytes per step)"
      % (dt * 1e6, B / dt / 1e6, in_bytes / dt / 1e9, in_bytes))
