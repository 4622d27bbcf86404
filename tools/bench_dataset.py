"""Wall time of DROWDeviceDataset.get_batch (window gather + odometry association + preprocess + cutout)
for a 4096-sample batch drawn from a synthetic 16 000-scan data set held in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import synth
from planar_optical_flow_amd.scan_store import DROWDeviceDataset

rng = np.random.default_rng(0)
seqs = []
for q in range(8):
    S = 2000
    sb = synth.make_batch(seed=500 + q, B=S, T=1)
    od = np.cumsum(rng.uniform(-0.02, 0.02, (S, 3)), axis=0).astype(np.float32)
    mk = lambda n: [[float(rng.uniform(1, 8)), float(rng.uniform(-1.6, 1.6))] for _ in range(n)]
    seqs.append(dict(scans=sb.scans[:, 0], scans_ns=np.arange(S), scans_t=(np.arange(S) * 0.08).astype(np.float32),
                     odoms_t=(np.arange(S) * 0.08).astype(np.float32), odoms=od, dets_ns=np.arange(S),
                     dets_wc=[mk(rng.integers(0, 3)) for _ in range(S)], dets_wa=[mk(rng.integers(0, 2)) for _ in range(S)],
                     dets_wp=[mk(rng.integers(0, 4)) for _ in range(S)]))
kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=56, padding_val=29.99, area_mode=True)
for cut in (None, kw):
    ds = DROWDeviceDataset(seqs, num_scans=5, cutout_kwargs=cut)
    idx = rng.integers(0, len(ds), 4096)
    for _ in range(3): ds.get_batch(idx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ds.get_batch(idx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("get_batch B=4096 (%s cutout): %.2f ms  %.2f M samples/s  (data set: %d samples)" % ("with" if cut else "no", dt * 1e3, 4096 / dt / 1e6, len(ds)), flush=True)
