import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import ops, synth
tab = ops.phi_table()
B, T, P = 2048, 5, 56
sb = synth.make_batch(seed=3, B=B, T=T)
scans = torch.from_numpy(sb.scans).cuda()
out = torch.empty((B, 450, T, P), dtype=torch.float32, device="cuda")
kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=P, padding_val=29.99, area_mode=True)
for _ in range(4):
    ops.cutout(scans, tab, out=out, **kw)
torch.cuda.synchronize()
kw["area_mode"] = False
for _ in range(4):
    ops.cutout(scans, tab, out=out, **kw)
torch.cuda.synchronize()
