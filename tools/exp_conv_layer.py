"""One trunk layer for counter passes: python tools/exp_conv_layer.py S Ci Co L pool [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
S, Ci, Co, L, pool = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
x = torch.randn((S, Ci, L), device="cuda")
wt = torch.randn((3, Ci, Co), device="cuda") * 0.05
sc = torch.ones(Co, device="cuda"); sh = torch.zeros(Co, device="cuda")
out = torch.empty((S, Co, L // 2 if pool else L), device="cuda")
for _ in range(iters):
    ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
torch.cuda.synchronize()
