"""One trunk layer of pof_conv3_bn_lrelu, a few launches -- the target of counter passes (tools/pmc_kernel.sh).
usage: pmc_conv_layer.py S Ci Co L pool"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from planar_optical_flow_amd import ops  # noqa: E402

S, Ci, Co, L, pool = (int(v) for v in sys.argv[1:6])
x = torch.randn((S, Ci, L), device="cuda")
wt = torch.randn((3, Ci, Co), device="cuda") * 0.05
sc, sh = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
out = torch.empty((S, Co, L // 2 if pool else L), device="cuda")
for _ in range(4):
    ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
torch.cuda.synchronize()
