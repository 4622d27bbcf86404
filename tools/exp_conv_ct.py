"""Trunk conv layers at S sequences per launch with the channel-group size chosen by the library or forced
(POF_CONV_CT=4|2|1): the launch-quantisation experiment behind the 64-channel rule of pof_conv3_bn_lrelu.
    python tools/exp_conv_ct.py 3600"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
S = int(sys.argv[1])
layers = [(S, 64, 64, 56, 0), (S, 64, 128, 56, 1), (S, 128, 128, 28, 0), (S, 128, 256, 28, 1), (S, 256, 256, 14, 0), (S, 256, 512, 14, 1), (S, 512, 256, 7, 0), (S, 256, 128, 7, 0), (S, 512, 256, 14, 0), (S, 256, 128, 28, 0)]
tot = 0
for (S_, Ci, Co, L, pool) in layers:
    x = torch.randn((S_, Ci, L), device="cuda"); wt = torch.randn((3, Ci, Co), device="cuda") * 0.05
    sc = torch.ones(Co, device="cuda"); sh = torch.zeros(Co, device="cuda")
    out = torch.empty((S_, Co, L // 2 if pool else L), device="cuda")
    for _ in range(3): ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    print("S=%d %3d->%3d L=%2d: %.3f ms %.1f TF" % (S_, Ci, Co, L, ms, 2.0 * S_ * L * Co * Ci * 3 / ms / 1e9))
print("sum %.3f" % tot)
