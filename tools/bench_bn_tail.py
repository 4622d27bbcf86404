"""Fused BatchNorm(train) + LeakyReLU + max-pool tail at the training shapes: forward and backward time, algorithmic
GB/s.  POF_LIB_PATH selects a library variant (tuning experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = 0.0
for (S, C, L, pool, G) in [(18000, 64, 56, False, 5), (18000, 128, 56, True, 5), (18000, 128, 28, False, 5),
                           (18000, 256, 28, True, 5), (3600, 256, 14, False, 1), (3600, 512, 14, True, 1),
                           (3600, 256, 7, False, 1)]:
    y = torch.randn(S, C, L, device="cuda")
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") - 0.5
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    z, mu, istd = ops.bn_lrelu_pool_forward(y, gam, bet, rm, rv, pool=pool, groups=G)
    dz = torch.randn_like(z)
    tf = timed(lambda: ops.bn_lrelu_pool_forward(y, gam, bet, rm, rv, pool=pool, groups=G))
    tb = timed(lambda: ops.bn_lrelu_pool_backward(y, dz, gam, bet, mu, istd, pool=pool, bias_grad=True, groups=G))
    el = S * C * L
    bf, bb = el * (4 + 4 + (2 if pool else 4)), el * (4 + (2 if pool else 4)) * 2 + el * 4
    tot += tf + tb
    print("[%5d x %3d x %2d pool=%d groups=%d] fwd %.3f ms %5.0f GB/s   bwd %.3f ms %5.0f GB/s" %
          (S, C, L, pool, G, tf, bf / tf / 1e6, tb, bb / tb / 1e6), flush=True)
print("sum %.3f ms" % tot)
