"""Experiment (round 3): library-kernel choices inside the box-head step -- the fc2 GEMM (256 x 512 -> 256) that
hipBLASLt runs as ONE 256 x 256 tile, and the point-wise convolution kernel at the PointNet's shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planar_optical_flow_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, K, N) in ((256, 1024, 512), (256, 512, 256), (256, 256, 3)):
    h = torch.randn(B, K, device=dev)
    W = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    print("linear %d x %d -> %d" % (B, K, N))
    for name, f in {"F.linear": lambda: F.linear(h, W, b),
                    "mm + add": lambda: torch.mm(h, W.t()) + b,
                    "(W h^T)^T + b": lambda: torch.mm(W, h.t()).t() + b,
                    "two halves of N": lambda: torch.cat((F.linear(h, W[:N // 2], b[:N // 2]), F.linear(h, W[N // 2:], b[N // 2:])), 1),
                    "bmm": lambda: torch.bmm(h.unsqueeze(0), W.t().unsqueeze(0)).squeeze(0) + b,
                    "einsum": lambda: torch.einsum("bk,nk->bn", h, W) + b}.items():
        if N < 8 and "halves" in name:
            continue
        print("   %-24s %7.1f us" % (name, timeit(f)))

for (S, Ci, Co, L) in ((256, 128, 1024, 64), (256, 1024, 128, 64), (256, 64, 128, 64), (256, 128, 64, 64), (256, 64, 64, 64),
                       (256, 3, 64, 64)):
    x = torch.randn(S, Ci, L, device=dev)
    wt = torch.randn(1, Ci, Co, device=dev)
    one, zero = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    out = torch.empty(S, Co, L, device=dev)
    us = timeit(lambda: ops.conv1d_bn_lrelu(x, wt, one, zero, negative_slope=1.0, out=out))
    print("conv k=1 %4d -> %4d  S=%d L=%d   %7.1f us   %5.1f TFLOP/s" % (Ci, Co, S, L, us, 2.0 * S * L * Ci * Co / us / 1e6))
