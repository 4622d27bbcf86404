"""The dense layers of the box head at batch 256 (fc1 1024 -> 512, fc2 512 -> 256, fc3 256 -> 3): the BLAS library's forward
in several formulations, sizes around the single-tile hole (an output of at most 256 x 256: 118 us), and pof_linear_bias."""
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


B, K, N = 256, 512, 256
h = torch.randn(B, K, device=dev)
W = torch.randn(N, K, device=dev)
b = torch.randn(N, device=dev)
ref = F.linear(h.double(), W.double(), b.double())
print("backend", torch.backends.cuda.preferred_blas_library())
cands = {
    "F.linear": lambda: F.linear(h, W, b),
    "two halves of K": lambda: torch.addmm(F.linear(h[:, :K // 2], W[:, :K // 2], b), h[:, K // 2:], W[:, K // 2:].t()),
    "K halves, contiguous": None,
    "conv kernel, L = 1": lambda: ops.conv1d_bn_lrelu(h.view(B, K, 1), W.t().contiguous().view(1, K, N), torch.ones(N, device=dev), b,
                                                      negative_slope=1.0).view(B, N),
    "matmul in 4 row blocks": lambda: torch.cat([F.linear(h[i:i + 64], W, b) for i in range(0, B, 64)], 0),
    "N = 512 (W twice)": lambda: F.linear(h, torch.cat((W, W), 0))[:, :N] + b,
}
h1, h2, W1, W2 = h[:, :K // 2].contiguous(), h[:, K // 2:].contiguous(), W[:, :K // 2].contiguous(), W[:, K // 2:].contiguous()
cands["K halves, contiguous"] = lambda: torch.addmm(F.linear(h1, W1, b), h2, W2.t())
for name, f in cands.items():
    out = f()
    print("   %-26s %7.1f us   max err %.2e" % (name, timeit(f), float((out.double() - ref).abs().max())))
for (bb, kk, nn) in ((256, 1024, 512), (256, 512, 256), (256, 256, 3)):
    hh, ww, bv = torch.randn(bb, kk, device=dev), torch.randn(nn, kk, device=dev), torch.randn(nn, device=dev)
    out = torch.empty(bb, nn, device=dev)
    print("   pof_linear_bias %4d x %4d -> %4d  %7.1f us (eager launch rate; ~5 us as a graph node)"
          % (bb, kk, nn, timeit(lambda: ops.linear_bias(hh, ww, bv, out=out))))
dy = torch.randn(B, N, device=dev)
print("   dgrad dy W               %7.1f us" % timeit(lambda: torch.mm(dy, W)))
print("   wgrad dy^T h             %7.1f us" % timeit(lambda: torch.mm(dy.t(), h)))
for lib in ("hipblas", "hipblaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        print("backend", torch.backends.cuda.preferred_blas_library())
        for (bb, kk, nn) in ((256, 1024, 512), (256, 512, 256), (256, 256, 3)):
            hh, ww, bv = torch.randn(bb, kk, device=dev), torch.randn(nn, kk, device=dev), torch.randn(nn, device=dev)
            print("   F.linear %4d x %4d -> %4d  %7.1f us" % (bb, kk, nn, timeit(lambda: F.linear(hh, ww, bv))))
    except Exception as e:  # noqa: BLE001
        print(lib, "failed:", e)
torch.backends.cuda.preferred_blas_library("hipblaslt")
Wt = W.t().contiguous()
ht = h.t().contiguous()
print("layouts (256 x 512 -> 256)")
print("   NN  h @ Wt                %7.1f us" % timeit(lambda: torch.mm(h, Wt)))
print("   TN  ht^T @ Wt             %7.1f us" % timeit(lambda: torch.mm(ht.t(), Wt)))
print("   NT  h @ W^T               %7.1f us" % timeit(lambda: torch.mm(h, W.t())))
print("   TT  ht^T @ W^T            %7.1f us" % timeit(lambda: torch.mm(ht.t(), W.t())))
for n in (192, 224, 248, 256, 264, 288, 320, 384):
    Wn = torch.randn(n, K, device=dev)
    print("   F.linear N = %3d          %7.1f us" % (n, timeit(lambda: F.linear(h, Wn))))
for bsz in (128, 192, 255, 256, 257, 320, 512):
    hb = torch.randn(bsz, K, device=dev)
    print("   F.linear B = %3d (N 256)  %7.1f us" % (bsz, timeit(lambda: F.linear(hb, W))))
