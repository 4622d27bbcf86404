"""Experiment: band-correlation time vs batch (waves per SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
C, n = 256, 57
for B in (256, 1024, 2048, 3072, 4096, 8192, 16384):
    f1 = torch.randn((B, C, n), device="cuda"); f2 = torch.randn((B, C, n), device="cuda")
    out = torch.empty((B, 11, n), device="cuda")
    for _ in range(3): ops.band_correlation(f1, f2, 3, 5, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.band_correlation(f1, f2, 3, 5, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("B=%d: %.1f us  %.0f GB/s  %.1f us per 1024 samples" % (B, ms * 1e3, B * (2 * C * n * 4 + 11 * n * 4) / ms / 1e6, ms * 1e3 * 1024 / B), flush=True)
