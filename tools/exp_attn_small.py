"""Spatial attention forward / backward at small batches (streaming: 1 scan; training: 8 windows): time vs the minimum
segment length of the merge walk (POF_ATTN_LMIN)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
N, E, F = 450, 128, 3584
for B in (1, 8):
    g = torch.Generator(device="cuda").manual_seed(1)
    ex = torch.randn((B, N, E), device="cuda", generator=g) * 0.3; et = torch.randn((B, N, E), device="cuda", generator=g) * 0.3
    x = torch.randn((B, N, F), device="cuda", generator=g); t = torch.randn((B, N, F), device="cuda", generator=g)
    out, band, prob = ops.spatial_attention(ex, et, x, t, 0.5, 11)
    go, gb = torch.randn_like(out), torch.randn_like(band)
    def timed(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = timed(lambda: ops.spatial_attention(ex, et, x, t, 0.5, 11))
    tb = timed(lambda: ops.spatial_attention_backward(ex, et, t, prob, go, gb, 0.5, 11))
    print("B=%d  forward %.1f us  backward %.1f us" % (B, tf, tb))
