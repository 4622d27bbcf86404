"""Attention launches only at B = 64 (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
B, N, E, F = 64, 450, 128, 3584
g = torch.Generator(device="cuda").manual_seed(10)
ex = torch.randn((B, N, E), device="cuda", generator=g) * 0.3
et = torch.randn((B, N, E), device="cuda", generator=g) * 0.3
x = torch.randn((B, N, F), device="cuda", generator=g)
t = torch.randn((B, N, F), device="cuda", generator=g)
out = torch.empty_like(x)
for _ in range(10):
    ops.spatial_attention(ex, et, x, t, 0.5, 11, out=out)
torch.cuda.synchronize()
