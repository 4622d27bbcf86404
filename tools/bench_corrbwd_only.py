"""Band-correlation backward launches only (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
B, C, n = 4096, 256, 57
f1 = torch.randn((B, C, n), device="cuda"); f2 = torch.randn((B, C, n), device="cuda")
g = torch.randn((B, 11, n), device="cuda")
for _ in range(8):
    ops.band_correlation_backward(f1, f2, g, 3, 5)
torch.cuda.synchronize()
