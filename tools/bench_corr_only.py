"""Band-correlation launches only (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
B, C, n = 4096, 256, 57
f1 = torch.randn((B, C, n), device="cuda"); f2 = torch.randn((B, C, n), device="cuda")
out = torch.empty((B, 11, n), device="cuda")
for _ in range(12):
    ops.band_correlation(f1, f2, 3, 5, out=out)
torch.cuda.synchronize()
