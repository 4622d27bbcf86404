"""Per-layer timing of pof_conv3_bn_lrelu at the DR-SPAAM shapes (B = 32 -> 72000 / 14400 sequences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STREAM = len(sys.argv) > 2 and sys.argv[2].startswith("stream")
GRAPH = len(sys.argv) > 2 and sys.argv[2] == "stream"      # streaming step: one scan per window, timed inside a graph
SA, SC = (B * 450 if STREAM else B * 450 * 5), B * 450
layers = [(SA, 1, 64, 56, 0), (SA, 64, 64, 56, 0), (SA, 64, 128, 56, 1), (SA, 128, 128, 28, 0), (SA, 128, 128, 28, 0),
          (SA, 128, 256, 28, 1), (SC, 256, 256, 14, 0), (SC, 256, 256, 14, 0), (SC, 256, 512, 14, 1),
          (SC, 512, 256, 7, 0), (SC, 256, 128, 7, 0)]
if len(sys.argv) > 3 and sys.argv[3] == "sweep":      # time vs Ci at fixed output shape: slope = per-chunk cost, intercept = fixed cost
    layers = [(SC, Ci, 256, 7, 0) for Ci in (64, 128, 256, 512, 1024)] + [(SC, Ci, 64, 56, 0) for Ci in (16, 64, 256, 1024)] \
        + [(SC, Ci, 128, 28, 0) for Ci in (32, 128, 512)]
tot_ms = tot_fl = 0.0
for (S, Ci, Co, L, pool) in layers:
    x = torch.randn((S, Ci, L), device="cuda")
    wt = torch.randn((3, Ci, Co), device="cuda") * 0.05
    sc = torch.ones(Co, device="cuda"); sh = torch.zeros(Co, device="cuda")
    out = torch.empty((S, Co, L // 2 if pool else L), device="cuda")
    for _ in range(2): ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if GRAPH:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
    else:
        e0.record()
        for _ in range(5): ops.conv3_bn_lrelu(x, wt, sc, sh, pool=bool(pool), out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * S * L * Co * Ci * 3
    byt = 4.0 * S * (Ci * L + Co * (L // 2 if pool else L))
    tot_ms += ms; tot_fl += fl
    mfma_us = -(-S * L // 32) * -(-Co // 32) * (Ci * 3 + 1) // 2 * 0.030 / 1024      # all 1024 SIMDs busy, 30 ns per 32x32x2
    print("S=%6d Ci=%3d Co=%3d L=%2d pool=%d: %7.3f ms  %6.1f TFLOP/s  %6.0f GB/s  (MFMA-issue bound %.1f us)" % (S, Ci, Co, L, pool, ms, fl / ms / 1e9, byt / ms / 1e6, mfma_us), flush=True)
print("trunk total %.2f ms  %.1f TFLOP/s" % (tot_ms, tot_fl / tot_ms / 1e9))

# the first two units in one launch (pof_conv3_first_two) against the two launches above
if not (len(sys.argv) > 3 and sys.argv[3] == "sweep"):
    S, L = SA, 56
    x0 = torch.randn((S, 1, L), device="cuda")
    table = torch.randn((64, 4), device="cuda") * 0.3
    wt = torch.randn((3, 64, 64), device="cuda") * 0.05
    sc = torch.ones(64, device="cuda"); sh = torch.zeros(64, device="cuda")
    out = torch.empty((S, 64, L), device="cuda")
    for _ in range(2): ops.conv3_first_two(x0, table, wt, sc, sh, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv3_first_two(x0, table, wt, sc, sh, out=out)
    e1.record(); torch.cuda.synchronize()
    print("S=%6d first two units fused (1 -> 64 -> 64, L=56): %7.3f ms" % (S, e0.elapsed_time(e1) / 5))

