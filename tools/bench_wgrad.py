"""Weight-gradient pass of the trunk's conv layers (training, one scan of a B = 8 batch: S = 3600 sequences; and the
DROW shape S = 18000): pof_conv3_wgrad against the library's convolution_backward (weight only), per layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planar_optical_flow_amd import ops

S0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
P = int(sys.argv[2]) if len(sys.argv) > 2 else 56
layers = [(1, 64, P), (64, 64, P), (64, 128, P), (128, 128, P // 2), (128, 128, P // 2), (128, 256, P // 2),
          (256, 256, P // 4), (256, 256, P // 4), (256, 512, P // 4), (512, 256, P // 8), (256, 128, P // 8)]


def timed(fn, n=10):
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


tot_h = tot_l = 0.0
for ci, co, L in layers:
    x = torch.randn(S0, ci, L, device="cuda")
    dy = torch.randn(S0, co, L, device="cuda")
    w = torch.randn(co, ci, 3, device="cuda")
    th = timed(lambda: ops.conv3_wgrad(x, dy))
    tl = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1], [1], [1], False, [0], 1,
                                                           [False, True, False]))
    fl = 2.0 * S0 * L * 3 * ci * co
    tot_h += th
    tot_l += tl
    print("%4d -> %4d  L=%2d   hip %.3f ms (%5.1f TF)   library %.3f ms (%5.1f TF)" % (ci, co, L, th, fl / th / 1e9,
                                                                                      tl, fl / tl / 1e9), flush=True)
print("sum: hip %.2f ms, library %.2f ms" % (tot_h, tot_l))
