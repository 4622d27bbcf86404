"""Box head (PointNet) forward / training step at BASELINE config 4 shapes: MIOpen 1x1 convs vs the
channels-last GEMM form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planar_optical_flow_amd"))
import torch
from src.model.get_model import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).cuda()
x = torch.randn(B, 64, 3, device="cuda")
tgt = torch.randn(B, 3, device="cuda")

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for flag in (False, True):
    model.backbone.gemm_pointwise = flag
    model.eval()
    with torch.no_grad():
        ms_eval = timeit(lambda: model(x))
    model.train()
    def step():
        for p in model.parameters(): p.grad = None
        model.loss_fn(model(x), tgt).backward()
    ms_train = timeit(step)
    print("box head B=%d gemm_pointwise=%s: eval %.2f ms, train step %.2f ms" % (B, flag, ms_eval, ms_train), flush=True)
