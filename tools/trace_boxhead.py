"""Kernel trace target: graphed box-regression training steps (BASELINE configs[3]) -- run under
`rocprofv3 --kernel-trace --stats -- python3 tools/trace_boxhead.py [hip|modules]`."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
import torch  # noqa: E402
from src.model.get_model import get_model  # noqa: E402
from src.pipeline.optim import Optim  # noqa: E402
from planar_optical_flow_amd.graph_step import GraphedTrainStep  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "hip"
dev = torch.device("cuda:0")
torch.manual_seed(4)
model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).to(dev)
model.backbone.hip_train = mode == "hip"
model.train()
optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "epoch1": 100, "lr0": 1e-3, "lr1": 1e-6}})
x = torch.randn((256, 64, 3), device=dev) * 0.3
y = torch.randn((256, 3), device=dev) * 0.3
batch = {"input": x, "target": y}
step = GraphedTrainStep(model, optim.make_capturable(), batch)
for _ in range(5):
    optim.set_lr(0)
    step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 200
for _ in range(n):
    optim.set_lr(0)
    step(batch)
torch.cuda.synchronize()
print("%s: %.3f ms/step (graphed, %d steps)" % (mode, (time.perf_counter() - t0) / n * 1e3, n))
