"""Box-regression head training step (BASELINE configs[3] shape: batch 256 x 64 points), eager and as one hipGraph
replay.  Under rocprofv3 --kernel-trace --stats this shows where the device time of the step goes."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
import torch
from src.model.get_model import get_model
from src.pipeline.optim import Optim
from planar_optical_flow_amd.graph_step import GraphedTrainStep

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
per = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pointwise = len(sys.argv) > 3 and sys.argv[3] == "points"
torch.manual_seed(4)
model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).cuda().train()
model.backbone.train_pointwise = pointwise
optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "epoch1": 100, "lr0": 1e-3, "lr1": 1e-6}})
x = torch.randn((per, 64, 3), device="cuda") * 0.3
y = torch.randn((per, 3), device="cuda") * 0.3
batch = {"input": x, "target": y}
if mode == "graph":
    gs = GraphedTrainStep(model, optim.make_capturable(), batch)
    step = lambda: gs(batch)
else:
    def step():
        optim.zero_grad()
        loss = model.loss_fn(model(x), y)
        loss.backward()
        optim.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
for _ in range(n): step()
torch.cuda.synchronize()
print("box head train step [%s, %s] batch %d: %.3f ms" % (mode, "points-major GEMMs" if pointwise else "Conv1d modules", per,
                                                         (time.perf_counter() - t0) / n * 1e3))
