"""Kernel trace target: eager box-regression training steps, the last one bracketed by marker fills (see trace_order.py)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
import torch  # noqa: E402
from src.model.get_model import get_model  # noqa: E402
from src.pipeline.optim import Optim  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "hip"
dev = torch.device("cuda:0")
torch.manual_seed(4)
model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).to(dev)
model.backbone.hip_train = mode == "hip"
model.train()
optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "epoch1": 100, "lr0": 1e-3, "lr1": 1e-6}})
x = torch.randn((256, 64, 3), device=dev) * 0.3
y = torch.randn((256, 3), device=dev) * 0.3
mark = torch.empty(12345, dtype=torch.float64, device=dev)
for i in range(8):
    if i == 7:
        torch.cuda.synchronize()
        mark.fill_(1.0)
    optim.zero_grad()
    optim.set_lr(0)
    loss = model.loss_fn(model(x), y)
    loss.backward()
    optim.step()
mark.fill_(2.0)
torch.cuda.synchronize()
