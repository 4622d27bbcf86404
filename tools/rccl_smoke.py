"""One-rank RCCL smoke (run on a GPU box): the flat gradient bucket all-reduced on device tensors, the
SyncBatchNorm forward / backward collectives and a plain all-reduce through backend "nccl" (= RCCL on ROCm)."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
from planar_optical_flow_amd import dist as pd          # noqa: E402
from src.model.get_model import get_model               # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
cfg = {"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0}
torch.manual_seed(5)
ref = get_model(cfg).cuda().train()
torch.manual_seed(5)
model = get_model(cfg).cuda()
pd.convert_sync_batchnorm(model).train()
x = torch.randn(12, 64, 3, device="cuda")
y = torch.randn(12, 3, device="cuda")
model.loss_fn(model(x), y).backward()
ref.loss_fn(ref(x), y).backward()
red = pd.GradientAllReduce(model)
before = {i: p.grad.clone() for i, p in enumerate(red.params) if p.grad is not None}
red(force=True)                                   # ncclAllReduce on the device bucket, one rank
assert red.bucket.is_cuda and all(torch.equal(red.params[i].grad, g) for i, g in before.items())
assert all(p.grad is not None for p in red.params)     # unused parameters received the reduced zeros
# one rank: SyncBatchNorm (statistics through RCCL) == stock BatchNorm
# (gradients of a bias in front of a BatchNorm are pure round-off in both: compare on the global gradient scale)
scale = max(q.grad.abs().max().item() for q in ref.parameters() if q.grad is not None)
worst = 0.0
for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
    if p.grad is not None and q.grad is not None:
        err = (p.grad - q.grad).abs().max().item() / scale
        worst = max(worst, err)
        assert err < 1e-4, (n, err)
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
assert t.sum().item() == 4.0

# a whole optimisation step -- SyncBatchNorm collectives, the gradient all-reduce (gradients are views of the flat
# bucket), Adam -- captured as ONE hipGraph and replayed; against the same step issued eagerly (same modules, same
# collectives).  The replayed steps must not synchronise with the host (sync debug mode "error" raises if they do).
from planar_optical_flow_amd.graph_step import GraphedTrainStep, make_capturable     # noqa: E402
torch.manual_seed(6)
gm = get_model(cfg).cuda()
pd.convert_sync_batchnorm(gm).train()
torch.manual_seed(6)
em = get_model(cfg).cuda()                      # the eager twin: the same modules, collectives issued eagerly
pd.convert_sync_batchnorm(em).train()
gopt = torch.optim.Adam(gm.parameters(), lr=1e-3, amsgrad=True)
eopt = torch.optim.Adam(em.parameters(), lr=1e-3, amsgrad=True)
ered = pd.GradientAllReduce(em, always=True)
make_capturable(gopt)
red2 = pd.GradientAllReduce(gm, always=True)
gstep = GraphedTrainStep(gm, gopt, {"input": x, "target": y}, reducer=red2)
assert gstep._fused_collective and len(gstep._graphs) == 1
assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(red2.params, red2.views))
worst_loss = 0.0
for it in range(4):
    xb = torch.randn(12, 64, 3, device="cuda")
    yb = torch.randn(12, 3, device="cuda")
    red2.set_stop(it == 2)
    torch.cuda.set_sync_debug_mode("error")
    gl = gstep({"input": xb, "target": yb})
    torch.cuda.set_sync_debug_mode("default")
    eopt.zero_grad(set_to_none=False)
    el = em.loss_fn(em(xb), yb)
    el.backward()
    ered()
    eopt.step()
    worst_loss = max(worst_loss, abs(gl.item() - el.item()) / max(abs(el.item()), 1e-6))
    assert red2.stop_requested() == (it == 2), it       # the flag of THIS step's all-reduce, read after it
assert worst_loss < 1e-5, worst_loss
# the gradients of the last step (the graph's live in the flat bucket), on the scale of the largest one.  (Weights are
# not compared: where a gradient is pure round-off -- a bias in front of a BatchNorm -- Adam normalises the round-off
# to steps of +-lr on either side; the losses above do not depend on those parameters.)
gscale = max(q.grad.abs().max().item() for q in em.parameters() if q.grad is not None)
wdiff = max((p.grad - q.grad).abs().max().item() for p, q in zip(gm.parameters(), em.parameters())) / gscale
assert wdiff < 1e-4, wdiff
dist.destroy_process_group()
print("RCCL_OK worst relative gradient difference SyncBN vs BatchNorm: %.2e; captured step with collectives: "
      "loss diff %.1e, gradient diff %.1e of scale, no host sync in the replay" % (worst, worst_loss, wdiff))
