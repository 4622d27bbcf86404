"""Kernel trace target: Prototype inference (fused units) and training (units as ConvUnitTrain nodes) -- run under
`rocprofv3 --kernel-trace --stats -- python3 tools/trace_prototype.py`; the library's convolution kernels
(igemm_*, batched_transpose, naive_conv_*) must not appear in the stats."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from planar_optical_flow_amd.src.depracted.model.prototype import Prototype, flow_loss  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(7)
model = Prototype(in_channel=1, max_displacement=5).to(dev).train()
g = torch.Generator(device=dev).manual_seed(11)
s1 = torch.randn((256, 450, 1), device=dev, generator=g)
s2 = torch.randn((256, 450, 1), device=dev, generator=g)
tgt = torch.randn((256, 450, 2), device=dev, generator=g) * 0.2
for _ in range(6):
    model.zero_grad(set_to_none=True)
    loss, _ = flow_loss(model(s1, s2), tgt)
    loss.backward()
model.eval()
model.fuse_for_inference()
with torch.no_grad():
    for _ in range(6):
        out = model(s1, s2)
torch.cuda.synchronize()
print("loss %.5f, inference output %s" % (float(loss), tuple(out.shape)))
