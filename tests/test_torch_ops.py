"""torch.library registration (planar_optical_flow_amd/torch_ops.py): schemas and fake kernels on the CPU,
opcheck / autograd / torch.compile tracing on the GPU."""
import numpy as np
import pytest
import torch

from planar_optical_flow_amd import torch_ops  # noqa: F401  (registers torch.ops.pof.*)

OPS = ("band_correlation", "band_correlation_backward", "spatial_attention", "spatial_attention_backward", "cutout",
       "conv3_bn_lrelu", "rotate_flow")


def test_ops_are_registered_with_schemas():
    for name in OPS:
        op = getattr(torch.ops.pof, name)
        schema = str(op.default._schema)
        assert schema.startswith("pof::" + name + "("), schema
    assert "Tensor? g_band" in str(torch.ops.pof.spatial_attention_backward.default._schema)


def test_fake_kernels_give_the_output_shapes_without_running_anything():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        f1 = torch.empty(4, 256, 57, device="cuda", dtype=torch.float16)
        y = torch.ops.pof.band_correlation(f1, f1, 3, 5)
        assert tuple(y.shape) == (4, 11, 57) and y.dtype == torch.float32 and y.device.type == "cuda"
        ex, x = torch.empty(2, 3600, 128, device="cuda"), torch.empty(2, 3600, 3584, device="cuda", dtype=torch.float16)
        o, b, p = torch.ops.pof.spatial_attention(ex, ex, x, x, 0.5, 11)
        assert o.shape == x.shape and o.dtype == torch.float16 and tuple(b.shape) == tuple(p.shape) == (2, 3600, 11)
        tab = torch.empty(3 * 3600, dtype=torch.float64, device="cuda")
        c = torch.ops.pof.cutout(torch.empty(3, 11, 3600, device="cuda"), tab, 2, True, True, 1.0, 0.5, 56, 29.99, True, True)
        assert tuple(c.shape) == (3, 1800, 11, 56) and c.dtype == torch.float16
        z = torch.ops.pof.conv3_bn_lrelu(torch.empty(7, 64, 56, device="cuda"), torch.empty(3, 64, 128, device="cuda"),
                                         torch.empty(128, device="cuda"), torch.empty(128, device="cuda"), True, 0.1)
        assert tuple(z.shape) == (7, 128, 28)


def test_cpu_tensors_are_refused():
    with pytest.raises(NotImplementedError):
        torch.ops.pof.band_correlation(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8), 3, 2)


@pytest.mark.gpu
def test_opcheck_and_autograd_against_the_plain_torch_formulation():
    """torch.library.opcheck (schema, fake kernel, autograd registration, AOT dispatch) on real inputs, and the
    registered autograd formulas against torch autograd of the reference formulation (integer data: exact)."""
    from test_hip_parity import _torch_fusion
    gen = torch.Generator(device="cpu").manual_seed(5)
    f1 = torch.randint(-3, 4, (2, 12, 57), generator=gen).float().cuda().requires_grad_(True)
    f2 = torch.randint(-3, 4, (2, 12, 57), generator=gen).float().cuda().requires_grad_(True)
    torch.library.opcheck(torch.ops.pof.band_correlation, (f1, f2, 3, 5))
    g = torch.randint(-3, 4, (2, 11, 57), generator=gen).float().cuda()
    (torch.ops.pof.band_correlation(f1, f2, 3, 5) * g).sum().backward()
    d1, d2 = f1.grad.clone(), f2.grad.clone()
    f1.grad = f2.grad = None
    (_torch_fusion(f1, f2, 3, 5) * g).sum().backward()
    assert torch.equal(d1, f1.grad) and torch.equal(d2, f2.grad)
    ex = (torch.randn(1, 40, 128, device="cuda") * 0.3).requires_grad_(True)
    et = (torch.randn(1, 40, 128, device="cuda") * 0.3).requires_grad_(True)
    x = torch.randn(1, 40, 64, device="cuda", requires_grad=True)
    t = torch.randn(1, 40, 64, device="cuda", requires_grad=True)
    torch.library.opcheck(torch.ops.pof.spatial_attention, (ex, et, x, t, 0.5, 11),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    tab = __import__("planar_optical_flow_amd.ops", fromlist=["ops"]).phi_table()
    fl = torch.randn(3, 450, 2, device="cuda", dtype=torch.float64, requires_grad=True)
    torch.library.opcheck(torch.ops.pof.rotate_flow, (fl, tab, True))
    (torch.ops.pof.rotate_flow(fl, tab, True) * 2.0).sum().backward()
    want = torch.ops.pof.rotate_flow(torch.full_like(fl, 2.0), tab, False)      # inverse rotation of the gradient
    assert torch.allclose(fl.grad, want, rtol=0, atol=1e-14)


@pytest.mark.gpu
def test_torch_compile_traces_through_the_ops():
    """A Prototype forward + backward under torch.compile (aot_eager: the dispatcher / functionalisation / AOT
    autograd stack without a code generator) equals eager: the correlation is one opaque node with a fake kernel
    and an autograd formula instead of a graph break."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planar_optical_flow_amd"))
    from src.depracted.model.prototype import Prototype
    torch.manual_seed(7)
    model = Prototype(in_channel=1, max_displacement=5).cuda().train()
    s1, s2 = torch.randn(3, 450, 1, device="cuda"), torch.randn(3, 450, 1, device="cuda")
    import torch._dynamo as dynamo
    dynamo.reset()
    compiled = torch.compile(model, backend="aot_eager", fullgraph=True)       # fullgraph: no graph break allowed
    out_c = compiled(s1, s2)
    out_c.square().mean().backward()
    gc = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    model.zero_grad()
    out_e = model(s1, s2)
    out_e.square().mean().backward()
    ge = [p.grad for p in model.parameters() if p.grad is not None]
    # (train-mode BatchNorm decomposed by AOT autograd vs the fused eager kernel: float32 round-off only)
    assert len(gc) == len(ge) > 0
    assert torch.allclose(out_c, out_e, rtol=1e-3, atol=1e-4), (out_c - out_e).abs().max().item()
    scale = max(b.abs().max().item() for b in ge)
    for a, b in zip(gc, ge):                                    # on the global gradient scale (float32 round-off)
        assert (a - b).abs().max().item() <= 2e-3 * scale, ((a - b).abs().max().item(), scale)
