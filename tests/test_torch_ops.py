"""torch.library registration (planar_optical_flow_amd/torch_ops.py): schemas and fake kernels on the CPU,
opcheck / autograd / torch.compile tracing on the GPU."""
import numpy as np
import pytest
import torch

from planar_optical_flow_amd import torch_ops  # noqa: F401  (registers torch.ops.pof.*)

OPS = ("band_correlation", "band_correlation_backward", "spatial_attention", "spatial_attention_backward", "cutout",
       "conv3_bn_lrelu", "rotate_flow", "bn_lrelu_pool", "bn_lrelu_pool_backward", "conv3_wgrad", "conv1_wgrad",
       "conv1d_bn_lrelu", "bn_lrelu_rowmax", "bn_lrelu_rowmax_backward", "linear_bias",
       "regression_loss2")


def test_ops_are_registered_with_schemas():
    for name in OPS:
        op = getattr(torch.ops.pof, name)
        schema = str(op.default._schema)
        assert schema.startswith("pof::" + name + "("), schema
    assert "Tensor? g_band" in str(torch.ops.pof.spatial_attention_backward.default._schema)
    bn = str(torch.ops.pof.bn_lrelu_pool.default._schema)
    assert "running_mean" in bn and "!" in bn and "groups=1" in bn        # mutates its running statistics


def test_training_ops_fake_kernels():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        y = torch.empty(90, 64, 48, device="cuda")
        g = torch.empty(64, device="cuda")
        z, mu, istd = torch.ops.pof.bn_lrelu_pool(y, g, g, None, None, 0.1, 1e-5, 0.1, True, 3)
        assert tuple(z.shape) == (90, 64, 24) and tuple(mu.shape) == tuple(istd.shape) == (3 * 64,)
        dy, dg, db, ds = torch.ops.pof.bn_lrelu_pool_backward(y, z, g, g, mu, istd, 0.1, True, True, 3)
        assert dy.shape == y.shape and tuple(dg.shape) == tuple(db.shape) == tuple(ds.shape) == (64,)
        dw = torch.ops.pof.conv3_wgrad(torch.empty(90, 32, 48, device="cuda"), y)
        assert tuple(dw.shape) == (64, 32, 3)
        assert tuple(torch.ops.pof.conv1_wgrad(torch.empty(90, 32, 48, device="cuda"), y).shape) == (64, 32, 1)
        wt = torch.empty(3, 64, 128, device="cuda")
        assert tuple(torch.ops.pof.conv1d_bn_lrelu(y, wt, g, g, 2, 0.1).shape) == (90, 128, 24)
        y64 = torch.empty(90, 64, 64, device="cuda")
        zm, mu2, _ = torch.ops.pof.bn_lrelu_rowmax(y64, g, g, None, None, 0.1, 1e-5, 0.1, 2)
        assert tuple(zm.shape) == (90, 64) and tuple(mu2.shape) == (128,)
        dy2, _, _, ds2 = torch.ops.pof.bn_lrelu_rowmax_backward(y64, zm, g, g, mu2, mu2, 0.1, False, 2)
        assert dy2.shape == y64.shape and tuple(ds2.shape) == (0,)


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


# ---------------------------------------------------------------------------------- N2 training tail
def _torch_tail(y, bn, slope, pool):
    out = torch.nn.functional.leaky_relu(bn(y), slope)
    return torch.max_pool1d(out, 2) if pool else out


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,L,pool", [(37, 64, 48, False), (37, 128, 48, True), (19, 256, 12, True),
                                         (23, 128, 6, False), (5, 512, 12, True), (3, 6, 10, True),
                                         (300, 64, 48, True), (1, 8, 256, False)])
def test_bn_lrelu_pool_matches_torch_modules(S, C, L, pool):
    """Forward, running statistics and all three gradients of the fused training tail against
    BatchNorm1d(train) -> leaky_relu -> max_pool1d of torch itself (float64 copy as the referee)."""
    from planar_optical_flow_amd import torch_ops
    g = torch.Generator(device="cuda").manual_seed(S * 1000 + C + L)
    y = (torch.randn(S, C, L, device="cuda", generator=g) * 1.7 + 0.4).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.uniform_(-0.5, 0.5, generator=g)
        bn.running_mean.uniform_(-1, 1, generator=g)
        bn.running_var.uniform_(0.5, 2, generator=g)
    ref_bn = torch.nn.BatchNorm1d(C).cuda().double()
    ref_bn.load_state_dict(bn.state_dict())
    y64 = y.detach().double().requires_grad_(True)

    z = torch_ops.bn_lrelu_pool_train(y, bn, 0.1, pool)
    z64 = _torch_tail(y64, ref_bn, 0.1, pool)
    assert z.shape == z64.shape
    assert torch.allclose(z.double(), z64, rtol=1e-5, atol=2e-5)
    assert torch.allclose(bn.running_mean.double(), ref_bn.running_mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), ref_bn.running_var, rtol=1e-6, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1

    gz = torch.randn(z.shape, device="cuda", generator=g)
    z.backward(gz)
    z64.backward(gz.double())
    scale = float(y64.grad.abs().max())
    assert float((y.grad.double() - y64.grad).abs().max()) <= 2e-5 * max(scale, 1.0)
    for got, want in ((bn.weight.grad, ref_bn.weight.grad), (bn.bias.grad, ref_bn.bias.grad)):
        assert float((got.double() - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,L,groups", [(256, 1024, 64, 1), (37, 24, 4, 1), (12, 16, 256, 2), (9, 5, 8, 1), (40, 128, 32, 1)])
def test_bn_lrelu_rowmax_matches_torch(S, C, L, groups):
    """pool mode 2 -- max over the whole row inside the apply pass, gradient routed to the row's first maximum -- against
    BatchNorm1d(train) -> leaky_relu -> torch.max(dim=2) in float64 (forward, running statistics, all gradients)."""
    from planar_optical_flow_amd import ops
    g = torch.Generator(device="cuda").manual_seed(S + C + L)
    y = torch.randn(S, C, L, device="cuda", generator=g) * 1.5 + 0.3
    gam = torch.rand(C, device="cuda", generator=g) + 0.5
    bet = torch.randn(C, device="cuda", generator=g) * 0.2
    dz = torch.randn(S, C, device="cuda", generator=g)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    z, mu, istd = ops.bn_lrelu_pool_forward(y, gam, bet, rm, rv, pool=2, groups=groups)
    dy, dgam, dbet, dsum = ops.bn_lrelu_pool_backward(y, dz, gam, bet, mu, istd, pool=2, bias_grad=True, groups=groups)
    assert z.shape == (S, C)
    y64 = y.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm64, rv64 = torch.zeros(C, device="cuda", dtype=torch.float64), torch.ones(C, device="cuda", dtype=torch.float64)
    outs = []
    for part in y64.chunk(groups, dim=0):
        u = torch.nn.functional.batch_norm(part, rm64, rv64, g64, b64, True, 0.1, 1e-5)
        outs.append(torch.max(torch.nn.functional.leaky_relu(u, 0.1), 2)[0])
    z64 = torch.cat(outs, dim=0)
    z64.backward(dz.double())
    assert float((z.double() - z64).abs().max()) <= 2e-5 * float(z64.abs().max())
    for got, want in ((dy, y64.grad), (dgam, g64.grad), (dbet, b64.grad)):
        assert float((got.double() - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1.0)
    assert float(dsum.abs().max()) <= 1e-3 * max(float(dz.abs().sum(dim=0).max()), 1.0)     # zero in exact arithmetic
    assert torch.allclose(rm.double(), rm64, rtol=1e-5, atol=1e-6) and torch.allclose(rv.double(), rv64, rtol=1e-5, atol=1e-6)
    # ties: the FIRST maximum of a row takes the whole gradient
    yt = torch.zeros(2, 4, 8, device="cuda")
    yt[:, :, 3] = 1.0
    yt[:, :, 6] = 1.0
    yt[0] *= 2.0
    zt, mt, it = ops.bn_lrelu_pool_forward(yt, torch.ones(4, device="cuda"), torch.zeros(4, device="cuda"), pool=2)
    dyt, _, dbt = ops.bn_lrelu_pool_backward(yt, torch.ones(2, 4, device="cuda"), torch.ones(4, device="cuda"),
                                             torch.zeros(4, device="cuda"), mt, it, pool=2)
    assert torch.equal(dbt, torch.full((4,), 2.0, device="cuda"))
    assert not ops.bn_lrelu_pool_supported(4, 4, 12, pool=2) and ops.bn_lrelu_pool_supported(4, 4, 16, pool=2)
    with pytest.raises(ValueError):
        ops.bn_lrelu_pool_forward(torch.zeros(4, 4, 12, device="cuda"), torch.ones(4, device="cuda"),
                                  torch.zeros(4, device="cuda"), pool=2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,N", [(256, 1024, 512), (256, 512, 256), (256, 256, 3), (1, 8, 1), (37, 100, 45), (300, 12, 70),
                                   (1024, 512, 256), (5, 4, 33)])
def test_linear_bias_equals_torch_linear(B, K, N):
    """pof::linear_bias (the box head's dense layers, src/model/box_regression.py:26-45) against torch's float64 Linear;
    exact on small integers; gradients through the registered autograd formula against F.linear's."""
    from planar_optical_flow_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B + K + N)
    x = torch.randn(B, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    want = torch.nn.functional.linear(x.double(), w.double(), b.double())
    got = ops.linear_bias(x, w, b)
    assert float((got.double() - want).abs().max()) <= 2e-6 * K ** 0.5 * max(float(want.abs().max()), 1.0)
    assert torch.equal(got, ops.linear_bias(x, w, b))                      # deterministic
    xi = torch.randint(-4, 5, (B, K), device="cuda", generator=g).float()
    wi = torch.randint(-4, 5, (N, K), device="cuda", generator=g).float()
    assert torch.equal(ops.linear_bias(xi, wi).double(), xi.double() @ wi.double().t())     # no bias; exact sums
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, b))
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    up = torch.randn(B, N, device="cuda", generator=g)
    torch.ops.pof.linear_bias(xa, wa, ba).backward(up)
    torch.nn.functional.linear(xr, wr, br).backward(up)
    for a_, r_ in ((xa, xr), (wa, wr), (ba, br)):
        assert float((a_.grad - r_.grad).abs().max()) <= 1e-4 * max(float(r_.grad.abs().max()), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(256, 3), (48, 5), (1, 3), (3000, 5)])
def test_regression_loss2_equals_the_composed_form(B, T):
    """pof::regression_loss2 (src/model/box_regression.py:52-67) -- loss and gradient in one launch -- against the
    composed torch form in float64, and through the model's own loss function."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planar_optical_flow_amd"))
    from src.model.box_regression import regression_loss2
    g = torch.Generator(device="cuda").manual_seed(B + T)
    pred = torch.randn(B, T, device="cuda", generator=g)
    tgt = torch.randn(B, T, device="cuda", generator=g)
    pred[0, 0] = tgt[0, 0]                                       # |x| at 0: gradient 0 (torch.abs' convention)
    p64 = pred.double().cpu().requires_grad_(True)
    want = regression_loss2(p64, tgt.double().cpu(), alpha=0.3)             # CPU float64: the composed reference form
    want.backward()
    p = pred.clone().requires_grad_(True)
    got = regression_loss2(p, tgt, alpha=0.3)
    assert got.dim() == 0 and abs(float(got) - float(want)) <= 2e-7 * abs(float(want))
    (got * 2.0).backward()
    assert float((p.grad.double().cpu() - 2.0 * p64.grad).abs().max()) <= 1e-7 * float(p64.grad.abs().max()) * 2.0 + 1e-12
    assert float(p.grad[0, 0]) == 0.0
    torch.library.opcheck(torch.ops.pof.regression_loss2, (pred.clone().requires_grad_(True), tgt, 0.5),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    assert regression_loss2(torch.zeros(4, 4, device="cuda"), torch.zeros(4, 4, device="cuda")) is None   # as the reference


@pytest.mark.gpu
def test_linear_bias_rejects_what_it_cannot_take():
    from planar_optical_flow_amd import ops
    with pytest.raises(ValueError):
        ops.linear_bias(torch.zeros(4, 6, device="cuda"), torch.zeros(3, 6, device="cuda"))      # K % 4
    with pytest.raises(ValueError):
        ops.linear_bias(torch.zeros(4, 8, device="cuda"), torch.zeros(3, 12, device="cuda"))
    with pytest.raises((ValueError, RuntimeError, TypeError)):
        ops.linear_bias(torch.zeros(4, 8), torch.zeros(3, 8))                                        # CPU tensors: no fallback
    assert ops.linear_bias(torch.zeros(0, 8, device="cuda"), torch.zeros(3, 8, device="cuda")).shape == (0, 3)


@pytest.mark.gpu
def test_bn_lrelu_pool_rejects_unsupported_shapes():
    from planar_optical_flow_amd import ops
    assert not ops.bn_lrelu_pool_supported(4, 3, 5)            # C*L % 4 != 0
    assert not ops.bn_lrelu_pool_supported(4, 4, 300)          # L > 256
    assert not ops.bn_lrelu_pool_supported(4, 4, 7, pool=True)  # odd L cannot be pooled in pairs
    with pytest.raises(ValueError):
        ops.bn_lrelu_pool_forward(torch.zeros(4, 3, 5, device="cuda"), torch.ones(3, device="cuda"),
                                  torch.zeros(3, device="cuda"))
    with pytest.raises(TypeError):
        ops.bn_lrelu_pool_forward(torch.zeros(4, 4, 8), torch.ones(4), torch.zeros(4))   # CPU tensors: no fallback


@pytest.mark.gpu
def test_drow_training_step_fused_tail_matches_module_path():
    """One DROW training step (loss, every parameter gradient, running statistics) with the fused tail against
    the same step through the plain torch modules."""
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import DROW
    torch.manual_seed(3)
    a = DROW(num_scans=2).cuda().train()
    b = DROW(num_scans=2).cuda().train()
    b.load_state_dict(a.state_dict())
    b.fused_train_tail = False
    x = torch.rand(2, 60, 2, 48, device="cuda") * 3
    outs = []
    for m in (a, b):
        cls, reg = m(x)
        loss = cls.square().mean() + reg.square().mean()
        loss.backward()
        outs.append((cls.detach(), reg.detach(), loss.detach()))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-3, atol=1e-4)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-3, atol=1e-4)
    gscale = max(float(p.grad.abs().max()) for p in b.parameters() if p.grad is not None)
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            assert float((p.grad - q.grad).abs().max()) <= 2e-3 * gscale, n
    for (n, p), q in zip(a.named_buffers(), b.buffers()):
        assert torch.allclose(p.double(), q.double(), rtol=1e-4, atol=1e-5), n


@pytest.mark.gpu
@pytest.mark.parametrize("S,Ci,Co,L", [(50, 1, 64, 48), (33, 64, 128, 48), (21, 256, 512, 12), (17, 512, 256, 6),
                                        (9, 256, 128, 7)])
def test_conv3_train_matches_conv1d(S, Ci, Co, L):
    """Forward, data, weight and bias gradients of the HIP training convolution against torch's Conv1d."""
    from planar_optical_flow_amd import torch_ops
    torch.manual_seed(S + Ci)
    conv = torch.nn.Conv1d(Ci, Co, 3, padding=1).cuda()
    ref = torch.nn.Conv1d(Ci, Co, 3, padding=1).cuda().double()
    ref.load_state_dict(conv.state_dict())
    x = torch.randn(S, Ci, L, device="cuda", requires_grad=True)
    x64 = x.detach().double().requires_grad_(True)
    y = torch_ops.conv3_train(x, conv)
    y64 = ref(x64)
    assert float((y.detach().double() - y64.detach()).abs().max()) <= 2e-5 * max(float(y64.detach().abs().max()), 1.0)
    gy = torch.randn_like(y)
    y.backward(gy)
    y64.backward(gy.double())
    for got, want in ((x.grad, x64.grad), (conv.weight.grad, ref.weight.grad), (conv.bias.grad, ref.bias.grad)):
        assert float((got.double() - want).abs().max()) <= 5e-5 * max(float(want.abs().max()), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("S,Ci,Co,L,pool", [(40, 1, 64, 48, False), (33, 64, 128, 48, True), (21, 256, 512, 12, True),
                                             (17, 512, 256, 6, False)])
def test_trunk_unit_train_matches_modules(S, Ci, Co, L, pool):
    """The one-node training unit (HIP conv + fused tail + bias gradient from the dgrad pass) against
    Conv1d -> BatchNorm1d(train) -> LeakyReLU -> max_pool1d in float64."""
    from planar_optical_flow_amd import torch_ops
    torch.manual_seed(S + Co)
    conv, bn = torch.nn.Conv1d(Ci, Co, 3, padding=1).cuda(), torch.nn.BatchNorm1d(Co).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    rconv, rbn = torch.nn.Conv1d(Ci, Co, 3, padding=1).cuda().double(), torch.nn.BatchNorm1d(Co).cuda().double()
    rconv.load_state_dict(conv.state_dict())
    rbn.load_state_dict(bn.state_dict())
    x = torch.randn(S, Ci, L, device="cuda", requires_grad=True)
    x64 = x.detach().double().requires_grad_(True)
    z = torch_ops.trunk_unit_train(x, conv, bn, 0.1, pool)
    z64 = _torch_tail(rconv(x64), rbn, 0.1, pool)
    assert torch.allclose(z.detach().double(), z64.detach(), rtol=1e-4, atol=1e-4)
    gz = torch.randn_like(z)
    z.backward(gz)
    z64.backward(gz.double())
    wscale = float(rconv.weight.grad.abs().max())
    for got, want, scale in ((x.grad, x64.grad, float(x64.grad.abs().max())),
                             (conv.weight.grad, rconv.weight.grad, wscale),
                             (conv.bias.grad, rconv.bias.grad, wscale),     # zero up to rounding in both
                             (bn.weight.grad, rbn.weight.grad, float(rbn.weight.grad.abs().max())),
                             (bn.bias.grad, rbn.bias.grad, float(rbn.bias.grad.abs().max()))):
        assert float((got.double() - want).abs().max()) <= 1e-4 * max(scale, 1.0)
    assert torch.allclose(bn.running_var.double(), rbn.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("S,Ci,Co,L", [(50, 1, 64, 56), (33, 64, 64, 48), (29, 64, 128, 56), (21, 128, 256, 28),
                                        (40, 256, 512, 14), (37, 512, 256, 7), (19, 256, 128, 6), (11, 3, 5, 9),
                                        (70, 96, 160, 12), (300, 128, 128, 24), (1, 64, 64, 56)])
def test_conv3_wgrad_matches_autograd(S, Ci, Co, L):
    """Split-K MFMA weight gradient against torch's float64 convolution weight gradient (ragged channel counts,
    odd lengths, sequence counts that do not fill the last stage / split)."""
    from planar_optical_flow_amd import ops
    g = torch.Generator(device="cuda").manual_seed(S * 7 + Ci + Co + L)
    x = torch.randn(S, Ci, L, device="cuda", generator=g)
    dy = torch.randn(S, Co, L, device="cuda", generator=g)
    w = torch.zeros(Co, Ci, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv1d(x.double(), w, padding=1).backward(dy.double())
    dw = ops.conv3_wgrad(x, dy)
    assert dw.shape == (Co, Ci, 3)
    scale = float(w.grad.abs().max())
    assert float((dw.double() - w.grad).abs().max()) <= 2e-5 * max(scale, 1.0)


@pytest.mark.gpu
def test_conv3_wgrad_exact_on_integer_data_and_deterministic():
    from planar_optical_flow_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randint(-3, 4, (200, 64, 28), device="cuda", generator=g).float()
    dy = torch.randint(-3, 4, (200, 128, 28), device="cuda", generator=g).float()
    w = torch.zeros(128, 64, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv1d(x.double(), w, padding=1).backward(dy.double())
    dw = ops.conv3_wgrad(x, dy)
    assert torch.equal(dw.double(), w.grad)            # sums of small integers: exact in float32
    assert torch.equal(dw, ops.conv3_wgrad(x, dy))
    xr, dyr = torch.randn_like(x), torch.randn_like(dy)
    assert torch.equal(ops.conv3_wgrad(xr, dyr), ops.conv3_wgrad(xr, dyr))   # no atomics: bit-stable


@pytest.mark.gpu
@pytest.mark.parametrize("S,Ci,Co,L", [(256, 128, 1024, 64), (64, 3, 64, 64), (31, 64, 128, 64), (17, 70, 33, 31),
                                        (5, 64, 64, 1), (9, 128, 2, 60), (33, 64, 64, 48)])
def test_conv1_wgrad_matches_autograd_and_is_exact_on_integers(S, Ci, Co, L):
    """One-tap form of the split-K kernel (the PointNet's / the Prototype head's point-wise convolutions) against
    torch's float64 weight gradient; on small integers every sum is exact in float32."""
    from planar_optical_flow_amd import ops
    g = torch.Generator(device="cuda").manual_seed(S + 3 * Ci + 5 * Co + L)
    for integers in (False, True):
        if integers:
            x = torch.randint(-3, 4, (S, Ci, L), device="cuda", generator=g).float()
            dy = torch.randint(-3, 4, (S, Co, L), device="cuda", generator=g).float()
        else:
            x = torch.randn(S, Ci, L, device="cuda", generator=g)
            dy = torch.randn(S, Co, L, device="cuda", generator=g)
        want = torch.einsum("sol,sil->oi", dy.double(), x.double()).unsqueeze(-1)
        assert ops.conv3_wgrad_supported(S, Ci, Co, L, 1)
        dw = ops.conv3_wgrad(x, dy, kernel_size=1)
        assert dw.shape == (Co, Ci, 1)
        if integers:
            assert torch.equal(dw.double(), want)
        else:
            assert float((dw.double() - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1.0)
        assert torch.equal(dw, ops.conv3_wgrad(x, dy, kernel_size=1))


@pytest.mark.gpu
def test_pointwise_weight_grad_chunks_long_rows():
    """Rows longer than the kernel's LDS stage (the Prototype head's 450 points) are cut into chunks."""
    from planar_optical_flow_amd import torch_ops, ops
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(6, 64, 450, device="cuda", generator=g)
    dy = torch.randn(6, 2, 450, device="cuda", generator=g)
    assert not ops.conv3_wgrad_supported(6, 64, 2, 450, 1)
    dw = torch_ops._pointwise_weight_grad(x, dy, torch.empty(2, 64, 1, device="cuda"))
    want = torch.einsum("sol,sil->oi", dy.double(), x.double()).unsqueeze(-1)
    assert float((dw.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.gpu
def test_conv3_train_exact_on_integer_data():
    """Forward and data gradient of the training convolution on small integers: every product and partial sum is
    exact in float32, so the result must equal torch's float64 convolution exactly."""
    from planar_optical_flow_amd import torch_ops
    g = torch.Generator(device="cuda").manual_seed(9)
    conv = torch.nn.Conv1d(64, 128, 3, padding=1).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randint(-2, 3, conv.weight.shape, device="cuda", generator=g).float())
        conv.bias.copy_(torch.randint(-2, 3, conv.bias.shape, device="cuda", generator=g).float())
    ref = torch.nn.Conv1d(64, 128, 3, padding=1).cuda().double()
    ref.load_state_dict(conv.state_dict())
    x = torch.randint(-3, 4, (90, 64, 28), device="cuda", generator=g).float().requires_grad_(True)
    x64 = x.detach().double().requires_grad_(True)
    y = torch_ops.conv3_train(x, conv)
    y64 = ref(x64)
    assert torch.equal(y.detach().double(), y64.detach())
    gy = torch.randint(-2, 3, y.shape, device="cuda", generator=g).float()
    y.backward(gy)
    y64.backward(gy.double())
    assert torch.equal(x.grad.double(), x64.grad)
    assert torch.equal(conv.weight.grad.double(), ref.weight.grad)
    assert torch.equal(conv.bias.grad.double(), ref.bias.grad)


@pytest.mark.gpu
def test_training_trunk_under_no_grad_and_frozen_weights():
    """Training-mode statistics without autograd (BatchNorm calibration) and with frozen trunk weights
    (FlowDROW_pretrained freezes DR-SPAAM): the fused unit runs forward only and still updates the running stats."""
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import DROW
    torch.manual_seed(4)
    m = DROW(num_scans=2).cuda().train()
    ref = DROW(num_scans=2).cuda().train()
    ref.load_state_dict(m.state_dict())
    ref.fused_train_tail = False
    for p in list(m.parameters()) + list(ref.parameters()):
        p.requires_grad_(False)
    x = torch.rand(2, 40, 2, 48, device="cuda") * 3
    with torch.no_grad():
        a = m(x)
    b = ref(x)
    assert torch.allclose(a[0], b[0], rtol=1e-3, atol=1e-4) and torch.allclose(a[1], b[1], rtol=1e-3, atol=1e-4)
    for (n, p), q in zip(m.named_buffers(), ref.buffers()):
        assert torch.allclose(p.double(), q.double(), rtol=1e-4, atol=1e-5), n


@pytest.mark.gpu
def test_training_trunk_compiles():
    """torch.compile of a training step through the fused units: the custom ops carry fake kernels, the shape
    probes are plain Python -- whatever the compiler makes of the frame, the step must agree with eager."""
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import DROW
    torch.manual_seed(6)
    m = DROW(num_scans=2).cuda().train()
    ref = DROW(num_scans=2).cuda().train()
    ref.load_state_dict(m.state_dict())
    x = torch.rand(2, 40, 2, 48, device="cuda") * 3

    def step(model, xx):
        cls, reg = model(xx)
        return cls.square().mean() + reg.square().mean()

    loss_c = torch.compile(step)(m, x)
    loss_c.backward()
    loss_e = step(ref, x)
    loss_e.backward()
    assert torch.allclose(loss_c, loss_e, rtol=1e-4, atol=1e-6)
    gscale = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for (n, p), q in zip(m.named_parameters(), ref.parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 2e-3 * gscale, n


@pytest.mark.gpu
@pytest.mark.parametrize("G,Sg,C,L,pool", [(5, 40, 64, 48, True), (3, 17, 128, 12, False), (2, 1, 8, 6, True)])
def test_bn_lrelu_pool_groups_equal_separate_calls(G, Sg, C, L, pool):
    """`groups` statistics groups in one launch against G separate calls on the G slices: outputs, saved statistics,
    running statistics (updated in order) bit-identical; the gradients of the slices too, gamma / beta gradients the
    sums over the groups."""
    from planar_optical_flow_amd import ops
    g = torch.Generator(device="cuda").manual_seed(G * 100 + C)
    y = torch.randn(G * Sg, C, L, device="cuda", generator=g) * 1.5 + 0.3
    for k in range(G):
        y[k * Sg:(k + 1) * Sg] += k * 0.7            # different statistics per group
    gam = torch.rand(C, device="cuda", generator=g) + 0.5
    bet = torch.rand(C, device="cuda", generator=g) - 0.5
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    z, mu, istd = ops.bn_lrelu_pool_forward(y, gam, bet, rm, rv, pool=pool, groups=G)
    rm1, rv1 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    dz = torch.randn(z.shape, device="cuda", generator=g)
    dy, dgam, dbet, dsum = ops.bn_lrelu_pool_backward(y, dz, gam, bet, mu, istd, pool=pool, bias_grad=True, groups=G)
    acc_g, acc_b, acc_s = (torch.zeros(C, device="cuda", dtype=torch.float64) for _ in range(3))
    for k in range(G):
        sl = slice(k * Sg, (k + 1) * Sg)
        zk, muk, ik = ops.bn_lrelu_pool_forward(y[sl].contiguous(), gam, bet, rm1, rv1, pool=pool)
        assert torch.equal(zk, z[sl]) and torch.equal(muk, mu[k * C:(k + 1) * C]) and torch.equal(ik, istd[k * C:(k + 1) * C])
        dyk, dgk, dbk, dsk = ops.bn_lrelu_pool_backward(y[sl].contiguous(), dz[sl].contiguous(), gam, bet, muk, ik,
                                                        pool=pool, bias_grad=True)
        assert torch.equal(dyk, dy[sl])
        acc_g += dgk.double(); acc_b += dbk.double(); acc_s += dsk.double()
    assert torch.equal(rm, rm1) and torch.equal(rv, rv1)
    for got, want in ((dgam, acc_g), (dbet, acc_b), (dsum, acc_s)):
        assert float((got.double() - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), 1.0)


@pytest.mark.gpu
def test_spatial_drow_grouped_scans_equal_scan_by_scan():
    """SpatialDROW training step with all scans of the window stacked through blocks 1-2 (statistics groups)
    against the scan-by-scan pass of the same fused units and against the plain torch modules: predictions,
    every parameter gradient, running statistics and batch counters."""
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW
    torch.manual_seed(12)
    kw = dict(num_scans=3, num_pts=48, alpha=0.5, window_size=7, pedestrian_only=True)
    models = [SpatialDROW(**kw).cuda().train() for _ in range(3)]
    for m in models[1:]:
        m.load_state_dict(models[0].state_dict())
    models[1].grouped_scans = False
    models[2].fused_train_tail = False
    x = torch.rand(2, 50, 3, 48, device="cuda") * 3
    outs = []
    for m in models:
        cls, reg, sim = m(x)
        (cls.square().mean() + reg.square().mean() + sim.square().mean() * 1e-3).backward()
        outs.append((cls.detach(), reg.detach()))
    assert models[0]._grouped_blocks_ok(("conv_block_1", "conv_block_2"), 3 * 2 * 50, 48, torch.float32, 3)
    for k, tol in ((1, 1e-5), (2, 1e-3)):
        assert torch.allclose(outs[0][0], outs[k][0], rtol=tol, atol=tol * 0.1)
        assert torch.allclose(outs[0][1], outs[k][1], rtol=tol, atol=tol * 0.1)
        gscale = max(float(p.grad.abs().max()) for p in models[k].parameters() if p.grad is not None)
        for (n, p), q in zip(models[0].named_parameters(), models[k].parameters()):
            assert (p.grad is None) == (q.grad is None), n
            if p.grad is not None:
                assert float((p.grad - q.grad).abs().max()) <= (2e-4 if k == 1 else 2e-3) * gscale, (k, n)
        for (n, p), q in zip(models[0].named_buffers(), models[k].buffers()):
            assert torch.allclose(p.double(), q.double(), rtol=1e-4, atol=1e-5), (k, n)


@pytest.mark.gpu
def test_fused_tail_refuses_one_value_per_channel_like_torch():
    from planar_optical_flow_amd import torch_ops
    bn = torch.nn.BatchNorm1d(8).cuda().train()
    y = torch.randn(1, 8, 1, device="cuda")
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        bn(y)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        torch_ops.bn_lrelu_pool_train(y, bn, 0.1, False)


@pytest.mark.gpu
@pytest.mark.parametrize("S,Ci,Co,L,K,stride,groups", [(6, 1, 64, 450, 3, 2, 2), (4, 64, 128, 225, 3, 2, 2), (4, 128, 256, 113, 3, 2, 1),
                                                        (3, 139, 128, 113, 3, 1, 1), (2, 192, 128, 225, 3, 1, 1), (5, 16, 8, 64, 1, 1, 1),
                                                        (4, 6, 12, 7, 3, 2, 2), (3, 5, 16, 9, 3, 1, 1)])
def test_conv_unit_train_equals_modules(S, Ci, Co, L, K, stride, groups):
    """torch_ops.ConvUnitTrain -- Conv1d(k = 1 | 3, stride 1 | 2) -> BatchNorm1d(train) -> LeakyReLU(0.01) as one autograd node
    on the HIP kernels (stride-2 gradients through the even / odd decomposition, weight gradients of long sequences
    through halo chunks) -- against the torch modules in float64: output, running statistics and all five gradients
    to 1e-4 of each tensor's scale; `groups` batches normalised separately like separate module calls."""
    import torch.nn as nn
    from planar_optical_flow_amd import torch_ops
    torch.manual_seed(S * 100 + Ci + K + stride)
    conv = nn.Conv1d(Ci, Co, K, stride=stride, padding=K // 2).cuda()
    bn = nn.BatchNorm1d(Co).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    conv64, bn64 = nn.Conv1d(Ci, Co, K, stride=stride, padding=K // 2).cuda().double(), nn.BatchNorm1d(Co).cuda().double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    bn64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    x = torch.randn(S, Ci, L, device="cuda", requires_grad=True)
    x64 = x.detach().double().requires_grad_(True)
    z = torch_ops.conv_unit_train(x, conv, bn, 0.01, groups=groups)
    z64 = torch.cat([torch.nn.functional.leaky_relu(bn64(conv64(part)), 0.01) for part in x64.chunk(groups, dim=0)], dim=0)
    up = torch.randn_like(z)
    z.backward(up)
    z64.backward(up.double())

    def close(a, b, what, rel=1e-4):
        scale = float(b.abs().max()) or 1.0
        err = float((a.double() - b).abs().max())
        assert err <= rel * scale, (what, err, scale)
    close(z, z64, "output")
    close(bn.running_mean, bn64.running_mean, "running mean")
    close(bn.running_var, bn64.running_var, "running var")
    assert int(bn.num_batches_tracked) == groups
    close(x.grad, x64.grad, "d x")
    close(conv.weight.grad, conv64.weight.grad, "d weight")
    close(bn.weight.grad, bn64.weight.grad, "d gamma")
    close(bn.bias.grad, bn64.bias.grad, "d beta")
    # the conv bias in front of a BatchNorm has a zero gradient: round-off on both sides, on the scale of d gamma
    assert float(conv.bias.grad.abs().max()) <= 1e-4 * float(bn64.weight.grad.abs().max()) + 1e-6


@pytest.mark.gpu
def test_prototype_hip_training_equals_module_training():
    """Prototype.forward in training mode on the device: every unit through ConvUnitTrain / Conv1dTrain (hip_train) against
    the same model through the torch modules: prediction, loss, every gradient (1e-3 of the largest gradient) and every
    BatchNorm's running statistics and batch counter."""
    import copy
    from planar_optical_flow_amd.src.depracted.model.prototype import Prototype, flow_loss
    torch.manual_seed(11)
    a = Prototype(in_channel=1, max_displacement=5).cuda().train()
    b = copy.deepcopy(a)
    b.hip_train = False
    s1, s2 = torch.randn(6, 450, 1, device="cuda"), torch.randn(6, 450, 1, device="cuda")
    tgt = torch.randn(6, 450, 2, device="cuda") * 0.2
    pa, pb = a(s1, s2), b(s1, s2)
    assert float((pa - pb).abs().max()) <= 1e-3 * float(pb.abs().max())
    la, lb = flow_loss(pa, tgt)[0], flow_loss(pb, tgt)[0]
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-4 * abs(float(lb))
    top = max(float(q.grad.abs().max()) for q in b.parameters())
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 1e-3 * top, n
    for (n, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        assert float((u.double() - v.double()).abs().max()) <= 1e-4 * max(1.0, float(v.double().abs().max())), n
