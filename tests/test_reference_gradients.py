"""Backward passes against gradients produced by the REFERENCE's own autograd graph (tests/golden/gradients.npz,
written by tools/gen_golden.py::gen_gradients on the CPU of the build container):

  A9   Prototype._fusion                        prototype.py:118-156
  A10  _SpatialAttention.forward                dr_spaam.py:163-217
  N2   one SpatialDROW / Prototype / box-head training step (loss + parameter gradients)
                                                dr_spaam.py:41-121, 220-277; prototype.py:57-109;
                                                src/model/box_regression.py:20-143

Bar: 1e-4 of the gradient's scale (max |g| of the tensor) -- float32 sums in a different order on both sides.
"""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, rel=1e-4, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    scale = float(np.abs(want).max()) or 1.0
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert err <= rel * scale, "%s: max error %.3e against scale %.3e (bar %.1e of scale)" % (what, err, scale, rel)


@pytest.fixture(scope="module")
def ops():
    from planar_optical_flow_amd import ops as o
    return o


@pytest.mark.parametrize("tag,K,md", [("corr", 3, 5), ("corr_s", 3, 3)])
def test_band_correlation_backward_equals_reference_autograd(ops, golden, tag, K, md):
    g = golden("gradients")
    f1, f2, up = T(g[tag + "_f1"]), T(g[tag + "_f2"]), T(g[tag + "_up"])
    out = ops.band_correlation(f1, f2, K, md)
    close(out, g[tag + "_out"], 1e-4, tag + " forward")
    d1, d2 = ops.band_correlation_backward(f1, f2, up, K, md)
    close(d1, g[tag + "_d1"], 1e-4, tag + " d feat1")
    close(d2, g[tag + "_d2"], 1e-4, tag + " d feat2")
    # and through the registered autograd op the Prototype network calls
    from src.depracted.model.prototype import fusion
    a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    (fusion(a, b, K, md) * up).sum().backward()
    close(a.grad, g[tag + "_d1"], 1e-4, tag + " d feat1 (torch op)")
    close(b.grad, g[tag + "_d2"], 1e-4, tag + " d feat2 (torch op)")


@pytest.mark.parametrize("tag,alpha,w", [("attn", 0.5, 11), ("attn_w7", 0.3, 7)])
def test_spatial_attention_backward_equals_reference_autograd(ops, golden, tag, alpha, w):
    g = golden("gradients")
    x, t = g[tag + "_x"], g[tag + "_t"]
    B, N, C, P = x.shape
    # (i) the HIP backward op on the reference's own embeddings
    ex, et = T(g[tag + "_emb_x"].reshape(B, N, 128)), T(g[tag + "_emb_t"].reshape(B, N, 128))
    xf, tf = T(x.reshape(B, N, C * P)), T(t.reshape(B, N, C * P))
    out, band, prob = ops.spatial_attention(ex, et, xf, tf, alpha, w)
    close(out.reshape(B, N, C, P), g[tag + "_out"], 1e-4, tag + " out")
    close(band, g[tag + "_band"], 1e-4, tag + " band")
    up_out, up_band = T(g[tag + "_up_out"].reshape(B, N, C * P)), T(g[tag + "_up_band"])
    for fused in (True, False):
        dex, det, dx, dt = ops.spatial_attention_backward(ex, et, tf, prob, up_out, up_band, alpha, w, fused=fused)
        # the reference's x / template gradients also hold the path through the embedding convolution: compare
        # the op's share only where it is the whole gradient (d emb), the rest through the module below
        close(dex.reshape(B * N, 128, 1), g[tag + "_demb_x"], 1e-4, tag + " d emb_x fused=%s" % fused)
        close(det.reshape(B * N, 128, 1), g[tag + "_demb_t"], 1e-4, tag + " d emb_t fused=%s" % fused)
    # (ii) the module (embedding GEMM + BatchNorm(eval) + LeakyReLU + HIP gate) with the reference's parameters
    from src.depracted.model.dr_spaam import _SpatialAttention
    att = _SpatialAttention(n_pts=P, n_channel=C, alpha=alpha, window_size=w).to(DEV)
    sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd_")}
    att.load_state_dict({k.replace("conv_0_", "conv.0.").replace("conv_1_", "conv.1."): v for k, v in sd.items()})
    att.eval()
    xd, td = T(x).requires_grad_(True), T(t).requires_grad_(True)
    o2, b2 = att(xd, td)
    close(o2, g[tag + "_out"], 1e-4, tag + " module out")
    torch.autograd.backward([o2, b2], [T(g[tag + "_up_out"]), T(g[tag + "_up_band"])])
    close(xd.grad, g[tag + "_dx"], 1e-4, tag + " d x")
    close(td.grad, g[tag + "_dt"], 1e-4, tag + " d x_template")
    for k, p in att.named_parameters():
        close(p.grad, g[tag + "_dp_" + k.replace(".", "_")], 1e-4, tag + " d " + k)


def test_spatial_drow_training_step_equals_reference(golden):
    """One training step of SpatialDROW (seeded weights = the reference's, BatchNorm with batch statistics, the
    reference's own loss adapter) on the device: trunk through TrunkUnitTrain (HIP conv forward / dgrad / wgrad +
    fused BatchNorm tail), gate through the HIP attention forward / backward."""
    from src.depracted.model.dr_spaam import SpatialDROW
    from src.utils import eval_utils
    g = golden("gradients")
    torch.manual_seed(3)
    m = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).to(DEV)
    m.train()
    loss, tb, _ = eval_utils.model_fn_obj_det(m, {"input": g["sd_x"], "target_flow_cls": g["sd_cls"],
                                                  "target_flow_reg": g["sd_reg"]})
    loss.backward()
    np.testing.assert_allclose([float(loss.detach()), tb["cls_loss"], tb["reg_loss"]], g["sd64_loss"], rtol=1e-5)
    names = [k for k, _ in m.named_parameters()]
    assert names == list(g["sd_names"])
    params = dict(m.named_parameters())
    # Yardstick: the fixture also holds the gradients of the reference's modules run in float64.  The reference's own
    # float32 gradients are up to 7e-4 (relative to the tensor's scale) away from those -- train-mode BatchNorm
    # backward cancels means through ten layers -- so the bar for the device is: its distance from the float64
    # gradients is within 3x the reference's float32 distance, or within 1e-4 of the tensor's scale.
    top = float(g["sd64_gabs"].max())
    for i, k in enumerate(names):
        gr = params[k].grad.double()
        ab64, ab32 = float(g["sd64_gabs"][i]), float(g["sd_gabs"][i])
        s64, s32 = float(g["sd64_gsum"][i]), float(g["sd_gsum"][i])
        tol_abs = 3.0 * abs(ab32 - ab64) + 1e-4 * ab64 + 1e-9 * top
        tol_sum = 3.0 * abs(s32 - s64) + 1e-4 * ab64 + 1e-9 * top
        assert abs(float(gr.abs().sum()) - ab64) <= tol_abs, (k, float(gr.abs().sum()), ab64, ab32)
        assert abs(float(gr.sum()) - s64) <= tol_sum, (k, float(gr.sum()), s64, s32)
    # whole tensors
    for key in g.files:
        if not key.startswith("sd64_grad_"):
            continue
        name = key[len("sd64_grad_"):]
        head = name.endswith("_head")
        name = name[:-5] if head else name
        match = [k for k in names if k.replace(".", "_") == name]
        assert len(match) == 1, name
        gr = params[match[0]].grad
        want64, ref32 = g[key], g["sd_grad_" + key[len("sd64_grad_"):]]
        got = (gr[:want64.shape[0]] if head else gr).detach().cpu().numpy().astype(np.float64)
        scale = float(np.abs(want64).max())
        err = float(np.abs(got - want64).max())
        ref_err = float(np.abs(ref32.astype(np.float64) - want64).max())
        assert err <= 3.0 * ref_err + 1e-4 * scale + 1e-12 * top, (match[0], err, ref_err, scale)
    bufs = dict(m.named_buffers())
    close(bufs["conv_block_1.0.1.running_mean"], g["sd_run_mean_b1"], 1e-4, "running mean")
    close(bufs["conv_block_4.1.1.running_var"], g["sd_run_var_b4"], 1e-4, "running var")


def test_prototype_training_step_equals_reference(golden):
    from src.depracted.model.prototype import Prototype, flow_loss
    g = golden("gradients")
    torch.manual_seed(7)
    m = Prototype(in_channel=1, max_displacement=5).to(DEV)
    m.train()
    pred = m(T(g["pt_s1"]), T(g["pt_s2"]))
    close(pred, g["pt_pred"], 1e-3, "prediction")
    loss, _ = flow_loss(pred, T(g["pt_tgt"]))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["pt_loss"]), rtol=2e-4)
    names = [k for k, _ in m.named_parameters()]
    assert names == list(g["pt_names"])
    for i, (k, p) in enumerate(m.named_parameters()):
        ab = float(g["pt_gabs"][i])
        gr = p.grad.double()
        assert abs(float(gr.abs().sum()) - ab) <= 5e-3 * ab + 1e-7, (k, float(gr.abs().sum()), ab)
        assert abs(float(gr.sum()) - float(g["pt_gsum"][i])) <= 5e-3 * ab + 1e-7, (k,)
    close(dict(m.named_parameters())[str(g["pt_first_name"])].grad, g["pt_grad_first"], 2e-3, "d first conv")


@pytest.mark.parametrize("hip", [True, False])
def test_box_head_training_step_equals_reference(golden, hip):
    """One training step of the box-regression head (BASELINE configs[3]; seeded weights = the reference's, dropout
    off) against the reference's own autograd: PointNet units as ConvUnitTrain nodes on the HIP kernels (float32-MFMA
    convolution forward / data gradient, one-tap split-K weight gradient, fused BatchNorm tail) and, for comparison, as
    the torch modules.  Bar as for SpatialDROW: within 3x the reference's own float32 distance from its float64 run,
    or 1e-4 of the tensor's scale."""
    from src.model.get_model import get_model
    g = golden("gradients")
    torch.manual_seed(67)
    m = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0}).to(DEV)
    m.train()
    m.backbone.hip_train = hip
    pred = m(T(g["bh_in"]))
    close(pred, g["bh_pred"], 2e-4, "prediction")
    loss = m.loss_fn(pred, T(g["bh_tgt"]))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["bh64_loss"]), rtol=2e-5)
    params = dict(m.named_parameters())
    names = list(g["bh_names"])
    assert names == [k for k, p in m.named_parameters() if p.grad is not None]
    top = float(g["bh64_gabs"].max())
    for i, k in enumerate(names):
        gr = params[k].grad.double()
        ab64, ab32 = float(g["bh64_gabs"][i]), float(g["bh_gabs"][i])
        s64, s32 = float(g["bh64_gsum"][i]), float(g["bh_gsum"][i])
        tol_abs = 3.0 * abs(ab32 - ab64) + 1e-4 * ab64 + 1e-9 * top
        tol_sum = 3.0 * abs(s32 - s64) + 1e-4 * ab64 + 1e-9 * top
        assert abs(float(gr.abs().sum()) - ab64) <= tol_abs, (k, float(gr.abs().sum()), ab64, ab32)
        assert abs(float(gr.sum()) - s64) <= tol_sum, (k, float(gr.sum()), s64, s32)
    for key in g.files:
        if not key.startswith("bh64_grad_"):
            continue
        name = key[len("bh64_grad_"):]
        match = [k for k in names if k.replace(".", "_") == name]
        assert len(match) == 1, name
        want64, ref32 = g[key], g["bh_grad_" + name]
        got = params[match[0]].grad.detach().cpu().numpy().astype(np.float64)
        scale = float(np.abs(want64).max())
        err = float(np.abs(got - want64).max())
        ref_err = float(np.abs(ref32.astype(np.float64) - want64).max())
        assert err <= 3.0 * ref_err + 1e-4 * scale + 1e-12 * top, (match[0], err, ref_err, scale)
    close(dict(m.named_buffers())["backbone.conv4.1.running_var"], g["bh_run_var_c4"], 1e-4, "running var")
