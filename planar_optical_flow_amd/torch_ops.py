"""torch.library registration of the HIP ops: ``torch.ops.pof.*``.

The C-ABI entry points are reached through ``ops.py`` (ctypes); registering them as custom operators makes them
first-class for the dispatcher: autograd formulas instead of hand-rolled ``autograd.Function`` objects, fake
(meta) kernels so that ``torch.compile`` / ``make_fx`` / FakeTensorMode can trace through a model that calls them
without running a kernel, and an ``opcheck``-able schema.  Device kernels only: a CPU tensor raises (no fallback).

    pof::band_correlation(Tensor f1, Tensor f2, int kernel_size, int max_displacement) -> Tensor
    pof::band_correlation_backward(Tensor f1, Tensor f2, Tensor g, int kernel_size, int max_displacement)
        -> (Tensor, Tensor)
    pof::spatial_attention(Tensor emb_x, Tensor emb_t, Tensor x, Tensor tmpl, float alpha, int window)
        -> (Tensor out, Tensor band, Tensor prob)
    pof::spatial_attention_backward(Tensor emb_x, Tensor emb_t, Tensor tmpl, Tensor prob, Tensor g_out,
        Tensor? g_band, float alpha, int window) -> (Tensor, Tensor, Tensor, Tensor)
    pof::cutout(Tensor scans, Tensor tab, int stride, bool centered, bool fixed, float window_width,
        float window_depth, int num_cutout_pts, float padding_val, bool area_mode, bool half_out) -> Tensor
    pof::conv3_bn_lrelu(Tensor x, Tensor wt, Tensor scale, Tensor shift, bool pool, float negative_slope) -> Tensor
    pof::rotate_flow(Tensor flow, Tensor tab, bool to_canonical) -> Tensor
    pof::bn_lrelu_pool(Tensor y, Tensor gamma, Tensor beta, Tensor(a!)? running_mean, Tensor(b!)? running_var,
        float momentum, float eps, float negative_slope, bool pool, int groups=1)
        -> (Tensor z, Tensor mean, Tensor invstd)
    pof::conv3_wgrad(Tensor x, Tensor dy) -> Tensor
    pof::linear_bias(Tensor x, Tensor weight, Tensor? bias) -> Tensor
    pof::regression_loss2(Tensor pred, Tensor target, float alpha) -> (Tensor loss, Tensor dpred)
    pof::bn_lrelu_pool_backward(Tensor y, Tensor dz, Tensor gamma, Tensor beta, Tensor mean, Tensor invstd,
        float negative_slope, bool pool, bool bias_grad, int groups=1) -> (Tensor, Tensor, Tensor, Tensor)
"""
from typing import Optional, Tuple

import torch

from . import ops

# ------------------------------------------------------------------------------------------------- A9
_DIRECT = {}


def _direct(name, **kw):
    """``torch.library.custom_op`` that also keeps the plain function: inside an autograd.Function the kernels are
    called through ``_K`` -- directly in eager mode (a registered op costs ~25 us of dispatcher / wrapper time per call,
    and the eager box-head step, ~40 such calls for 0.7 ms of kernels, is host-bound), through the registered op with
    its fake kernel under torch.compile."""
    def wrap(fn):
        _DIRECT[name] = fn
        return torch.library.custom_op("pof::" + name, **kw)(fn)
    return wrap


class _Kernels:
    def __getattr__(self, name):
        if torch.compiler.is_compiling():
            return getattr(torch.ops.pof, name)
        return _DIRECT[name]


_K = _Kernels()


@torch.library.custom_op("pof::band_correlation", mutates_args=(), device_types="cuda")
def band_correlation(f1: torch.Tensor, f2: torch.Tensor, kernel_size: int, max_displacement: int) -> torch.Tensor:
    return ops.band_correlation(f1.contiguous(), f2.contiguous(), kernel_size, max_displacement)


@band_correlation.register_fake
def _(f1, f2, kernel_size, max_displacement):
    B, _, n = f1.shape
    return f1.new_empty((B, 2 * max_displacement + 1, n), dtype=torch.float32)


@torch.library.custom_op("pof::band_correlation_backward", mutates_args=(), device_types="cuda")
def band_correlation_backward(f1: torch.Tensor, f2: torch.Tensor, g: torch.Tensor, kernel_size: int,
                              max_displacement: int) -> Tuple[torch.Tensor, torch.Tensor]:
    d1, d2 = ops.band_correlation_backward(f1.contiguous(), f2.contiguous(), g.contiguous().float(), kernel_size,
                                           max_displacement)
    return d1, d2


@band_correlation_backward.register_fake
def _(f1, f2, g, kernel_size, max_displacement):
    return torch.empty_like(f1, dtype=torch.float32), torch.empty_like(f2, dtype=torch.float32)


def _corr_setup(ctx, inputs, output):
    f1, f2, ctx.kernel_size, ctx.max_displacement = inputs
    ctx.save_for_backward(f1, f2)


def _corr_backward(ctx, grad):
    f1, f2 = ctx.saved_tensors
    d1, d2 = torch.ops.pof.band_correlation_backward(f1, f2, grad, ctx.kernel_size, ctx.max_displacement)
    return d1, d2, None, None


band_correlation.register_autograd(_corr_backward, setup_context=_corr_setup)


# ------------------------------------------------------------------------------------------------- A10
@torch.library.custom_op("pof::spatial_attention", mutates_args=(), device_types="cuda")
def spatial_attention(emb_x: torch.Tensor, emb_t: torch.Tensor, x: torch.Tensor, tmpl: torch.Tensor, alpha: float,
                      window: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    out, band, prob = ops.spatial_attention(emb_x.contiguous(), emb_t.contiguous(), x.contiguous(), tmpl.contiguous(),
                                            alpha, window)
    return out, band, prob


@spatial_attention.register_fake
def _(emb_x, emb_t, x, tmpl, alpha, window):
    B, N, _ = emb_x.shape
    w = 2 * int(window / 2) + 1
    return (torch.empty_like(x), emb_x.new_empty((B, N, w), dtype=torch.float32),
            emb_x.new_empty((B, N, w), dtype=torch.float32))


@torch.library.custom_op("pof::spatial_attention_backward", mutates_args=(), device_types="cuda")
def spatial_attention_backward(emb_x: torch.Tensor, emb_t: torch.Tensor, tmpl: torch.Tensor, prob: torch.Tensor,
                               g_out: torch.Tensor, g_band: Optional[torch.Tensor], alpha: float, window: int
                               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    gb = None if g_band is None else g_band.contiguous().float()
    dex, det, dx, dt = ops.spatial_attention_backward(emb_x, emb_t, tmpl, prob, g_out.contiguous().float(), gb, alpha,
                                                      window)
    return dex, det, dx, dt


@spatial_attention_backward.register_fake
def _(emb_x, emb_t, tmpl, prob, g_out, g_band, alpha, window):
    return (torch.empty_like(emb_x), torch.empty_like(emb_t), torch.empty_like(tmpl, dtype=torch.float32),
            torch.empty_like(tmpl, dtype=torch.float32))


def _attn_setup(ctx, inputs, output):
    emb_x, emb_t, _, tmpl, ctx.alpha, ctx.window = inputs
    ctx.save_for_backward(emb_x, emb_t, tmpl, output[2])
    ctx.set_materialize_grads(False)


def _attn_backward(ctx, g_out, g_band, g_prob):
    emb_x, emb_t, tmpl, prob = ctx.saved_tensors
    if g_prob is not None:
        raise RuntimeError("pof::spatial_attention: the softmax weights are an auxiliary output without a gradient")
    if g_out is None:
        g_out = torch.zeros_like(tmpl, dtype=torch.float32)
    dex, det, dx, dt = torch.ops.pof.spatial_attention_backward(emb_x, emb_t, tmpl, prob, g_out, g_band, ctx.alpha,
                                                                ctx.window)
    return dex, det, dx, dt, None, None


spatial_attention.register_autograd(_attn_backward, setup_context=_attn_setup)


# ------------------------------------------------------------------------------------------------- A8
@torch.library.custom_op("pof::cutout", mutates_args=(), device_types="cuda")
def cutout(scans: torch.Tensor, tab: torch.Tensor, stride: int, centered: bool, fixed: bool, window_width: float,
           window_depth: float, num_cutout_pts: int, padding_val: float, area_mode: bool, half_out: bool
           ) -> torch.Tensor:
    return ops.cutout(scans.contiguous(), tab, stride=stride, centered=centered, fixed=fixed,
                      window_width=window_width, window_depth=window_depth, num_cutout_pts=num_cutout_pts,
                      padding_val=padding_val, area_mode=area_mode,
                      out_dtype=torch.float16 if half_out else torch.float32)


@cutout.register_fake
def _(scans, tab, stride, centered, fixed, window_width, window_depth, num_cutout_pts, padding_val, area_mode, half_out):
    B, T, N = scans.shape
    return scans.new_empty((B, (N + stride - 1) // stride, T, num_cutout_pts),
                           dtype=torch.float16 if half_out else torch.float32)


# ------------------------------------------------------------------------------------------------- N2
@_direct("conv3_bn_lrelu", mutates_args=(), device_types="cuda")
def conv3_bn_lrelu(x: torch.Tensor, wt: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, pool: bool,
                   negative_slope: float) -> torch.Tensor:
    return ops.conv3_bn_lrelu(x.contiguous(), wt, scale, shift, pool=pool, negative_slope=negative_slope)


@conv3_bn_lrelu.register_fake
def _(x, wt, scale, shift, pool, negative_slope):
    S, _, L = x.shape
    return x.new_empty((S, wt.shape[2], L // 2 if pool else L))


@_direct("conv1d_bn_lrelu", mutates_args=(), device_types="cuda")
def conv1d_bn_lrelu(x: torch.Tensor, wt: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, stride: int,
                    negative_slope: float) -> torch.Tensor:
    return ops.conv1d_bn_lrelu(x.contiguous(), wt, scale, shift, stride=stride, negative_slope=negative_slope)


@conv1d_bn_lrelu.register_fake
def _(x, wt, scale, shift, stride, negative_slope):
    S, _, L = x.shape
    return x.new_empty((S, wt.shape[2], (L + stride - 1) // stride))


@_direct("linear_bias", mutates_args=(), device_types="cuda")
def linear_bias(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return ops.linear_bias(x.contiguous(), weight.contiguous(), bias)


@linear_bias.register_fake
def _(x, weight, bias):
    return x.new_empty((x.shape[0], weight.shape[0]))


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, g):
    # the two backward GEMMs (outputs B x K and N x K) are shapes the BLAS library runs well
    x, weight = ctx.saved_tensors
    dx = g.mm(weight) if ctx.needs_input_grad[0] else None
    dw = g.t().mm(x) if ctx.needs_input_grad[1] else None
    db = g.sum(dim=0) if ctx.has_bias and ctx.needs_input_grad[2] else None
    return dx, dw, db


linear_bias.register_autograd(_linear_backward, setup_context=_linear_setup)


@_direct("regression_loss2", mutates_args=(), device_types="cuda")
def regression_loss2(pred: torch.Tensor, target: torch.Tensor, alpha: float) -> Tuple[torch.Tensor, torch.Tensor]:
    loss, dpred = ops.regression_loss2(pred.contiguous(), target.contiguous(), alpha)
    return loss, dpred


@regression_loss2.register_fake
def _(pred, target, alpha):
    return pred.new_empty(()), torch.empty_like(pred)


def _loss2_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _loss2_backward(ctx, g_loss, g_dpred):
    (dpred,) = ctx.saved_tensors
    return dpred * g_loss, None, None


regression_loss2.register_autograd(_loss2_backward, setup_context=_loss2_setup)


def linear_small(x, linear):
    """``linear(x)`` for a ``torch.nn.Linear`` on a small batch (the box head's dense layers): the forward on
    ``pof::linear_bias`` when the shape allows (2-D float32 input on the device, in_features a multiple of 4)."""
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and linear.in_features % 4 == 0 \
            and linear.weight.dtype == torch.float32:
        if torch.compiler.is_compiling():
            return torch.ops.pof.linear_bias(x, linear.weight, linear.bias)
        return _LinearSmall.apply(x, linear.weight, linear.bias)
    return linear(x)


class _LinearSmall(torch.autograd.Function):
    """Eager form of pof::linear_bias with its autograd formula (see _direct)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _DIRECT["linear_bias"](x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        return _linear_backward(ctx, g)


class _RegressionLoss2(torch.autograd.Function):
    """Eager form of pof::regression_loss2 with its autograd formula."""

    @staticmethod
    def forward(ctx, pred, target, alpha):
        loss, dpred = _DIRECT["regression_loss2"](pred, target, alpha)
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (dpred,) = ctx.saved_tensors
        return dpred * g_loss, None, None


def regression_loss2_fused(pred, target, alpha):
    """The box-regression loss (src/model/box_regression.py:52-67) and its gradient in one launch."""
    if torch.compiler.is_compiling():
        return torch.ops.pof.regression_loss2(pred, target, float(alpha))[0]
    return _RegressionLoss2.apply(pred, target, float(alpha))


# ------------------------------------------------------------------------------------------------- N2 (training)
@_direct("bn_lrelu_pool", mutates_args=("running_mean", "running_var"), device_types="cuda")
def bn_lrelu_pool(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: Optional[torch.Tensor],
                  running_var: Optional[torch.Tensor], momentum: float, eps: float, negative_slope: float,
                  pool: bool, groups: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    z, mean, invstd = ops.bn_lrelu_pool_forward(y.contiguous(), gamma.contiguous(), beta.contiguous(), running_mean,
                                                running_var, momentum, eps, negative_slope, pool, groups=groups)
    return z, mean, invstd


@bn_lrelu_pool.register_fake
def _(y, gamma, beta, running_mean, running_var, momentum, eps, negative_slope, pool, groups=1):
    S, C, L = y.shape
    return (y.new_empty((S, C, L // 2 if pool else L)), y.new_empty((groups * C,)), y.new_empty((groups * C,)))


@_direct("bn_lrelu_pool_backward", mutates_args=(), device_types="cuda")
def bn_lrelu_pool_backward(y: torch.Tensor, dz: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                           mean: torch.Tensor, invstd: torch.Tensor, negative_slope: float, pool: bool,
                           bias_grad: bool, groups: int = 1
                           ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    res = ops.bn_lrelu_pool_backward(y, dz.contiguous().float(), gamma.contiguous(), beta.contiguous(), mean, invstd,
                                     negative_slope, pool, bias_grad=bias_grad, groups=groups)
    if bias_grad:
        return res
    return res[0], res[1], res[2], gamma.new_empty((0,))


@bn_lrelu_pool_backward.register_fake
def _(y, dz, gamma, beta, mean, invstd, negative_slope, pool, bias_grad, groups=1):
    return (torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(beta),
            gamma.new_empty((gamma.shape[0] if bias_grad else 0,)))


@_direct("bn_lrelu_rowmax", mutates_args=("running_mean", "running_var"), device_types="cuda")
def bn_lrelu_rowmax(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: Optional[torch.Tensor],
                    running_var: Optional[torch.Tensor], momentum: float, eps: float, negative_slope: float,
                    groups: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """max over the row of leaky_relu(batch_norm_train(y)): y [S,C,L] -> z [S,C] (the PointNet's max over points fused
    into the tail's apply pass: the activation is never written)."""
    z, mean, invstd = ops.bn_lrelu_pool_forward(y.contiguous(), gamma.contiguous(), beta.contiguous(), running_mean,
                                                running_var, momentum, eps, negative_slope, 2, groups=groups)
    return z, mean, invstd


@bn_lrelu_rowmax.register_fake
def _(y, gamma, beta, running_mean, running_var, momentum, eps, negative_slope, groups=1):
    S, C, L = y.shape
    return (y.new_empty((S, C)), y.new_empty((groups * C,)), y.new_empty((groups * C,)))


@_direct("bn_lrelu_rowmax_backward", mutates_args=(), device_types="cuda")
def bn_lrelu_rowmax_backward(y: torch.Tensor, dz: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                             mean: torch.Tensor, invstd: torch.Tensor, negative_slope: float, bias_grad: bool,
                             groups: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    res = ops.bn_lrelu_pool_backward(y, dz.contiguous().float(), gamma.contiguous(), beta.contiguous(), mean, invstd,
                                     negative_slope, 2, bias_grad=bias_grad, groups=groups)
    if bias_grad:
        return res
    return res[0], res[1], res[2], gamma.new_empty((0,))


@bn_lrelu_rowmax_backward.register_fake
def _(y, dz, gamma, beta, mean, invstd, negative_slope, bias_grad, groups=1):
    return (torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(beta),
            gamma.new_empty((gamma.shape[0] if bias_grad else 0,)))


class BnLreluPool(torch.autograd.Function):
    """Autograd wrapper of pof::bn_lrelu_pool.  The operator updates the running statistics in place, and the
    dispatcher only accepts autograd formulas for functional operators -- hence a Function around the two ops
    (both still visible to torch.compile through their fake kernels)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, momentum, eps, negative_slope, pool):
        z, mean, invstd = _K.bn_lrelu_pool(y, gamma, beta, running_mean, running_var, momentum, eps,
                                                      negative_slope, pool)
        ctx.save_for_backward(y, gamma, beta, mean, invstd)
        ctx.negative_slope, ctx.pool = negative_slope, pool
        return z

    @staticmethod
    def backward(ctx, g_z):
        y, gamma, beta, mean, invstd = ctx.saved_tensors
        dy, dgamma, dbeta, _ = _K.bn_lrelu_pool_backward(y, g_z, gamma, beta, mean, invstd,
                                                                    ctx.negative_slope, ctx.pool, False)
        return dy, dgamma, dbeta, None, None, None, None, None, None


def _bn_train_args(bn, groups=1, shape=None):
    """(running_mean, running_var, momentum, eps) of a BatchNorm in training mode, batch counter advanced as the
    module itself does -- by `groups` when that many batches go through it in one grouped call.  `shape` = the
    [S, C, L] of the input: one value per channel and group is refused as torch.nn.functional.batch_norm does."""
    if shape is not None and (shape[0] // groups) * shape[2] <= 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(shape),))
    momentum = bn.momentum
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(groups)
        if momentum is None:    # cumulative moving average
            if groups != 1:
                raise ValueError("grouped statistics need a fixed BatchNorm momentum")
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return rm, rv, float(momentum if momentum is not None else 0.0), float(bn.eps)


def bn_lrelu_pool_train(y, bn, negative_slope=0.1, pool=False):
    """Training-mode tail of a trunk unit on the fused kernels: z = max_pool1d?(leaky_relu(bn(y))) for a
    ``torch.nn.BatchNorm1d`` in training mode, with its running statistics and batch counter updated as the
    module itself would."""
    rm, rv, momentum, eps = _bn_train_args(bn, 1, y.shape)
    return BnLreluPool.apply(y, bn.weight, bn.bias, rm, rv, momentum, eps, float(negative_slope), bool(pool))


@_direct("conv3_wgrad", mutates_args=(), device_types="cuda")
def conv3_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    return ops.conv3_wgrad(x.contiguous(), dy.contiguous())


@conv3_wgrad.register_fake
def _(x, dy):
    return x.new_empty((dy.shape[1], x.shape[1], 3))


@_direct("conv1_wgrad", mutates_args=(), device_types="cuda")
def conv1_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    return ops.conv3_wgrad(x.contiguous(), dy.contiguous(), kernel_size=1)


@conv1_wgrad.register_fake
def _(x, dy):
    return x.new_empty((dy.shape[1], x.shape[1], 1))


def _wgrad_supported(S, ci, co, L, k):
    """Shapes the split-K weight-gradient kernel stages in LDS -- the arithmetic of make_wgrad() in csrc/conv_wgrad.hip
    restated in plain Python, so that torch.compile can evaluate it on symbolic sizes (a ctypes query of the C ABI
    cannot be traced inside an autograd.Function).  tests/test_abi.py holds the two against each other over a sweep."""
    if S < 1 or ci < 1 or co < 1 or L < 1 or L > 256 or k not in (1, 3):
        return False
    lh = (L + 1) // 2
    mt = 128 if co > 64 else 64
    per_seq = (mt * (2 * lh + 3) + 64 * (2 * lh + 5)) * 4
    vec = 4 if L % 4 == 0 else (2 if L % 2 == 0 else 1)
    per_row = (L + vec - 1) // vec
    tpr_log2 = 0
    while (1 << tpr_log2) < per_row:
        tpr_log2 += 1
    if tpr_log2 > 8:
        return False
    stage = {4: 32, 2: 24, 1: 16}[vec]
    max_g = (stage // vec) * (256 >> tpr_log2) // 128
    if max_g < 1:
        return False
    g = max(1, min(8, 48 * 1024 // per_seq, max_g))
    if g * per_seq > 64 * 1024:
        return False
    tiles = ((co + mt - 1) // mt) * ((ci + 63) // 64)
    return tiles <= 65535


def _pointwise_weight_grad(x, dy, weight):
    """dL/dweight of the k = 1 convolution, dw[co][ci] = sum dy[s][co][l] x[s][ci][l]: the split-K MFMA kernel's
    one-tap form; rows too long for its LDS stage are cut into chunks (no halo: positions do not interact)."""
    S, ci, L = (int(v) for v in x.shape)       # (int: a symbolic size under torch.compile is specialised here)
    co = int(dy.shape[1])
    if _wgrad_supported(S, ci, co, L, 1):
        return _K.conv1_wgrad(x, dy)
    for lc in (64, 56, 48, 40, 32, 24, 16):
        nc = -(-L // lc)
        if _wgrad_supported(S * nc, ci, co, lc, 1):
            pad = nc * lc - L
            xc = torch.nn.functional.pad(x, (0, pad)).reshape(S, ci, nc, lc).permute(0, 2, 1, 3).reshape(S * nc, ci, lc)
            dyc = torch.nn.functional.pad(dy, (0, pad)).reshape(S, co, nc, lc).permute(0, 2, 1, 3).reshape(S * nc, co, lc)
            return _K.conv1_wgrad(xc.contiguous(), dyc.contiguous())
    return torch.ops.aten.convolution_backward(dy, x, weight, None, [1], [0], [1], False, [0], 1,
                                               [False, True, False])[1]


def _weight_grad(x, dy, weight):
    """dL/dweight of the k = 3, stride 1 convolution: the split-K MFMA kernel.  It stages whole sequences in LDS
    (DR-SPAAM's cutouts: L <= 56); longer sequences (the Prototype's 57 ... 450 points) are cut into chunks with a
    one-element halo: chunk c holds x[c Lc - 1 .. (c + 1) Lc] and dy[c Lc .. (c + 1) Lc) between two zeros, so that
    sum_j dy_c[j] x_c[j + t - 1] over the chunks is the gradient of the whole sequence.  The library's kernel is
    left for shapes neither form takes."""
    S, ci, L = (int(v) for v in x.shape)
    co = int(dy.shape[1])
    if _wgrad_supported(S, ci, co, L, 3):
        return _K.conv3_wgrad(x, dy)
    for lc in (62, 54, 46, 38, 30, 22, 14):
        nc = -(-L // lc)
        if _wgrad_supported(S * nc, ci, co, lc + 2, 3):
            xp = torch.nn.functional.pad(x, (1, nc * lc - L + 1))
            xc = xp.unfold(2, lc + 2, lc).permute(0, 2, 1, 3).reshape(S * nc, ci, lc + 2)
            dyc = torch.nn.functional.pad(torch.nn.functional.pad(dy, (0, nc * lc - L)).reshape(S, co, nc, lc), (1, 1))
            dyc = dyc.permute(0, 2, 1, 3).reshape(S * nc, co, lc + 2)
            return _K.conv3_wgrad(xc.contiguous(), dyc.contiguous())
    return torch.ops.aten.convolution_backward(dy, x, weight, None, [1], [1], [1], False, [0], 1,
                                               [False, True, False])[1]


_CONSTS = {}


def _const(n, value, like):
    """A cached [n] float32 vector of ones / zeros on ``like``'s device (unit scale, zero shift of the plain
    convolution) -- not a fill kernel per call."""
    if torch.compiler.is_compiling():
        return torch.full((int(n),), float(value), dtype=torch.float32, device=like.device)
    key = (like.device, int(n), float(value))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full((int(n),), float(value), dtype=torch.float32, device=like.device)
    return t


_LAYOUT_SCOPE = None


class weight_layout_scope:
    """Within this scope (one forward pass of a model) the kernel layouts of a convolution weight are built once
    per weight instead of once per call -- the first two trunk blocks run once per scan of the window with the
    same weights.  The cache dies with the scope, so it can never outlive a parameter update."""

    def __enter__(self):
        global _LAYOUT_SCOPE
        self._outer = _LAYOUT_SCOPE
        if _LAYOUT_SCOPE is None:
            _LAYOUT_SCOPE = {}
        return self

    def __exit__(self, *exc):
        global _LAYOUT_SCOPE
        _LAYOUT_SCOPE = self._outer
        return False


def _weight_layouts(weight, need_dgrad):
    """([3][Ci][Co] forward layout, [3][Co][Ci] tap-reversed layout of the data gradient or None)."""
    scope = None if torch.compiler.is_compiling() else _LAYOUT_SCOPE
    entry = scope.get(id(weight)) if scope is not None else None
    if entry is None:
        entry = [weight.detach().permute(2, 1, 0).contiguous(), None, weight]   # keeps `weight` alive: id stays unique
        if scope is not None:
            scope[id(weight)] = entry
    if need_dgrad and entry[1] is None:
        entry[1] = weight.detach().flip(2).permute(2, 0, 1).contiguous()
    return entry[0], entry[1]


class Conv3Train(torch.autograd.Function):
    """Conv1d(kernel_size=3, padding=1) of a trunk unit in training: forward and the data gradient on the
    float32-MFMA implicit-GEMM kernel of the inference trunk (``pof::conv3_bn_lrelu`` with unit scale, the bias
    as shift and slope 1 = no activation; the data gradient is the same convolution of dy with the taps
    reversed and the channel roles swapped).  Both read and write [S][C][L] as it lies -- the library's NHWC
    kernels transpose every operand first (13 % of a training step).  The weight gradient is the split-K MFMA
    kernel ``pof::conv3_wgrad``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        co = weight.shape[0]
        wt, wd = _weight_layouts(weight, ctx.needs_input_grad[0])
        shift = bias.detach() if bias is not None else _const(co, 0.0, weight)
        y = _K.conv3_bn_lrelu(x, wt, _const(co, 1.0, weight), shift, False, 1.0)
        ctx.save_for_backward(x, weight, wd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, wd = ctx.saved_tensors                                       # wd [3][Co][Ci]
        co, ci, _ = weight.shape
        gy = gy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _K.conv3_bn_lrelu(gy, wd, _const(ci, 1.0, weight), _const(ci, 0.0, weight), False, 1.0)
        dw = _weight_grad(x, gy, weight) if ctx.needs_input_grad[1] else None
        db = gy.sum(dim=(0, 2)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def conv3_train(x, conv):
    """``conv(x)`` for a ``torch.nn.Conv1d(kernel_size=3, padding=1)`` on the HIP forward / data-gradient kernels."""
    return Conv3Train.apply(x.contiguous(), conv.weight, conv.bias)


class TrunkUnitTrain(torch.autograd.Function):
    """One whole trunk unit in training -- Conv1d(3, pad 1) -> BatchNorm1d(train) -> LeakyReLU [-> max_pool1d(2)] --
    as one autograd node: Conv3Train's kernels for the convolution and its data gradient, pof::bn_lrelu_pool for
    the tail.  Being one node lets the tail's dgrad pass hand the convolution its bias gradient (the sum of dy it
    is writing anyway) instead of a reduction pass of its own; only the input x and the convolution output y are
    kept for the backward pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, negative_slope, pool,
                groups):
        co = weight.shape[0]
        wt, wd = _weight_layouts(weight, ctx.needs_input_grad[0])
        shift = bias.detach() if bias is not None else _const(co, 0.0, weight)
        y = _K.conv3_bn_lrelu(x, wt, _const(co, 1.0, weight), shift, False, 1.0)
        z, mean, invstd = _K.bn_lrelu_pool(y, gamma, beta, running_mean, running_var, momentum, eps,
                                                      negative_slope, pool, groups)
        ctx.save_for_backward(x, weight, y, gamma, beta, mean, invstd, wd)
        ctx.has_bias, ctx.negative_slope, ctx.pool, ctx.groups = bias is not None, negative_slope, pool, groups
        return z

    @staticmethod
    def backward(ctx, g_z):
        x, weight, y, gamma, beta, mean, invstd, wd = ctx.saved_tensors
        co, ci, _ = weight.shape
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        dy, dgamma, dbeta, db = _K.bn_lrelu_pool_backward(y, g_z, gamma, beta, mean, invstd,
                                                                     ctx.negative_slope, ctx.pool, want_db, ctx.groups)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _K.conv3_bn_lrelu(dy, wd, _const(ci, 1.0, weight), _const(ci, 0.0, weight), False, 1.0)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(x, dy, weight)
        return dx, dw, (db if want_db else None), dgamma, dbeta, None, None, None, None, None, None, None


def trunk_unit_train(x, conv, bn, negative_slope=0.1, pool=False, groups=1):
    """``max_pool1d?(leaky_relu(bn(conv(x))))`` for a trunk unit's modules in training mode (see TrunkUnitTrain).
    ``groups`` > 1: x holds that many equally long batches one after the other (the scans of a window); each is
    normalised with its own batch statistics and the module's running statistics see them in order -- the result
    of sending the batches through the unit one by one, in one launch per pass."""
    rm, rv, momentum, eps = _bn_train_args(bn, groups, (x.shape[0], conv.out_channels, x.shape[2]))
    return TrunkUnitTrain.apply(x.contiguous(), conv.weight, conv.bias, bn.weight, bn.bias, rm, rv, momentum, eps,
                                float(negative_slope), bool(pool), int(groups))


class ConvUnitTrain(torch.autograd.Function):
    """A unit of the Prototype flow network in training -- Conv1d(kernel 1 | 3, padding kernel // 2, stride 1 | 2) ->
    BatchNorm1d(train) -> LeakyReLU -- as one autograd node on the HIP kernels (round 3).  Forward: the float32-MFMA
    convolution ``pof_conv1d_bn_lrelu`` (unit scale, bias as shift, slope 1) and the fused BatchNorm tail.  Backward:
    the tail's fused backward, then

      kernel 3, stride 1   data gradient = the same convolution of dy with the taps reversed and the channel roles
                           swapped; weight gradient = ``pof::conv3_wgrad``                        (as TrunkUnitTrain)
      kernel 1             data gradient = the k = 1 convolution of dy with the transposed weight; weight gradient =
                           ``pof::conv1_wgrad`` (the split-K kernel's one-tap form)
      kernel 3, stride 2   y[l] = W0 x[2l-1] + W1 x[2l] + W2 x[2l+1], so with x_even[l] = x[2l], x_odd[l] = x[2l+1]:
                           dx_even[l] = W1' dy[l]                 (a k = 1 convolution),
                           dx_odd[l]  = W2' dy[l] + W0' dy[l+1]   (a k = 3 convolution with taps {0, W2', W0'}),
                           dW1 = centre tap of wgrad(x_even, dy), (dW0, dW2) = taps 0, 1 of wgrad(x_odd, dy)
                           -- every pass on the stride-1 kernels that exist, on de-interleaved copies.
    Only x and the convolution output y are kept for the backward pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, negative_slope, stride,
                groups, rowmax=False):
        co, ci, k = weight.shape
        wt = weight.detach().permute(2, 1, 0).contiguous()
        shift = bias.detach() if bias is not None else _const(co, 0.0, weight)
        y = _K.conv1d_bn_lrelu(x, wt, _const(co, 1.0, weight), shift, stride, 1.0)
        if rowmax:      # z [S, Co] = max over the positions, taken inside the tail's apply pass
            z, mean, invstd = _K.bn_lrelu_rowmax(y, gamma, beta, running_mean, running_var, momentum, eps,
                                                            negative_slope, groups)
        else:
            z, mean, invstd = _K.bn_lrelu_pool(y, gamma, beta, running_mean, running_var, momentum, eps,
                                                          negative_slope, False, groups)
        ctx.save_for_backward(x, weight, y, gamma, beta, mean, invstd)
        ctx.has_bias, ctx.negative_slope, ctx.stride, ctx.groups = bias is not None, negative_slope, stride, groups
        ctx.rowmax = rowmax
        return z

    @staticmethod
    def backward(ctx, g_z):
        x, weight, y, gamma, beta, mean, invstd = ctx.saved_tensors
        co, ci, k = weight.shape
        stride = ctx.stride
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.rowmax:
            dy, dgamma, dbeta, db = _K.bn_lrelu_rowmax_backward(y, g_z.contiguous(), gamma, beta, mean, invstd,
                                                                           ctx.negative_slope, want_db, ctx.groups)
        else:
            dy, dgamma, dbeta, db = _K.bn_lrelu_pool_backward(y, g_z.contiguous(), gamma, beta, mean, invstd,
                                                                         ctx.negative_slope, False, want_db, ctx.groups)
        one, zero = _const(ci, 1.0, weight), _const(ci, 0.0, weight)
        w = weight.detach()
        lin, lo = x.shape[2], dy.shape[2]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if k == 3 and stride == 1:
                dx = _K.conv1d_bn_lrelu(dy, w.flip(2).permute(2, 0, 1).contiguous(), one, zero, 1, 1.0)
            elif k == 1:
                dx = _K.conv1d_bn_lrelu(dy, w.permute(2, 0, 1).contiguous(), one, zero, 1, 1.0)
            else:
                dx = torch.empty_like(x)
                w1 = w[:, :, 1:2].permute(2, 0, 1).contiguous()                                  # [1, Co, Ci]
                dx[:, :, 0::2] = _K.conv1d_bn_lrelu(dy, w1, one, zero, 1, 1.0)     # even positions: lo of them
                if lin > 1:
                    wo = torch.stack((torch.zeros_like(w[:, :, 0]), w[:, :, 2], w[:, :, 0]), dim=0)   # taps dy[l-1], dy[l], dy[l+1]
                    dxo = _K.conv1d_bn_lrelu(dy, wo.contiguous(), one, zero, 1, 1.0)
                    dx[:, :, 1::2] = dxo[:, :, :lin // 2]
        if ctx.needs_input_grad[1]:
            if k == 3 and stride == 1:
                dw = _weight_grad(x, dy, weight)
            elif k == 1:
                dw = _pointwise_weight_grad(x, dy, weight)
            else:
                xe = x[:, :, 0::2].contiguous()                                                  # [S, Ci, lo]
                xo = x.new_zeros((x.shape[0], ci, lo))
                xo[:, :, :lin // 2] = x[:, :, 1::2]
                w3 = weight.new_empty((co, ci, 3))
                ge, go = _weight_grad(xe, dy, w3), _weight_grad(xo, dy, w3)
                dw = torch.stack((go[:, :, 0], ge[:, :, 1], go[:, :, 1]), dim=2)
        return dx, dw, (db if want_db else None), dgamma, dbeta, None, None, None, None, None, None, None, None


def conv_unit_train(x, conv, bn, negative_slope, groups=1, rowmax=False):
    """``leaky_relu(bn(conv(x)))`` for the modules of a Prototype unit in training mode (see ConvUnitTrain).  ``groups`` > 1:
    x holds that many equally long batches one after the other (the two scans of a pair), each normalised with its own
    batch statistics, the running statistics updated once per group in order -- what sending them through the unit
    one by one does (prototype.py:70-80), in one launch per pass.  ``rowmax``: the result is the maximum over the
    positions, [S, Co] (output length a power of two >= 4) -- the PointNet's last unit and its max over points."""
    lo = (x.shape[2] + conv.stride[0] - 1) // conv.stride[0]
    rm, rv, momentum, eps = _bn_train_args(bn, groups, (x.shape[0], conv.out_channels, lo))
    return ConvUnitTrain.apply(x.contiguous().float(), conv.weight, conv.bias, bn.weight, bn.bias, rm, rv, momentum, eps,
                               float(negative_slope), int(conv.stride[0]), int(groups), bool(rowmax))


def conv_unit_train_supported(conv, length, channels_ok=True):
    """Shapes ConvUnitTrain takes: kernel 3 (stride 1 | 2) or 1, and a BatchNorm tail within the fused kernels' range
    (output length <= 256, Co x length a multiple of 4)."""
    k, st = conv.kernel_size[0], conv.stride[0]
    if not ((k == 3 and st in (1, 2)) or (k == 1 and st == 1)) or conv.padding[0] != k // 2:
        return False
    lo = (length + st - 1) // st
    return lo <= 256 and (conv.out_channels * lo) % 4 == 0


class Conv1dTrain(torch.autograd.Function):
    """A plain Conv1d(kernel 1, stride 1) in training on the HIP kernels (the Prototype's point-wise head, whose 450-point
    BatchNorm is outside the fused tail's range and stays a torch module): forward and data gradient are k = 1
    convolutions on the float32-MFMA kernel, the weight gradient is ``pof::conv1_wgrad``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        co, ci, k = weight.shape
        shift = bias.detach() if bias is not None else _const(co, 0.0, weight)
        y = _K.conv1d_bn_lrelu(x, weight.detach().permute(2, 1, 0).contiguous(), _const(co, 1.0, weight), shift,
                                          1, 1.0)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        co, ci, _ = weight.shape
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _K.conv1d_bn_lrelu(dy, weight.detach().permute(2, 0, 1).contiguous(), _const(ci, 1.0, weight),
                                               _const(ci, 0.0, weight), 1, 1.0)
        if ctx.needs_input_grad[1]:
            dw = _pointwise_weight_grad(x, dy, weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 2))
        return dx, dw, db


# ------------------------------------------------------------------------------------------------- A4
@torch.library.custom_op("pof::rotate_flow", mutates_args=(), device_types="cuda")
def rotate_flow(flow: torch.Tensor, tab: torch.Tensor, to_canonical: bool) -> torch.Tensor:
    return ops.rotate_flow(flow.contiguous(), tab, to_canonical)


@rotate_flow.register_fake
def _(flow, tab, to_canonical):
    return torch.empty_like(flow)


def _rot_setup(ctx, inputs, output):
    _, tab, ctx.to_canonical = inputs
    ctx.save_for_backward(tab)


def _rot_backward(ctx, g):
    (tab,) = ctx.saved_tensors
    # the per-point rotation is orthogonal: the gradient is the inverse rotation of the incoming gradient
    return torch.ops.pof.rotate_flow(g, tab, not ctx.to_canonical), None, None


rotate_flow.register_autograd(_rot_backward, setup_context=_rot_setup)
