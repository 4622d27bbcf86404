"""HIP-backed ``_SpatialAttention`` of the reference's
``src/depracted/model/dr_spaam.py`` (:124-217) and its masked ``flow_loss``
(:22-27).

The module keeps the reference's parameters (``conv`` = Conv1d(n_channel, 128,
kernel_size=n_pts) + BatchNorm1d + LeakyReLU(0.1); same state-dict keys).  The
embedding stays a dense MIOpen/rocBLAS op; the windowed similarity, the masked
softmax and the weighted template merge -- a full N x N GEMM pair in the
reference -- run as two banded HIP launches (pof_spatial_attention).
Forward and backward are HIP (the backward reuses the register-ring merge kernel in
transposed form), so SpatialDROW trains through the gate.
"""
import torch
import torch.nn as nn

from planar_optical_flow_amd import ops


def flow_loss(pred, target, mask=None):
    err = torch.norm(pred - target, dim=-1)
    return torch.mean(err[mask == 1.0]) if mask is not None else torch.mean(err)


class _WindowedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb_x, emb_t, x, tmpl, alpha, window):
        ex, et = emb_x.contiguous().float(), emb_t.contiguous().float()
        xx, tt = x.contiguous().float(), tmpl.contiguous().float()
        out, band, prob = ops.spatial_attention(ex, et, xx, tt, alpha, window)
        ctx.save_for_backward(ex, et, tt, prob)
        ctx.cfg = (alpha, window)
        return out, band

    @staticmethod
    def backward(ctx, g_out, g_band):
        ex, et, tt, prob = ctx.saved_tensors
        alpha, window = ctx.cfg
        gb = None if g_band is None else g_band.contiguous().float()
        dex, det, dx, dt = ops.spatial_attention_backward(ex, et, tt, prob, g_out.contiguous().float(), gb,
                                                          alpha, window)
        return dex, det, dx.view_as(g_out), dt.view_as(g_out), None, None


class _SpatialAttention(nn.Module):
    def __init__(self, n_pts, n_channel, alpha=0.5, window_size=7):
        super().__init__()
        self._alpha, self._window_size = alpha, window_size
        self.conv = nn.Sequential(nn.Conv1d(n_channel, 128, kernel_size=n_pts, padding=0), nn.BatchNorm1d(128),
                                  nn.LeakyReLU(negative_slope=0.1, inplace=True))
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, a=0.1, nonlinearity="leaky_relu")
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x, x_template):
        """x, x_template [B, n_cutout, n_channel, n_pts] -> (fused template of the
        same shape, pre-softmax window similarities [B, n_cutout, window])."""
        B, N, C, P = x.shape
        emb_x = self.conv(x.reshape(B * N, C, P)).view(B, N, 128)
        emb_t = self.conv(x_template.reshape(B * N, C, P)).view(B, N, 128)
        out, band = _WindowedAttention.apply(emb_x, emb_t, x, x_template, self._alpha, self._window_size)
        return out, band
