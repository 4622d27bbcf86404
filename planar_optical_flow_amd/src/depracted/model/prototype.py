"""HIP-backed pieces of the reference's ``src/depracted/model/prototype.py``:
the banded patch correlation ``Prototype._fusion`` (:118-156) and the per-sample
EPE ``flow_loss`` (:27-32).  The 1-D conv encoder/decoder around them is plain
``torch.nn`` in the reference and stays on MIOpen.

The correlation is an autograd Function (HIP forward and backward), so it composes
with the torch encoder/decoder for end-to-end training.
"""
import torch

from planar_optical_flow_amd import ops


class _BandCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat1, feat2, kernel_size, max_displacement):
        f1, f2 = feat1.contiguous().float(), feat2.contiguous().float()
        ctx.save_for_backward(f1, f2)
        ctx.cfg = (kernel_size, max_displacement)
        return ops.band_correlation(f1, f2, kernel_size, max_displacement)

    @staticmethod
    def backward(ctx, grad):
        f1, f2 = ctx.saved_tensors
        d1, d2 = ops.band_correlation_backward(f1, f2, grad.contiguous().float(), *ctx.cfg)
        return d1, d2, None, None


def fusion(feat1, feat2, kernel_size=3, max_displacement=5):
    """(B,C,n) x2 -> (B, 2*max_displacement+1, n): correlation of the clamped
    `kernel_size`-tap patch around i in feat1 with the patches around
    clamp(i+d) in feat2, d in [-max_displacement, +max_displacement]."""
    return _BandCorrelation.apply(feat1, feat2, kernel_size, max_displacement)


def flow_loss(pred, target, mask=None):
    """-> (mean over samples of the per-sample EPE, per-sample EPE [B]).
    Differentiable (plain torch); the evaluation-side reduction is
    ``src.utils.eval_utils.loss_fn_eval``."""
    err_batch = torch.mean(torch.norm(pred - target, dim=-1), dim=1)
    return torch.mean(err_batch), err_batch
