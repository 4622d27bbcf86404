"""HIP-backed pieces of the reference's ``src/depracted/model/prototype.py``:
the banded patch correlation ``Prototype._fusion`` (:118-156) and the per-sample
EPE ``flow_loss`` (:27-32).  The 1-D conv encoder/decoder around them is plain
``torch.nn`` in the reference and stays on MIOpen.

The correlation is an autograd Function (HIP forward and backward), so it composes
with the torch encoder/decoder for end-to-end training.
"""
import torch

from planar_optical_flow_amd import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.pof.*)


def fusion(feat1, feat2, kernel_size=3, max_displacement=5):
    """(B,C,n) x2 -> (B, 2*max_displacement+1, n): correlation of the clamped
    `kernel_size`-tap patch around i in feat1 with the patches around
    clamp(i+d) in feat2, d in [-max_displacement, +max_displacement].  The registered operator
    ``torch.ops.pof.band_correlation`` (torch_ops.py): HIP forward / backward, autograd formula, fake kernel."""
    return torch.ops.pof.band_correlation(feat1.contiguous().float(), feat2.contiguous().float(), int(kernel_size),
                                          int(max_displacement))


def flow_loss(pred, target, mask=None):
    """-> (mean over samples of the per-sample EPE, per-sample EPE [B]).
    Differentiable (plain torch); the evaluation-side reduction is
    ``src.utils.eval_utils.loss_fn_eval``."""
    err_batch = torch.mean(torch.norm(pred - target, dim=-1), dim=1)
    return torch.mean(err_batch), err_batch


# ---------------------------------------------------------------------------------------
# The network around the correlation (reference :6-116): three stride-2 encoders on both scans,
# banded correlation of the deepest features, two decoders with skip connections from scan 1,
# nearest-neighbour upsampling, point-wise flow head.  Same attribute names / construction order /
# initialisation as the reference (state-dict compatible); `_fusion` is the HIP correlation.
# ---------------------------------------------------------------------------------------
import torch.nn as nn
import torch.nn.functional as F


def _unit(in_channel, out_channel, kernel_size=3, stride=1):
    return nn.Sequential(nn.Conv1d(in_channel, out_channel, kernel_size=kernel_size, stride=stride,
                                   padding=kernel_size // 2),
                         nn.BatchNorm1d(out_channel), nn.LeakyReLU(negative_slope=0.01, inplace=True))


class Prototype(nn.Module):
    def __init__(self, in_channel=1, max_displacement=5):
        super().__init__()
        self.max_displacement = max_displacement
        self.encoder_0 = _unit(in_channel, 64, stride=2)
        self.encoder_1 = _unit(64, 128, stride=2)
        self.encoder_2 = _unit(128, 256, stride=2)
        self.decoder_1 = _unit(2 * max_displacement + 1 + 128, 128)
        self.decoder_0 = _unit(128 + 64, 128)
        self.flow_reg = _unit(128 + in_channel, 2, kernel_size=1)
        self.loss_fn = flow_loss
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, a=0.1, nonlinearity="leaky_relu")
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _fusion(self, feat1, feat2, kernel_size=3, max_displacement=5):
        return fusion(feat1, feat2, kernel_size, max_displacement)

    @staticmethod
    def _upsample(x, size):
        return F.interpolate(x, size=size, mode="nearest")

    def forward(self, scan1, scan2=None):
        """scan1, scan2 [B, n_pts, n_channel] -> per-point flow [B, n_pts, 2]."""
        if scan2 is None:
            scan2 = scan1
        s1, s2 = scan1.permute(0, 2, 1), scan2.permute(0, 2, 1)
        skips, f1, f2 = [], s1, s2
        for enc in (self.encoder_0, self.encoder_1, self.encoder_2):
            f1, f2 = enc(f1), enc(f2)
            skips.append(f1)
        out = self._fusion(f1, f2, max_displacement=self.max_displacement)
        for dec, skip in ((self.decoder_1, skips[1]), (self.decoder_0, skips[0])):
            out = dec(torch.cat((skip, self._upsample(out, size=skip.shape[-1])), dim=1))
        out = self.flow_reg(torch.cat((s1, self._upsample(out, size=s1.shape[-1])), dim=1))
        return out.permute(0, 2, 1)
