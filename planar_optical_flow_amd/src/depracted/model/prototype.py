"""HIP-backed pieces of the reference's ``src/depracted/model/prototype.py``:
the banded patch correlation ``Prototype._fusion`` (:118-156) and the per-sample
EPE ``flow_loss`` (:27-32).  The 1-D conv encoder / decoder around them is plain
``torch.nn`` in the reference; here the modules keep the reference's parameters and
state-dict keys, training runs them through torch, and inference -- after
``fuse_for_inference()`` -- runs every unit (stride-2 and stride-1 k = 3 convolutions, the
point-wise head, BatchNorm folded, LeakyReLU) on the float32-MFMA kernel ``pof_conv1d_bn_lrelu``
(round 3), both scans of a pair in one launch per layer.

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

    _UNITS = ("encoder_0", "encoder_1", "encoder_2", "decoder_1", "decoder_0", "flow_reg")

    def fuse_for_inference(self, enable=True):
        """Fold every unit's Conv1d bias + BatchNorm (running statistics) into (transposed weight [k, Ci, Co], scale,
        shift) for ``pof_conv1d_bn_lrelu``.  Call after loading a checkpoint and after ``.cuda()``; eval-mode forwards on
        the device then leave MIOpen entirely.  ``train()`` drops the folded copy."""
        self._fused = None
        if enable:
            fused = {}
            with torch.no_grad():
                for name in self._UNITS:
                    conv, bn, act = getattr(self, name)
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias + (conv.bias - bn.running_mean) * scale
                    fused[name] = (conv.weight.permute(2, 1, 0).contiguous().float(), scale.float().contiguous(),
                                   shift.float().contiguous(), int(conv.stride[0]), float(act.negative_slope))
            self._fused = fused
        return self

    def train(self, mode=True):
        if mode:
            self._fused = None
        return super().train(mode)

    def _unit_hip(self, name, x):
        wt, scale, shift, stride, slope = self._fused[name]
        return ops.conv1d_bn_lrelu(x.contiguous(), wt, scale, shift, stride=stride, negative_slope=slope)

    def _forward_fused(self, scan1, scan2):
        B = scan1.shape[0]
        s1 = scan1.permute(0, 2, 1).contiguous().float()
        both = torch.cat((s1, scan2.permute(0, 2, 1).contiguous().float()), dim=0)     # [2B, C, n]: one launch per layer
        skips, f = [], both
        for name in ("encoder_0", "encoder_1", "encoder_2"):
            f = self._unit_hip(name, f)
            skips.append(f[:B])
        out = self._fusion(f[:B], f[B:], max_displacement=self.max_displacement)
        for name, skip in (("decoder_1", skips[1]), ("decoder_0", skips[0])):
            out = self._unit_hip(name, torch.cat((skip, self._upsample(out, size=skip.shape[-1])), dim=1))
        out = self._unit_hip("flow_reg", torch.cat((s1, self._upsample(out, size=s1.shape[-1])), dim=1))
        return out.permute(0, 2, 1)

    hip_train = True    # training on the device: every unit through torch_ops.ConvUnitTrain (False: the torch modules)

    def _unit_train(self, name, x, groups=1):
        conv, bn, act = getattr(self, name)
        if type(bn) is nn.BatchNorm1d and torch_ops.conv_unit_train_supported(conv, x.shape[2]):
            return torch_ops.conv_unit_train(x, conv, bn, act.negative_slope, groups=groups)
        if conv.kernel_size[0] == 1 and groups == 1:       # the point-wise head: HIP convolution, torch BatchNorm
            return act(bn(torch_ops.Conv1dTrain.apply(x.contiguous().float(), conv.weight, conv.bias)))
        if groups == 1:
            return act(bn(conv(x)))
        return torch.cat([act(bn(conv(part))) for part in x.chunk(groups, dim=0)], dim=0)

    def _forward_train_hip(self, scan1, scan2):
        """The training forward with every unit as one autograd node on the HIP kernels (convolution forward / data /
        weight gradients + the fused BatchNorm(train) tail).  The two scans of a pair go through the encoders as two
        statistics groups of one launch: each is normalised with its own batch statistics and the running statistics
        are updated scan 1 first, then scan 2 -- what the reference's two calls per encoder do (prototype.py:70-80)."""
        B = scan1.shape[0]
        s1 = scan1.permute(0, 2, 1).contiguous().float()
        f = torch.cat((s1, scan2.permute(0, 2, 1).contiguous().float()), dim=0)
        skips = []
        for name in ("encoder_0", "encoder_1", "encoder_2"):
            f = self._unit_train(name, f, groups=2)
            skips.append(f[:B])
        out = self._fusion(f[:B], f[B:], max_displacement=self.max_displacement)
        for name, skip in (("decoder_1", skips[1]), ("decoder_0", skips[0])):
            out = self._unit_train(name, torch.cat((skip, self._upsample(out, size=skip.shape[-1])), dim=1))
        out = self._unit_train("flow_reg", torch.cat((s1, self._upsample(out, size=s1.shape[-1])), dim=1))
        return out.permute(0, 2, 1)

    def forward(self, scan1, scan2=None):
        """scan1, scan2 [B, n_pts, n_channel] -> per-point flow [B, n_pts, 2]."""
        if scan2 is None:
            scan2 = scan1
        if getattr(self, "_fused", None) is not None and not self.training and scan1.is_cuda \
                and not torch.is_grad_enabled():
            return self._forward_fused(scan1, scan2)
        if self.training and scan1.is_cuda and self.hip_train:
            return self._forward_train_hip(scan1, scan2)
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
