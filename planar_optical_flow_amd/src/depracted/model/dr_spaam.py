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

from planar_optical_flow_amd import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.pof.*)


def flow_loss(pred, target, mask=None):
    err = torch.norm(pred - target, dim=-1)
    return torch.mean(err[mask == 1.0]) if mask is not None else torch.mean(err)


class _WindowedAttention:
    """The gate's banded similarity / softmax / template merge as the registered operator
    ``torch.ops.pof.spatial_attention`` (planar_optical_flow_amd/torch_ops.py: HIP forward and backward kernels,
    autograd formula and fake kernel registered with torch.library, so autograd and torch.compile see one op)."""

    @staticmethod
    def apply(emb_x, emb_t, x, tmpl, alpha, window):
        out, band, _ = torch.ops.pof.spatial_attention(emb_x.contiguous().float(), emb_t.contiguous().float(),
                                                       x.contiguous().float(), tmpl.contiguous().float(),
                                                       float(alpha), int(window))
        return out, band


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

    def _embed(self, flat):
        """The embedding conv spans the whole cutout (kernel_size = n_pts), i.e. it is the dense product
        [B*N, C*P] x [C*P, 128]: run it as one library GEMM (same parameters, same gradients) instead of
        a convolution, then the module's own BatchNorm and LeakyReLU."""
        conv, bn, act = self.conv[0], self.conv[1], self.conv[2]
        folded = getattr(self, "_folded", None)
        if folded is not None and not self.training and not torch.is_grad_enabled():
            # inference after fold_for_inference(): the BatchNorm lives in the GEMM's weights and bias -- two launches
            # (GEMM with bias, LeakyReLU) instead of five; a launch costs ~5 us in the streaming step's graph
            return torch.nn.functional.leaky_relu(torch.nn.functional.linear(flat, folded[0], folded[1]),
                                                  act.negative_slope)
        return act(bn(torch.nn.functional.linear(flat, conv.weight.reshape(conv.out_channels, -1), conv.bias)))

    def fold_for_inference(self, enable=True):
        """Fold the embedding's BatchNorm (running statistics) into its weight and bias.  Call after loading a
        checkpoint (``DROW.fuse_for_inference`` does); ``train()`` drops the folded copy."""
        self._folded = None
        if enable:
            conv, bn = self.conv[0], self.conv[1]
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (conv.weight.reshape(conv.out_channels, -1) * scale[:, None]).contiguous()
                b = (bn.bias + (conv.bias - bn.running_mean) * scale).contiguous()
            self._folded = (w, b)
        return self

    def train(self, mode=True):
        if mode:
            self._folded = None
        return super().train(mode)

    def forward(self, x, x_template):
        """x, x_template [B, n_cutout, n_channel, n_pts] -> (fused template of the
        same shape, pre-softmax window similarities [B, n_cutout, window])."""
        B, N, C, P = x.shape
        emb_x, emb_t = self._embed(x.reshape(B * N, C * P)).view(B, N, 128), \
            self._embed(x_template.reshape(B * N, C * P)).view(B, N, 128)
        out, band = _WindowedAttention.apply(emb_x, emb_t, x, x_template, self._alpha, self._window_size)
        return out, band


# ---------------------------------------------------------------------------------------
# DROW / SpatialDROW / FlowDROW_pretrained (reference :8-19, :41-121, :220-322).
# Same sub-module names, construction order and initialisation as the reference, so a
# reference checkpoint loads with load_state_dict and a seeded construction reproduces the
# reference's weights.  The conv trunks run on the HIP MFMA kernels (inference: fuse_for_inference();
# training: torch_ops.TrunkUnitTrain); the temporal gate is the
# HIP attention above.
# ---------------------------------------------------------------------------------------
def _conv(in_channel, out_channel, kernel_size, padding):
    return nn.Sequential(nn.Conv1d(in_channel, out_channel, kernel_size=kernel_size, padding=padding),
                         nn.BatchNorm1d(out_channel), nn.LeakyReLU(negative_slope=0.1, inplace=True))


def _conv3x3(in_channel, out_channel):
    return _conv(in_channel, out_channel, kernel_size=3, padding=1)


def _init_weights(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nn.init.kaiming_normal_(m.weight, a=0.1, nonlinearity="leaky_relu")
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class DROW(nn.Module):
    """Per-point detector on cutouts [B, n_cutout, n_scan, n_pts] -> (pred_cls [B, n_cutout, 4|1],
    pred_reg [B, n_cutout, 2]); the scans of a window are fused by summation (:41-121)."""

    _TRUNK = ((1, 64, 64, 128), (128, 128, 128, 256), (256, 256, 256, 512))

    def __init__(self, dropout=0.5, num_scans=5, num_pts=48, focal_loss_gamma=0.0, pedestrian_only=False):
        super().__init__()
        import torch.nn.functional as F
        from .loss_utils import BinaryFocalLoss, FocalLoss
        self.dropout = 0.0   # the reference ignores its `dropout` argument (:47-48)
        for b, ch in enumerate(self._TRUNK, start=1):
            setattr(self, "conv_block_%d" % b, nn.Sequential(*[_conv3x3(ch[i], ch[i + 1]) for i in range(3)]))
        self.conv_block_4 = nn.Sequential(_conv3x3(512, 256), _conv3x3(256, 128))
        if pedestrian_only:
            self.conv_cls = nn.Conv1d(128, 1, kernel_size=1)
            self.cls_loss = BinaryFocalLoss(gamma=focal_loss_gamma) if focal_loss_gamma > 0.0 \
                else F.binary_cross_entropy
        else:
            self.conv_cls = nn.Conv1d(128, 4, kernel_size=1)
            self.cls_loss = FocalLoss(gamma=focal_loss_gamma) if focal_loss_gamma > 0.0 else F.cross_entropy
        self.conv_reg = nn.Conv1d(128, 2, kernel_size=1)
        _init_weights(self)

    # ---- inference on the HIP trunk kernels ------------------------------------------------
    def fuse_for_inference(self, enable=True):
        """Fold every conv3 + BatchNorm (running statistics) + bias of the four trunk blocks into
        (transposed weight, scale, shift) triples for ``pof_conv3_bn_lrelu``.  Call after loading a
        checkpoint and after ``.cuda()``; eval-mode forwards then run the trunk as float32-MFMA
        implicit GEMMs instead of MIOpen convolutions.  (Training mode has its own HIP route, _run_block_train.)"""
        self._fused = None
        if not enable:
            if getattr(self, "gate", None) is not None:
                self.gate.fold_for_inference(False)
            return self
        fused = {}
        with torch.no_grad():
            for name in ("conv_block_1", "conv_block_2", "conv_block_3", "conv_block_4"):
                layers = []
                for unit in getattr(self, name):
                    conv, bn = unit[0], unit[1]
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias + (conv.bias - bn.running_mean) * scale
                    layers.append((conv.weight.permute(2, 1, 0).contiguous().float(), scale.float().contiguous(),
                                   shift.float().contiguous()))
                fused[name] = layers
            # the single-channel first unit as a per-channel table for pof_conv3_first_two: taps x scale, shift
            wt0, sc0, sh0 = fused["conv_block_1"][0]
            if wt0.shape[1] == 1 and wt0.shape[2] <= 128:
                fused["first_unit_table"] = torch.cat((wt0[:, 0, :].t() * sc0[:, None], sh0[:, None]), dim=1).contiguous()
        self._fused = fused
        gate = getattr(self, "gate", None)
        if gate is not None:
            gate.fold_for_inference(True)
        return self

    def train(self, mode=True):
        """Entering training mode drops the folded inference parameters: the weights and the BatchNorm
        running statistics are about to change, so ``fuse_for_inference()`` has to be called again."""
        if mode:
            self._fused = None
        return super().train(mode)

    def _slope(self, name, i):
        return float(getattr(self, name)[i][2].negative_slope)

    def _run_block(self, x, name, pool):
        """One trunk block; pooled blocks pool after their last layer.  Three routes:
        eval + fuse_for_inference(): the HIP conv kernels (17 ms per B = 32 forward); eval without it on
        the GPU: the channels-last GEMM form below (39 ms; MIOpen's inference path takes 234 ms on these
        shapes); training on the GPU: every unit as one autograd node on the HIP kernels (_run_block_train);
        CPU: the plain torch modules."""
        fused = getattr(self, "_fused", None)
        if fused is not None and not self.training and x.is_cuda and not torch.is_grad_enabled():
            x = x.contiguous().float()
            layers = fused[name]

            table = fused.get("first_unit_table") if name == "conv_block_1" and getattr(self, "fuse_first_unit", True) else None

            def run(seqs):
                start = 0
                if table is not None and seqs.shape[1] == 1 and len(layers) > 1:
                    # units 0 and 1 in one launch: the 64-channel output of the first is never written (1 GB at B = 32)
                    wt, scale, shift = layers[1]
                    seqs = ops.conv3_first_two(seqs, table, wt, scale, shift, slope1=self._slope(name, 0),
                                               pool=pool and len(layers) == 2, negative_slope=self._slope(name, 1))
                    start = 2
                for i in range(start, len(layers)):
                    wt, scale, shift = layers[i]
                    seqs = ops.conv3_bn_lrelu(seqs, wt, scale, shift, pool=pool and i == len(layers) - 1)
                return seqs
            # large batches go through the block in slabs of sequences: the intermediate activations
            # (64 x 56 floats per sequence and layer) stay bounded, only the block output is full size
            slab = getattr(self, "fused_slab", 1 << 18)
            S = x.shape[0]
            if S <= slab:
                return run(x)
            first = run(x[:slab])
            out = torch.empty((S,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
            out[:slab] = first
            for s0 in range(slab, S, slab):
                out[s0:s0 + slab] = run(x[s0:s0 + slab])
            return out
        if x.is_cuda and not self.training and getattr(self, "gemm_trunk", True):
            return self._run_block_gemm(x, getattr(self, name), pool)
        if x.is_cuda and self.training and getattr(self, "fused_train_tail", True):
            return self._run_block_train(x, getattr(self, name), pool, getattr(self, "hip_train_conv", True))
        out = getattr(self, name)(x)
        return torch.max_pool1d(out, kernel_size=2) if pool else out

    @staticmethod
    def _unit_routes(unit, hip_conv, S, L, dtype, last, groups=1):
        """(conv on the HIP kernels?, tail on the fused passes?) for one trunk unit and input shape."""
        conv, bn = unit[0], unit[1]
        conv_ok = hip_conv and conv.kernel_size == (3,) and conv.padding == (1,) and conv.stride == (1,) \
            and conv.dilation == (1,) and conv.groups == 1 and conv.padding_mode == "zeros" \
            and dtype == torch.float32
        tail_ok = type(bn) is nn.BatchNorm1d and bn.training and bn.affine and dtype == torch.float32 \
            and (groups == 1 or bn.momentum is not None) \
            and ops.bn_lrelu_pool_supported(S, conv.out_channels, L, last, groups)
        return conv_ok, tail_ok

    @staticmethod
    def _run_block_train(x, block, pool, hip_conv=True, groups=1):
        """Training on the GPU: every unit -- convolution forward, data and weight gradient on the HIP MFMA kernels,
        BatchNorm(train) + LeakyReLU [+ max-pool] and its backward as the fused passes of ``pof::bn_lrelu_pool`` --
        as one autograd node (``torch_ops.TrunkUnitTrain``).  ``groups`` > 1: x holds that many
        batches one after the other, each with its own BatchNorm statistics (all units must take the fused
        route then; ``_grouped_blocks_ok`` checks)."""
        out = x
        for i, unit in enumerate(block):
            conv, bn, act = unit[0], unit[1], unit[2]
            last = pool and i == len(block) - 1
            conv_ok, tail_ok = DROW._unit_routes(unit, hip_conv, out.shape[0], out.shape[2], out.dtype, last, groups)
            if conv_ok and tail_ok:
                out = torch_ops.trunk_unit_train(out, conv, bn, act.negative_slope, last, groups)
                continue
            if groups != 1:
                raise RuntimeError("grouped trunk pass on a unit that cannot take the fused route")
            y = torch_ops.conv3_train(out, conv) if conv_ok else conv(out)
            if tail_ok and y.dtype == torch.float32:
                out = torch_ops.bn_lrelu_pool_train(y, bn, act.negative_slope, last)
            else:   # SyncBatchNorm, frozen statistics, autocast, odd shapes: the modules themselves
                out = act(bn(y))
                if last:
                    out = torch.max_pool1d(out, kernel_size=2)
        return out

    def _grouped_blocks_ok(self, names, S, L, dtype, groups):
        """Every unit of the pooled blocks `names` takes the fused route for S sequences of L points in `groups`
        statistics groups."""
        hip_conv = getattr(self, "hip_train_conv", True)
        for name in names:
            block = getattr(self, name)
            for i, unit in enumerate(block):
                last = i == len(block) - 1
                if not all(self._unit_routes(unit, hip_conv, S, L, dtype, last, groups)):
                    return False
            L //= 2
        return True

    @staticmethod
    def _run_block_gemm(x, block, pool):
        """conv3 + BatchNorm + LeakyReLU units as ONE large library GEMM each, channels last:
        cols[S*L, 3*Ci] (the three taps side by side) x W[3*Ci, Co] (+ bias) -> [S*L, Co], then the unit's
        own BatchNorm1d (2-D input: statistics over sequences and positions, as for [S, Co, L]) and
        LeakyReLU.  MIOpen's inference path has no fast algorithm for 10^5 sequences of <= 56 points;
        hipBLASLt runs these products near its float32 peak."""
        S, Ci, L = x.shape
        h = x.permute(0, 2, 1).contiguous()                                   # [S, L, Ci]
        for unit in block:
            conv, bn, act = unit[0], unit[1], unit[2]
            hp = torch.nn.functional.pad(h, (0, 0, 1, 1))                     # zero rows at both ends of every sequence
            cols = torch.cat((hp[:, 0:L], hp[:, 1:L + 1], hp[:, 2:L + 2]), dim=2).reshape(S * L, -1)
            w = conv.weight.permute(2, 1, 0).reshape(-1, conv.out_channels)   # rows (tap, ci)
            h = act(bn(torch.addmm(conv.bias, cols, w))).view(S, L, -1)
        if pool:
            h = h.view(S, L // 2, 2, h.shape[-1]).amax(dim=2)
        return h.permute(0, 2, 1).contiguous()                                # [S, Co, L']

    def _forward_conv(self, x, conv_block):
        name = next(n for n in ("conv_block_1", "conv_block_2", "conv_block_3") if getattr(self, n) is conv_block)
        out = self._run_block(x, name, pool=True)
        if self.dropout > 0:
            out = torch.dropout(out, self.dropout, self.training)
        return out

    def _forward_cutout(self, x):
        B, N, T, P = x.shape
        out = self._forward_conv(x.reshape(B * N * T, 1, P), self.conv_block_1)
        out = self._forward_conv(out, self.conv_block_2)
        return out.view(B, N, T, out.shape[-2], out.shape[-1])

    def _fuse_cutout(self, x):
        return torch.sum(x, dim=2)

    def _forward_fused_cutout(self, x):
        B, N, C, P = x.shape
        out = self._forward_conv(x.reshape(B * N, C, P), self.conv_block_3)
        out = self._run_block(out, "conv_block_4", pool=False)
        # average over the remaining positions, then the two 1x1 convolutions -- on a length-1 sequence
        # they are dense layers: issue them as library GEMMs (MIOpen runs 1x1 convs on [B*N, 128, 1]
        # through its naive kernel)
        if getattr(self, "_fused", None) is not None and not self.training and out.is_cuda \
                and not torch.is_grad_enabled() and self.conv_cls.out_channels <= 6:
            # inference after fuse_for_inference(): mean + both heads in one launch (pof_drow_heads)
            pred_cls, pred_reg = ops.drow_heads(out.contiguous().float(), self.conv_cls.weight, self.conv_cls.bias,
                                                self.conv_reg.weight, self.conv_reg.bias)
            return pred_cls.view(B, N, -1), pred_reg.view(B, N, 2)
        feat = out.mean(dim=-1)
        lin = torch.nn.functional.linear
        pred_cls = lin(feat, self.conv_cls.weight.squeeze(-1), self.conv_cls.bias)
        pred_reg = lin(feat, self.conv_reg.weight.squeeze(-1), self.conv_reg.bias)
        return pred_cls.view(B, N, -1), pred_reg.view(B, N, 2)

    def forward(self, x):
        return self._forward_fused_cutout(self._fuse_cutout(self._forward_cutout(x)))


class SpatialDROW(DROW):
    """DR-SPAAM: the per-scan features are fused auto-regressively through the windowed spatial
    attention gate instead of being summed (:220-277)."""

    def __init__(self, dropout=0.5, num_scans=5, num_pts=48, focal_loss_gamma=0.0, alpha=0.5, window_size=7,
                 pedestrian_only=False):
        super().__init__(dropout=dropout, num_scans=num_scans, num_pts=num_pts,
                         focal_loss_gamma=focal_loss_gamma, pedestrian_only=pedestrian_only)
        from math import ceil
        self.gate = _SpatialAttention(n_pts=int(ceil(num_pts / 4)), n_channel=256, alpha=alpha,
                                      window_size=window_size)
        self.loss_fn = flow_loss

    def _scan_features(self, x, t):
        return self._forward_cutout(x[:, :, t, :].unsqueeze(dim=2)).squeeze(dim=2)

    def _all_scan_features(self, x):
        """Training on the GPU: the trunk features of ALL scans of the window in one pass per layer instead of one
        pass per scan.  The reference sends the scans through blocks 1-2 one after the other (:246-262), i.e. every
        BatchNorm normalises each scan with that scan's own batch statistics and updates its running statistics
        once per scan; the convolutions are per-sequence anyway.  Here the scans are stacked scan-major and the fused
        units run with ``groups = n_scan`` statistics groups -- the same numbers, from 5x larger launches (no
        partly filled last round of workgroups per scan, one weight-gradient reduction instead of five plus four
        gradient accumulations per parameter).  Returns the per-scan features, or None when a unit cannot take the
        fused route (SyncBatchNorm, CPU, autocast, ...): the caller then goes scan by scan."""
        B, N, T, P = x.shape
        if not self.training and x.is_cuda and T > 1 and getattr(self, "_fused", None) is not None \
                and not torch.is_grad_enabled() and getattr(self, "grouped_scans", True):
            # inference on the folded trunk: nothing couples the sequences, the scans simply share the launches
            out = x.permute(2, 0, 1, 3).reshape(T * B * N, 1, P)
            for name in ("conv_block_1", "conv_block_2"):
                out = self._run_block(out, name, pool=True)
            return out.view(T, B, N, out.shape[-2], out.shape[-1]).unbind(0)
        if not (self.training and x.is_cuda and x.dtype == torch.float32 and T > 1 and self.dropout == 0
                and getattr(self, "fused_train_tail", True) and getattr(self, "grouped_scans", True)
                and self._grouped_blocks_ok(("conv_block_1", "conv_block_2"), T * B * N, P, x.dtype, T)):
            return None
        out = x.permute(2, 0, 1, 3).reshape(T * B * N, 1, P)                  # scan-major: one group per scan
        for name in ("conv_block_1", "conv_block_2"):
            out = self._run_block_train(out, getattr(self, name), True, getattr(self, "hip_train_conv", True), T)
        return out.view(T, B, N, out.shape[-2], out.shape[-1]).unbind(0)

    def forward(self, x, testing=False, fea_template=None):
        if testing:   # streaming inference: one new scan against the running template
            out = self._scan_features(x, 0)
            if fea_template is None:
                out_template = out.clone()
                _, feat_fused = self.gate(out, out_template)
            else:
                out_template, feat_fused = self.gate(out, fea_template)
            pred_cls, pred_reg = self._forward_fused_cutout(out_template)
            return pred_cls, pred_reg, out_template, feat_fused
        n_scan = x.shape[2]
        with torch_ops.weight_layout_scope():   # the same trunk weights serve every scan of the window
            feats = self._all_scan_features(x)
            scan = (lambda t: feats[t]) if feats is not None else (lambda t: self._scan_features(x, t))
            out_template = scan(0)
            for i in range(1, n_scan - 1):
                out_template, _ = self.gate(scan(i), out_template)
            out_template, feat_fused = self.gate(scan(n_scan - 1), out_template)
            pred_cls, pred_reg = self._forward_fused_cutout(out_template)
        return pred_cls, pred_reg, feat_fused


class FlowDROW_pretrained(nn.Module):
    """Frozen DR-SPAAM + a small conv head on the window similarities (plus the current range) that
    regresses the per-point flow (:279-322).  ``pre_trained_ckpt=None`` skips the checkpoint load
    (the reference always loads ./pre_trained_ckpts/dr_spaam_e40.pth)."""

    def __init__(self, dropout=0.5, num_scans=5, num_pts=48, focal_loss_gamma=0.0, alpha=0.5, window_size=7,
                 pedestrian_only=False, pre_trained_ckpt="./pre_trained_ckpts/dr_spaam_e40.pth"):
        super().__init__()
        self.dr_spaam = SpatialDROW(num_scans=num_scans, num_pts=num_pts, focal_loss_gamma=focal_loss_gamma,
                                    alpha=alpha, window_size=window_size, pedestrian_only=pedestrian_only)
        if pre_trained_ckpt is not None:
            ckpt = torch.load(pre_trained_ckpt, map_location="cpu", weights_only=False)
            self.dr_spaam.load_state_dict(ckpt["model_state"] if "model_state" in ckpt else ckpt)
        for param in self.dr_spaam.parameters():
            param.requires_grad = False
        self.conv1 = _conv(window_size, 128, kernel_size=3, padding=1)
        self.conv2 = _conv(128, 64, kernel_size=3, padding=1)
        self.conv3 = _conv(64, 32, kernel_size=3, padding=1)
        self.pw = _conv(32, 2, kernel_size=1, padding=0)
        self.loss_fn = flow_loss
        self._feat = None

    def forward(self, x, cur_scan, testing=False, fea_template=None):
        if testing:
            pred_cls, pred_reg, self._feat, feat_fused = self.dr_spaam(x, testing=testing, fea_template=self._feat)
        else:
            pred_cls, pred_reg, feat_fused = self.dr_spaam(x)
        # the reference's two permutes cancel: conv1 sees [B, n_cutout, window + 1], i.e. it treats the
        # cutouts as channels -- which only has the right shape when n_cutout == window_size (:311-314)
        feat = torch.cat((feat_fused, cur_scan.unsqueeze(dim=-1)), dim=-1)
        out = self.conv3(self.conv2(self.conv1(feat)))
        return pred_cls, pred_reg, self.pw(out).permute(0, 2, 1)
