"""Box-regression head of the reference (src/model/box_regression.py).

PointNet trunk (1x1 convs 3->64->64->128->1024 + BN + LeakyReLU(0.1), max over
points) and an FC head 1024->512->256->target_dim.  These are dense GEMMs and
stay on rocBLAS / MIOpen through PyTorch-ROCm; what
this module must preserve is the checkpoint ABI: the 72 state-dict keys,
including the unused top-level ``conv1..conv4`` the reference creates by
subclassing its PointNet, and the construction order (so that a given seed
yields the reference's initial weights).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .box_regression_fn import _model_eval_fn, _model_fn


def _pointwise(cin, cout):
    return nn.Sequential(nn.Conv1d(cin, cout, kernel_size=1, padding=0), nn.BatchNorm1d(cout),
                         nn.LeakyReLU(negative_slope=0.1, inplace=True))


def _dense(cin, cout, batch_norm=True, nonlinearity=True):
    layers = [nn.Linear(cin, cout)]
    if batch_norm:
        layers.append(nn.BatchNorm1d(cout))
    if nonlinearity:
        layers.append(nn.LeakyReLU(negative_slope=0.1, inplace=True))
    return layers[0] if len(layers) == 1 else nn.Sequential(*layers)


def regression_loss2(pred, target, alpha=0.5):
    """:52-67.  L1 on the box dimensions + alpha * L1 on the orientation residual
    (+ L1 on z for the 5-target 3-D variant).  On the device: loss and its gradient in one launch
    (``pof::regression_loss2``) instead of a dozen element-wise / reduction kernels each way."""
    if pred.is_cuda and pred.dim() == 2 and pred.shape[1] in (3, 5) and pred.shape[0] > 0 \
            and pred.dtype == torch.float32 and tuple(target.shape) == tuple(pred.shape) and not target.requires_grad:
        from planar_optical_flow_amd import torch_ops
        return torch_ops.regression_loss2_fused(pred, target.float(), alpha)
    ori = torch.mean(torch.abs(pred[..., -1] - target[..., -1]))
    if pred.shape[1] == 5:
        z = torch.mean(torch.abs(pred[..., 0] - target[..., 0]))
        dims = torch.mean(torch.sum(torch.abs(pred[:, 1:-1] - target[:, 1:-1]), dim=1))
        return z + dims + alpha * ori
    if pred.shape[1] == 3:
        dims = torch.mean(torch.sum(torch.abs(pred[:, :-1] - target[:, :-1]), dim=1))
        return dims + alpha * ori
    return None


class PointNet(nn.Module):
    def __init__(self, input_dim=3):
        super().__init__()
        self.conv1 = _pointwise(input_dim, 64)
        self.conv2 = _pointwise(64, 64)
        self.conv3 = _pointwise(64, 128)
        self.conv4 = _pointwise(128, 1024)

    gemm_pointwise = True   # GPU inference: the 1x1 convolutions as library GEMMs over [B*n, C]
    train_pointwise = False  # the same form in training mode (autograd through F.linear / BatchNorm on [B*n, C])
    hip_train = True        # GPU training: every unit one autograd node on the HIP kernels (torch_ops.ConvUnitTrain)

    def forward(self, x):  # x [B, C, n]
        if x.is_cuda and self.gemm_pointwise and (self.train_pointwise or not self.training):
            return self.forward_points(x.permute(0, 2, 1))
        if x.is_cuda and self.training and self.hip_train and torch.is_grad_enabled():
            from planar_optical_flow_amd import torch_ops
            units = (self.conv1, self.conv2, self.conv3, self.conv4)
            if all(type(u[1]) is nn.BatchNorm1d and torch_ops.conv_unit_train_supported(u[0], x.shape[2]) for u in units):
                # (per-rank BatchNorm only: a SyncBatchNorm1d unit keeps the module path, whose statistics are global)
                # float32-MFMA convolution + fused BatchNorm(train)/LeakyReLU tail forward; tail backward, data gradient
                # on the same convolution kernel, weight gradient on the split-K kernel's one-tap form.  The library's
                # path spends a third of its step on the weight-gradient kernels and their layout transposes.
                n = x.shape[2]
                fused_max = n >= 4 and n & (n - 1) == 0       # the max over points inside the last unit's tail
                for u in units[:3]:
                    x = torch_ops.conv_unit_train(x, u[0], u[1], u[2].negative_slope)
                x = torch_ops.conv_unit_train(x, units[3][0], units[3][1], units[3][2].negative_slope, rowmax=fused_max)
                return x if fused_max else torch.max(x, 2, keepdim=True)[0].view(-1, 1024)
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        return torch.max(x, 2, keepdim=True)[0].view(-1, 1024)

    def forward_points(self, pts):  # pts [B, n, C] (the layout the data arrives in)
        """A 1x1 convolution over [B, C, n] is the dense layer [B*n, C] x [C, C'] -- issued as that library
        GEMM (same parameters, same BatchNorm statistics: BatchNorm1d on a 2-D input reduces over B*n),
        MIOpen's 1x1 inference path reaches 9 TFLOP/s on these shapes (B = 4096: 9.4 ms -> 2.2 ms).  Used in
        eval mode; MIOpen's training-mode solvers are faster than autograd through this form (12 vs 16 ms)."""
        B, n, C = pts.shape
        h = pts.reshape(B * n, C)
        for unit in (self.conv1, self.conv2, self.conv3, self.conv4):
            conv, bn, act = unit[0], unit[1], unit[2]
            h = act(bn(F.linear(h, conv.weight.squeeze(-1), conv.bias)))
        return h.view(B, n, -1).amax(dim=1)


class BoundingBoxRegressor(PointNet):
    def __init__(self, cfg):
        super().__init__()  # creates the (unused) top-level conv1..conv4 of the checkpoint ABI
        self.dropout = cfg["dropout"]
        self.backbone = PointNet(input_dim=cfg["input_dim"])
        self.fc1 = _dense(1024, 512)
        self.fc2 = _dense(512, 256)
        self.fc3 = _dense(256, cfg["target_dim"], batch_norm=False, nonlinearity=False)
        self.loss_fn = regression_loss2
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, a=0.1, nonlinearity="leaky_relu")
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def model_fn(model, batch_data):
        return _model_fn(model, batch_data)

    @staticmethod
    def model_eval_fn(model, batch_data):
        return _model_eval_fn(model, batch_data)

    small_batch_rows = 1024   # at most this many rows: the dense layers' forward on pof::linear_bias (GPU)

    def _dense_forward(self, unit, x):
        """fc1 / fc2 / fc3 on a batch of a few hundred rows: the BLAS library runs an output of at most 256 x 256 as one
        tile on one CU (fc2: 118 us of a 0.9 ms training step); the small-batch MFMA kernel takes 32 x 32 tiles."""
        if not (x.is_cuda and x.shape[0] <= self.small_batch_rows):
            return unit(x)
        from planar_optical_flow_amd import torch_ops
        if isinstance(unit, nn.Linear):
            return torch_ops.linear_small(x, unit)
        x = torch_ops.linear_small(x, unit[0])
        for layer in list(unit)[1:]:
            x = layer(x)
        return x

    def forward(self, x):  # x [B, n, C]
        bb = self.backbone
        x = bb.forward_points(x) if (x.is_cuda and bb.gemm_pointwise and (bb.train_pointwise or not self.training)) \
            else bb(x.permute(0, 2, 1))
        x = self._dense_forward(self.fc2, self._dense_forward(self.fc1, x))
        if self.dropout > 0.0:
            x = F.dropout(x, p=self.dropout, training=self.training)
        return self._dense_forward(self.fc3, x)
