"""Classification losses of the DROW heads (reference: src/depracted/model/loss_utils.py).

``FocalLoss`` (multi-class, on logits) and ``BinaryFocalLoss`` / ``binary_focal_loss`` (on
probabilities) with the reference's argument names and reductions."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reduce(loss, reduction):
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":
        return loss
    raise RuntimeError


class FocalLoss(nn.Module):
    def __init__(self, gamma=0, alpha=None):
        super().__init__()
        self.gamma = gamma
        if isinstance(alpha, (float, int)):
            alpha = torch.Tensor([alpha, 1 - alpha])
        elif isinstance(alpha, list):
            alpha = torch.Tensor(alpha)
        self.alpha = alpha

    def forward(self, input, target, reduction="mean"):
        if input.dim() > 2:   # [N, C, ...] -> [N * ..., C]
            input = input.reshape(input.size(0), input.size(1), -1).transpose(1, 2).reshape(-1, input.size(1))
        target = target.reshape(-1, 1)
        logpt = F.log_softmax(input, dim=1).gather(1, target).reshape(-1)
        pt = logpt.exp()
        if self.alpha is not None:
            self.alpha = self.alpha.to(input)
            logpt = logpt * self.alpha.gather(0, target.reshape(-1))
        return _reduce(-1 * (1 - pt) ** self.gamma * logpt, reduction)


def binary_focal_loss(pred, target, gamma=2.0, alpha=-1, reduction="mean"):
    pos = -target * (1.0 - pred) ** gamma * torch.log(pred)
    neg = -(1.0 - target) * pred ** gamma * torch.log(1.0 - pred)
    if 0.0 <= alpha <= 1.0:
        pos, neg = pos * alpha, neg * (1.0 - alpha)
    return _reduce(pos + neg, reduction)


class BinaryFocalLoss(nn.Module):
    def __init__(self, gamma=2.0, alpha=-1):
        super().__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, pred, target, reduction="mean"):
        return binary_focal_loss(pred, target, self.gamma, self.alpha, reduction)
