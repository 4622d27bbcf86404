"""model_fn / model_eval_fn of the box head (reference src/model/box_regression_fn.py).

Differences from the reference, both on purpose:
* the evaluation IoU uses the batched HIP rotated-IoU kernel -- one launch for
  the whole batch instead of one numba.cuda launch per sample (:76-82);
* the device is taken from the model, so the same code runs under one process
  per GPU.
"""
import numpy as np
import torch

from ..utils.rotate_iou import rotate_iou_batched


def _to_model(x, model):
    dev = next(model.parameters()).device
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(dev, non_blocking=True).float()


def _np(x):
    """Host copy of a batch entry (NumPy array, list, or a device tensor of the device data sets)."""
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _model_fn(model, batch):
    """-> (loss tensor, tb_dict, rtn_dict{"pred"})."""
    tb_dict, rtn_dict = {}, {}
    pred = model(_to_model(batch["input"], model))
    loss = model.loss_fn(pred, _to_model(batch["target"], model))
    rtn_dict["pred"] = pred
    return loss, tb_dict, rtn_dict


def _model_eval_fn(model, batch):
    loss, tb_dict, rtn = _model_fn(model, batch)
    target = np.array(_np(batch["target"]), dtype=np.float64, copy=True)
    pred = rtn["pred"].detach().cpu().numpy().astype(np.float64)
    det_center, box_center = _np(batch["det_center"]), _np(batch["box_center"])
    inp = _np(batch["input"])
    is_3d = box_center.shape[1] == 3
    loss_z = np.zeros(len(pred))
    if is_3d:
        pred[:, 0] += det_center[:, -1]          # cz back to the global frame
        target[:, 0] += det_center[:, -1]
        loss_z = np.abs(pred[:, 0] - target[:, 0])
        loss_dims = np.sum(np.abs(pred[:, 1:-1] - target[:, 1:-1]), axis=1)
        centre = det_center[:, :2]
    else:
        loss_dims = np.sum(np.abs(pred[:, :-1] - target[:, :-1]), axis=1)
        centre = det_center
    pred[:, -1] += inp[:, 0, -1]                 # orientation = input angle + regressed residual
    pred = np.hstack((centre, pred))
    target[:, -1] = _np(batch["rot_z"])
    target = np.hstack((box_center[:, :2], target))
    dev_index = next(model.parameters()).device.index or 0
    per_sample = rotate_iou_batched(pred, batch["target_neighbor"], device_id=dev_index, is_3d=is_3d)
    ious = [float(np.max(v)) if len(v) else 0.0 for v in per_sample]
    loss_ori = np.abs(pred[:, -1] - target[:, -1])
    return loss, tb_dict, {"iou": float(np.mean(ious)), "loss_z": float(np.mean(loss_z)),
                           "loss_dim": float(np.mean(loss_dims)), "loss_ori": float(np.mean(loss_ori))}
