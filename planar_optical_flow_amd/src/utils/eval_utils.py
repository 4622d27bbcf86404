"""Flow metrics of the reference's ``src/utils/eval_utils.py`` on the HIP path."""
import numpy as np
import torch

from planar_optical_flow_amd import ops


def _as_dev_f32(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device="cuda", dtype=torch.float32).contiguous()


def loss_fn_eval(pred_flow, target_flow):
    """:129-134 -> (epe_batch [B], aae_batch [B] in degrees), float32 tensors.
    Note the reference's atan2(x, y) argument order, kept by the kernel."""
    p, t = _as_dev_f32(pred_flow), _as_dev_f32(target_flow)
    e, a, c = ops.flow_errors(p, t)
    return (e / c).float(), (a / c * (180.0 / np.pi)).float()


def flow_epe(pred, target, mask=None):
    """Masked mean end-point error (src/depracted/model/dr_spaam.py:22-27) as a
    Python float; evaluation-side twin of the differentiable loss."""
    p, t = _as_dev_f32(pred), _as_dev_f32(target)
    m = None if mask is None else _as_dev_f32(mask)
    e, _, c = ops.flow_errors(p, t, m)
    return (e.sum() / c.sum()).item()
