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


# ---------------------------------------------------------------------------------------
# batch -> model -> loss adapters (reference :10-29, :90-127, :136-155).  They accept the
# reference's NumPy batch dicts as well as the device batches of DROWDeviceDataset /
# DROWBatchPreprocessor (no host round trip then).
# ---------------------------------------------------------------------------------------
def model_fn(model, batch):
    """Prototype flow network on scan pairs: batch['scan_pair'] [B, 2, N, C], batch['flow_target_flow']."""
    pair = batch["scan_pair"]
    scan1, scan2 = _as_dev_f32(pair[:, 0]), _as_dev_f32(pair[:, 1])
    pred_flow = model(scan1, scan2)
    return model.loss_fn(pred_flow, _as_dev_f32(batch["flow_target_flow"]))[0]


def model_fn_dr_spaam(model, batch):
    """FlowDROW_pretrained step -> (masked flow loss, mean |pred| and mean |target| over the kept points)."""
    cur_scan = _as_dev_f32(batch["scans"][:, -1])
    target, mask = _as_dev_f32(batch["target_flow"]), _as_dev_f32(batch["exclude_mask"])
    _, _, pred_flow = model(_as_dev_f32(batch["input"]), cur_scan)
    loss = model.loss_fn(pred_flow, target, mask=mask)
    keep = mask == 1.0
    return loss, torch.norm(pred_flow, dim=-1)[keep].mean(), torch.norm(target, dim=-1)[keep].mean()


def model_fn_Bb_regression(model, batch):
    """Box head step of the legacy scripts: the loss is on target[:, 2:] (written for a data set whose target
    still carried the box centre; the current JRDBBoxRegressionDataset already strips it -- use
    ``BoundingBoxRegressor.model_fn`` with it, as the pipeline's Trainer does)."""
    pred = model(_as_dev_f32(batch["input"]))
    return model.loss_fn(pred, _as_dev_f32(batch["target"])[:, 2:])


def model_fn_eval(model, eval_loader):
    """-> (mean EPE, mean AAE in degrees) over the loader, per-batch means averaged like the reference."""
    epe = aae = 0.0
    with torch.no_grad():
        for batch in eval_loader:
            _, _, pred_flow = model(_as_dev_f32(batch["input"]), _as_dev_f32(batch["scans"][:, -1]))
            e, a = loss_fn_eval(pred_flow, batch["target_flow"])
            epe += e.mean().item()
            aae += a.mean().item()
    n = len(eval_loader)
    return epe / n, aae / n


def eval_dr_spaam(model, test_loader, cfg=None, output_dir=None, tb_logger=None):
    """Flow evaluation of the DR-SPAAM flow model (reference :221-262): mean EPE over the loader and, per
    sample, EPE / AAE and the predicted / target flow rotated back to the scanner frame -- one batched
    ``pof_rotate_flow`` per tensor instead of the reference's per-sample loop.  The video rendering that
    follows in the reference (:264-306) is visualisation and not rebuilt: with ``output_dir`` the arrays
    are written to ``<output_dir>/flow_eval.npz`` instead.
    -> dict(eval_loss, epe [S], aae [S], pred_flow [S,N,2], target_flow [S,N,2], scans [S,N], odom1)."""
    import os
    model.eval()
    tab = None
    epe_all, aae_all, pred_all, tgt_all, scan_all, odom_all = [], [], [], [], [], []
    total = 0.0
    with torch.no_grad():
        for batch in test_loader:
            cur_scan = _as_dev_f32(batch["scans"][:, -1])
            target = _as_dev_f32(batch["target_flow"])
            _, _, pred = model(_as_dev_f32(batch["input"]), cur_scan)
            pred = pred.float().contiguous()
            e, a = loss_fn_eval(pred, target)
            if tab is None:
                tab = ops.phi_table(num_pts=cur_scan.shape[-1], device=cur_scan.device)
            pred_all.append(ops.rotate_flow(pred, tab, to_canonical=False).cpu().numpy())
            tgt_all.append(ops.rotate_flow(target, tab, to_canonical=False).cpu().numpy())
            epe_all.append(e.cpu().numpy())
            aae_all.append(a.cpu().numpy())
            scan_all.append(cur_scan.cpu().numpy())
            o1 = batch["odom1"]
            odom_all.append(o1.cpu().numpy() if torch.is_tensor(o1) else np.asarray(o1))
            total += e.mean().item()
    res = {"eval_loss": total / max(len(test_loader), 1), "epe": np.concatenate(epe_all), "aae": np.concatenate(aae_all),
           "pred_flow": np.concatenate(pred_all), "target_flow": np.concatenate(tgt_all),
           "scans": np.concatenate(scan_all), "odom1": np.concatenate(odom_all)}
    print("Eval loss: ", res["eval_loss"])
    if tb_logger is not None:
        tb_logger.add_scalar("eval_loss", res["eval_loss"], 0)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        np.savez_compressed(os.path.join(output_dir, "flow_eval.npz"), **res)
    return res
