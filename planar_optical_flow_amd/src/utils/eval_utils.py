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


# ---------------------------------------------------------------------------------------
# detector loss (reference :31-88) and the legacy box-head evaluations (:436-641)
# ---------------------------------------------------------------------------------------
def _batch_get(batch, *names):
    for n in names:
        if n in batch:
            return batch[n]
    raise KeyError(names[0])


def model_fn_obj_det(model, batch, rtn_result=False):
    """DROW / SpatialDROW training step -> (total_loss, tb_dict, rtn_dict): classification loss over all points
    (the model's ``cls_loss``; sigmoid + binary form for one-logit models) plus, when the batch has foreground
    points, the mean Euclidean regression error on them.  Reads ``target_flow_cls`` / ``target_flow_reg`` like the
    reference, or the ``target_cls`` / ``target_reg`` keys of DROWDataset2 / the device batches."""
    out = model(_as_dev_f32(batch["input"]))
    pred_cls, pred_reg = out[0], out[1]
    tcls = _batch_get(batch, "target_flow_cls", "target_cls")
    treg = _as_dev_f32(_batch_get(batch, "target_flow_reg", "target_reg"))
    tcls = (torch.from_numpy(tcls) if isinstance(tcls, np.ndarray) else tcls).to(device=pred_cls.device).long()
    n_batch, n_pts = tcls.shape[:2]
    tcls = tcls.reshape(n_batch * n_pts)
    pred_cls = pred_cls.reshape(n_batch * n_pts, -1)
    if pred_cls.shape[1] == 1:
        cls_loss = model.cls_loss(torch.sigmoid(pred_cls.squeeze(-1)), tcls.float(), reduction="mean")
    else:
        cls_loss = model.cls_loss(pred_cls, tcls, reduction="mean")
    total, tb = cls_loss, {"cls_loss": cls_loss.item()}
    fg = tcls.ne(0)
    tb["fg_ratio"] = torch.sum(fg).item() / (n_batch * n_pts)
    pred_reg = pred_reg.reshape(n_batch * n_pts, -1)
    if tb["fg_ratio"] > 0.0:
        err = torch.nn.functional.mse_loss(pred_reg[fg], treg.reshape(n_batch * n_pts, -1)[fg], reduction="none")
        reg_loss = torch.sqrt(torch.sum(err, dim=1)).mean()
        total = total + reg_loss
        tb["reg_loss"] = reg_loss.item()
    rtn = {}
    if rtn_result:
        rtn = {"pred_reg": pred_reg.view(n_batch, n_pts, -1), "pred_cls": pred_cls.view(n_batch, n_pts, -1)}
    return total, tb, rtn


def _paired_iou(boxes, targets):
    """IoU of box i with target i ([B, 5] rows cx, cy, l, w, rot): B groups of one pair, one launch."""
    b, t = _as_dev_f32(boxes), _as_dev_f32(targets)
    return ops.rotate_iou(b[:, None, :], t[:, None, :])[:, 0, 0]


def _box_eval_batch(model, batch, canonical):
    """Legacy target layout [cx, cy, l, w, rot / pi]; -> (loss, iou [B], dimension error [B], orientation error [B])."""
    x, target = _as_dev_f32(batch["input"]), _as_dev_f32(batch["target"])
    center = _as_dev_f32(batch["det_center"])[:, :2]
    pred = model(x)
    loss = model.loss_fn(pred, target[:, 2:])
    pred, target = pred.clone(), target.clone()
    pred[:, -1] *= np.pi
    target[:, -1] *= np.pi
    if canonical:
        boxes = torch.cat((torch.zeros_like(center), pred), dim=1)
    else:
        boxes = torch.cat((center, pred), dim=1)
        target[:, :2] += center
    iou = _paired_iou(boxes, target)
    return loss, iou, (pred[:, :2] - target[:, 2:4]).abs().sum(dim=1), (pred[:, -1] - target[:, -1]).abs()


def model_fn_eval_box_reg(model, eval_loader):
    """-> (mean loss, mean dimension error, mean orientation error, mean IoU), per-batch means averaged (:520-559).
    The IoU of prediction i with target i comes from one paired launch per batch instead of the diagonal of the
    reference's full B x B matrix."""
    model.eval()
    loss_sum, dim, ori, iou = 0.0, [], [], []
    with torch.no_grad():
        for batch in eval_loader:
            loss, i, d, o = _box_eval_batch(model, batch, canonical=True)
            loss_sum += loss
            iou.append(i.mean().item()), dim.append(d.mean().item()), ori.append(o.mean().item())
    return loss_sum / len(eval_loader), np.mean(dim), np.mean(ori), np.mean(iou)


def eval_Bb_regression(model, test_loader, cfg=None, output_dir=None, tb_logger=None):
    """Per-sample box-head evaluation in the scanner frame (:436-518): the reference walks a batch-size-1 loader and
    plots a window of frames; here every sample of every batch is scored (IoU with the box placed at its detection
    centre, |dl| + |dw|, |d rot|) and the arrays are returned (and saved under output_dir) instead of plotted."""
    import os
    model.eval()
    loss_sum, dim, ori, iou = 0.0, [], [], []
    with torch.no_grad():
        for batch in test_loader:
            loss, i, d, o = _box_eval_batch(model, batch, canonical=False)
            loss_sum += float(loss)
            iou.append(i.cpu().numpy()), dim.append(d.cpu().numpy()), ori.append(o.cpu().numpy())
    res = {"eval_loss": loss_sum / max(len(test_loader), 1), "iou": np.concatenate(iou),
           "dimension_err": np.concatenate(dim), "orientation_err": np.concatenate(ori)}
    print("Eval loss: ", res["eval_loss"])
    print("avg dimension error: {:0.3f} [m]".format(res["dimension_err"].mean()))
    print("avg orientation error: {:0.3f} [rad]".format(res["orientation_err"].mean()))
    print("avg IOU: {:0.3f}".format(res["iou"].mean()))
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        np.savez_compressed(os.path.join(output_dir, "box_eval.npz"), **res)
    return res


def eval_BB_reg_baseline(dataset):
    """Baseline of the box head (:561-641): predict the data set's mean length / width at rot = pi / 2 around every
    detection centre and score it against the annotation -- all samples in one paired IoU launch.
    dataset: ``.targets`` rows [cx, cy, l, w, rot], ``.dets_center``.  -> dict(iou, dimension_err, orientation_err)."""
    targets = np.asarray(dataset.targets, dtype=np.float64)[:, :5]
    centers = np.asarray(dataset.dets_center, dtype=np.float64)[:, :2]
    pred = np.array([targets[:, 2].mean(), targets[:, 3].mean(), 0.5 * np.pi])
    boxes = np.hstack((centers, np.broadcast_to(pred, (len(centers), 3))))
    res = {"iou": _paired_iou(boxes, targets).cpu().numpy(),
           "dimension_err": np.abs(pred[:2] - targets[:, 2:4]).sum(axis=1),
           "orientation_err": np.abs(pred[2] - targets[:, 4])}
    print("Eval loss: ", res["dimension_err"].mean() + res["orientation_err"].mean())
    print("avg dimension error: {:0.3f} [m]".format(res["dimension_err"].mean()))
    print("avg orientation error: {:0.3f} [rad]".format(res["orientation_err"].mean()))
    print("avg IOU: {:0.3f}".format(res["iou"].mean()))
    return res


def eval(model, test_loader, cfg=None, output_dir=None, tb_logger=None):     # noqa: A001 -- the reference's name
    """Flow evaluation of the scan-pair (Prototype) network (reference :157-219): predictions shorter than 1e-6 are
    zeroed, the per-sample EPE comes from ``loss_fn_eval`` (HIP), the loss is the mean of the per-batch means.  The
    video rendering is visualisation and not rebuilt: with ``output_dir`` the arrays go to ``flow_eval.npz``.
    -> dict(eval_loss, epe [S], scans [S, N, C], pred_flow, target_flow)."""
    import os
    model.eval()
    total, scans, preds, targets, epes = 0.0, [], [], [], []
    with torch.no_grad():
        for batch in test_loader:
            pair = batch["scan_pair"]
            scan1, scan2 = _as_dev_f32(pair[:, 0]), _as_dev_f32(pair[:, 1])
            target = _as_dev_f32(batch["flow_target_flow"])
            pred = model(scan1, scan2).float().contiguous()
            pred[torch.norm(pred, dim=-1) <= 1e-6] = 0.0
            epe, _ = loss_fn_eval(pred, target)
            total += epe.mean().item()
            scans.append(scan1.cpu().numpy()), preds.append(pred.cpu().numpy())
            targets.append(target.cpu().numpy()), epes.append(epe.cpu().numpy())
    res = {"eval_loss": total / max(len(test_loader), 1), "epe": np.concatenate(epes), "scans": np.concatenate(scans),
           "pred_flow": np.concatenate(preds), "target_flow": np.concatenate(targets)}
    print("Eval loss: ", res["eval_loss"])
    if tb_logger is not None:
        tb_logger.add_scalar("eval_loss", res["eval_loss"], 0)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        np.savez_compressed(os.path.join(output_dir, "flow_eval.npz"), **res)
    return res


def eval_person_flow(model, test_loader, cfg=None, output_dir=None, tb_logger=None):
    """Detection + flow evaluation of a (pred_cls, pred_reg, pred_flow) network (reference :320-434).  The reference
    runs one sample per batch: NMS of the predicted centres on the host, EPE / AAE, flows back to the scanner
    frame, then a video.  Here a batch of any size takes one ``pof_nms`` launch for all its scans, one flow-error
    launch and one rotation launch per tensor; the arrays are returned (and written to ``person_flow_eval.npz``
    under output_dir) instead of rendered.  The model is called as ``model(input)``, or ``model(input, cur_scan)``
    for ``FlowDROW_pretrained``.
    -> dict(eval_loss, epe, aae, pred_flow, target_flow, scans, dets_xy (list), dets_cls (list), instance_masks)."""
    import inspect
    import os
    model.eval()
    takes_scan = len(inspect.signature(model.forward).parameters) > 1 and "cur_scan" in inspect.signature(model.forward).parameters
    tab = None
    total, out = 0.0, {k: [] for k in ("epe", "aae", "pred_flow", "target_flow", "scans", "instance_masks")}
    dets_xy, dets_cls = [], []
    with torch.no_grad():
        for batch in test_loader:
            scans = batch["scans"]
            scan = _as_dev_f32(scans[:, -2] if not takes_scan else scans[:, -1])
            x, target = _as_dev_f32(batch["input"]), _as_dev_f32(batch["target_flow"])
            pred_cls, pred_reg, pred_flow = model(x, scan) if takes_scan else model(x)
            pred_flow = pred_flow.float().contiguous()
            if tab is None:
                tab = ops.phi_table(num_pts=scan.shape[-1], device=scan.device)
            conf = torch.sigmoid(pred_cls[..., 0]).double().contiguous()
            xy, dc, num, inst = ops.nms_predicted_center(scan, tab, conf, pred_reg.double().contiguous(), 0.5)
            for b, m in enumerate(num.cpu().numpy()):
                dets_xy.append(xy[b, :m].cpu().numpy())
                dets_cls.append(dc[b, :m].cpu().numpy().reshape(-1, 1))
            e, a = loss_fn_eval(pred_flow, target)
            total += e.mean().item()
            out["epe"].append(e.cpu().numpy()), out["aae"].append(a.cpu().numpy())
            out["pred_flow"].append(ops.rotate_flow(pred_flow, tab, to_canonical=False).cpu().numpy())
            out["target_flow"].append(ops.rotate_flow(target, tab, to_canonical=False).cpu().numpy())
            out["scans"].append(scan.cpu().numpy()), out["instance_masks"].append(inst.cpu().numpy())
    res = {k: np.concatenate(v) for k, v in out.items()}
    res.update(eval_loss=total / max(len(test_loader), 1), dets_xy=dets_xy, dets_cls=dets_cls)
    print("Eval loss: ", res["eval_loss"])
    if tb_logger is not None:
        tb_logger.add_scalar("eval_loss", res["eval_loss"], 0)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        np.savez_compressed(os.path.join(output_dir, "person_flow_eval.npz"),
                            **{k: v for k, v in res.items() if k not in ("dets_xy", "dets_cls")},
                            dets_count=np.array([len(d) for d in dets_xy]),
                            dets_xy=np.concatenate(dets_xy) if dets_xy else np.zeros((0, 2)))
    return res
