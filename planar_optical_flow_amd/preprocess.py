"""Batched, device-resident equivalent of ``DROWDataset2.__getitem__`` +
``collate_batch`` (reference ``src/utils/dataset_dr_spaam.py:339-471``).

The reference builds every sample on the CPU in DataLoader workers
(get_regression_target, rphi_to_xy, get_displacement_from_odometry,
global_to_canonical_flow, the two masks, scans_to_cutout) and stacks them with
``np.array([...])``.  Here the whole batch is produced by three launches on the
MI355X, directly in the collated layout, from scan windows that already live in
HBM.  Keys and dtypes of the returned dict follow the reference's batch dict
after its ``.cuda().float()`` hop (eval_utils.py:98-101); masks and flow are
float32 on the device.
"""
import numpy as np
import torch

from . import ops


class DROWBatchPreprocessor:
    """cfg mirrors the reference's flat yaml (config/dr_spaam.yaml):
    cutout_kwargs, pedestrian_only, angle_inc / num_pts of the scanner."""

    def __init__(self, cutout_kwargs=None, pedestrian_only=False, angle_inc=np.radians(0.5), num_pts=450,
                 device="cuda", canonical_flow=True):
        self.cutout_kwargs = dict(cutout_kwargs) if cutout_kwargs else None
        self.pedestrian_only = pedestrian_only
        self.angle_inc, self.num_pts = angle_inc, num_pts
        self.device = torch.device(device)
        self.canonical_flow = canonical_flow
        self.tab = ops.phi_table(angle_inc, num_pts, self.device)
        self._ws = [None, None]      # two per-sample parameter workspaces (look-ahead mode)
        self._cur, self._primed = 0, None

    def make_detections(self, dets_wc, dets_wa, dets_wp):
        """Ragged python/NumPy detection lists (one entry per sample, each a list
        of (r, phi)) -> device CSR in the reference's wc + wa + wp order.  All three classes are kept
        also for pedestrian_only: the dynamic mask of the reference always uses all of them
        (dataset_dr_spaam.py:405), only the regression target ignores wheelchairs and walkers."""
        offs, rphi, cls = [0], [], []
        for wc, wa, wp in zip(dets_wc, dets_wa, dets_wp):
            groups = [(0, wc), (1, wa), (2, wp)]
            n = 0
            for c, d in groups:
                d = np.asarray(d, dtype=np.float64).reshape(-1, 2)
                rphi.append(d)
                cls.append(np.full(len(d), c, dtype=np.uint8))
                n += len(d)
            offs.append(offs[-1] + n)
        rphi = np.concatenate(rphi) if rphi else np.zeros((0, 2))
        cls = np.concatenate(cls) if cls else np.zeros(0, dtype=np.uint8)
        return ops.DetCSR.from_numpy(np.asarray(offs, dtype=np.int32), rphi, cls, self.device)

    def _workspace(self, slot, B, D):
        need = ops.scan_preprocess_workspace_bytes(B, D)
        ws = self._ws[slot]
        if ws is None or ws.numel() < need:
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def _announced(self, odom0, odom1, dets):
        """True when (odom0, odom1, dets) are the very objects the previous call announced as its look-ahead
        (identity, not addresses: the announced tensors are kept alive here, so a recycled allocation can
        never be mistaken for them).  They must not be modified in place in between."""
        p = self._primed
        return p is not None and odom0 is p[0] and odom1 is p[1] and dets is p[2]

    def __call__(self, scans, odom0, odom1, dets, lookahead=None):
        """scans [B,T+1,N] float32 (template rows then the current scan, as
        ``np.vstack((scans, cur_scan))`` in the reference), odom0/odom1 [B,3]
        float64, dets: DetCSR.  Returns the collated batch dict.

        lookahead = (odom0, odom1, dets) of the NEXT batch enables the chained launch: the per-sample
        parameters of the next batch (rigid motions, detection centres) are evaluated on spare workgroups
        of THIS batch's streaming launch, and the next call -- given those same tensors -- is a single
        launch (15 us instead of 20 us per 4096 scans).  Without it every call is self-contained.
        The announced tensors must be passed again as the same objects and left unmodified in between."""
        # pedestrian_only (utils.py:165-168): only persons can be associated (label 1); a zero association
        # radius keeps the other classes out of the target while they still carve the dynamic mask
        labels = (0, 0, 1) if self.pedestrian_only else (1, 2, 3)
        radii = (0.0, 0.0, 0.35) if self.pedestrian_only else (0.6, 0.4, 0.35)
        kw = dict(flow_kind=ops.FLOW_DISPLACEMENT, canonical=self.canonical_flow, labels=labels, assoc_radius=radii,
                  want=("flow", "target_cls", "target_reg", "exclude_mask"))
        if lookahead is None and self._primed is None:
            out = ops.scan_preprocess(scans, self.tab, odom0, odom1, dets, **kw)
        else:
            B, D = odom0.shape[0], int(dets.rphi.shape[0])
            ws = self._workspace(self._cur, B, D)
            if not self._announced(odom0, odom1, dets):
                ops.scan_preprocess(scans, self.tab, odom0, odom1, dets, workspace=ws, phases=1, **kw)
            nb = None
            if lookahead is not None:
                n0, n1, nd = lookahead
                nb = {"odom0": n0, "odom1": n1, "dets": nd,
                      "workspace": self._workspace(1 - self._cur, n0.shape[0], int(nd.rphi.shape[0]))}
            out = ops.scan_preprocess(scans, self.tab, odom0, odom1, dets, workspace=ws, phases=2, next_batch=nb,
                                      **kw)
            self._primed = tuple(lookahead) if lookahead is not None else None
            self._cur = 1 - self._cur
        batch = {
            "scans": scans,
            "target_cls": out["target_cls"],
            "target_reg": out["target_reg"],
            "target_flow": out["flow"],
            "exclude_mask": out["exclude_mask"],
            "phi_grid": self.tab[: self.num_pts],
            "odom1": odom1,
        }
        if self.cutout_kwargs is not None:
            batch["input"] = ops.cutout(scans, self.tab, stride=1, **self.cutout_kwargs)
        return batch


def collate_batch(batch, tensor_keys=("scans", "target_cls", "target_reg", "input", "target_flow",
                                      "exclude_mask", "odom")):
    """``collate_batch`` of the reference (dataset_dr_spaam.py:464-471): stack the
    tensor keys, keep the rest as python lists.  Accepts NumPy arrays or device
    tensors per sample (device tensors are stacked on the device)."""
    out = {}
    for k in batch[0]:
        vals = [s[k] for s in batch]
        if k in tensor_keys:
            out[k] = torch.stack(vals) if isinstance(vals[0], torch.Tensor) else np.array(vals)
        else:
            out[k] = vals
    return out
