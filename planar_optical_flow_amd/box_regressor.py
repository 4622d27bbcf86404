"""Drop-in for the reference's top-level ``box_regressor.py``.

``BoxRegressor(ckpt, gpu=True, is_3d=False)``; ``__call__(points, det_center,
det_ori)`` regresses one box, ``regress_batch`` many detections of one frame in
a single forward (the reference is launch bound at one detection per call,
box_regressor.py:43-92).  The radius query and the fixed-size resampling run on
the device; the RNG of the resampling is injectable (the reference uses the
global NumPy state, so parity on the resampled set is up to order).
"""
import numpy as np
import torch

from .src.model.box_regression import BoundingBoxRegressor


class BoxRegressor:
    def __init__(self, ckpt, gpu=True, is_3d=False, seed=None):
        self.cfg = {
            "model": {"type": "box_reg", "input_dim": 4 if is_3d else 3, "target_dim": 5 if is_3d else 3,
                      "dropout": 0.3},
            "segment_radius": 0.4, "result_dir": "./output_box_reg", "min_segment_size": 5, "input_size": 64,
        }
        self.is_3d, self.gpu = is_3d, gpu
        self.device = torch.device("cuda") if gpu else torch.device("cpu")
        model = BoundingBoxRegressor(self.cfg["model"])
        state = ckpt if isinstance(ckpt, dict) else torch.load(ckpt, map_location="cpu", weights_only=False)
        model.load_state_dict(state["model_state"])
        self.model = model.eval().to(self.device)
        self._gen = torch.Generator(device="cpu")
        if seed is not None:
            self._gen.manual_seed(seed)

    # ---- reference API ---------------------------------------------------------------
    def generate_segment(self, points, det_center, radius=0.4):
        """Points within `radius` of the detection centre (:94-105)."""
        p = torch.as_tensor(np.asarray(points), dtype=torch.float64, device=self.device)
        c = torch.as_tensor(np.asarray(det_center), dtype=torch.float64, device=self.device).reshape(1, -1)
        keep = torch.linalg.norm(p - c, dim=1) <= radius
        return p[keep].cpu().numpy()

    def _resample(self, seg):
        """> input_size: random subset; else repeat + pad + shuffle (:61-70)."""
        n, size = seg.shape[0], self.cfg["input_size"]
        perm = torch.randperm(n, generator=self._gen).to(seg.device)
        seg = seg[perm]
        if n > size:
            return seg[:size]
        rep, pad = size // n, size % n
        seg = torch.cat([seg.repeat_interleave(rep, dim=0), seg[:pad]], dim=0)
        return seg[torch.randperm(size, generator=self._gen).to(seg.device)]

    def _prepare(self, points_dev, det_center, det_ori):
        c = torch.as_tensor(np.asarray(det_center), dtype=torch.float64, device=self.device).reshape(1, -1)
        seg = points_dev[torch.linalg.norm(points_dev - c, dim=1) <= self.cfg["segment_radius"]]
        if seg.shape[0] < self.cfg["min_segment_size"]:
            return None
        seg = self._resample(seg) - c
        ori = torch.full((seg.shape[0], 1), float(det_ori), dtype=torch.float64, device=self.device)
        return torch.cat([seg, ori], dim=1).float()

    def _finish(self, pred, det_center, det_ori):
        pred = pred.astype(np.float32)
        det_center = np.asarray(det_center)
        if self.is_3d:
            pred[0] += det_center[-1]
        out = np.hstack((det_center[:2], pred))
        out[-1] = out[-1] + det_ori
        return out

    def __call__(self, points, det_center, det_ori):
        """-> [cx, cy, l, w, rot_z] (or [cx, cy, cz, l, w, h, rot_z]) or None when the
        segment has fewer than min_segment_size points."""
        p = torch.as_tensor(np.asarray(points), dtype=torch.float64, device=self.device)
        x = self._prepare(p, det_center, det_ori)
        if x is None:
            return None
        with torch.no_grad():
            pred = self.model(x[None])[0].cpu().numpy()
        return self._finish(pred, det_center, det_ori)

    # ---- batched form ------------------------------------------------------------------
    def regress_batch(self, points, det_centers, det_oris):
        """All detections of one frame in one forward pass.  Returns a list with one
        entry per detection (None where the segment is too small)."""
        p = torch.as_tensor(np.asarray(points), dtype=torch.float64, device=self.device)
        xs, idx = [], []
        for i, (c, o) in enumerate(zip(det_centers, det_oris)):
            x = self._prepare(p, c, o)
            if x is not None:
                xs.append(x)
                idx.append(i)
        out = [None] * len(det_centers)
        if xs:
            with torch.no_grad():
                pred = self.model(torch.stack(xs)).cpu().numpy()
            for j, i in enumerate(idx):
                out[i] = self._finish(pred[j], det_centers[i], det_oris[i])
        return out
