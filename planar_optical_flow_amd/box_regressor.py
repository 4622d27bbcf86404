"""Drop-in for the reference's top-level ``box_regressor.py``.

``BoxRegressor(ckpt, gpu=True, is_3d=False)``; ``__call__(points, det_center,
det_ori)`` regresses one box, ``regress_batch`` every detection of a frame with
ONE preparation launch (``pof_segment_inputs``: radius query + fixed-size
resampling of all segments) and one forward (the reference is launch bound at
one detection per call, box_regressor.py:43-92).  The resampling RNG is a
counter-based hash of (seed, call counter, detection, point); the reference uses
the global NumPy state, so parity on the resampled rows is on the multiset
(the network is invariant to their order).
"""
import numpy as np
import torch

from . import ops
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
        if not gpu:
            raise NotImplementedError("the segment preparation is a HIP kernel: BoxRegressor needs gpu=True")
        self._seed = 0 if seed is None else int(seed)
        self._calls = 0

    # ---- reference API ---------------------------------------------------------------
    def generate_segment(self, points, det_center, radius=0.4):
        """Points within `radius` of the detection centre (:94-105), in their original order."""
        pts = np.asarray(points, dtype=np.float64)
        c = np.asarray(det_center, dtype=np.float64).reshape(1, -1)
        _, _, mask = ops.segment_inputs(torch.from_numpy(np.ascontiguousarray(pts)).to(self.device),
                                        torch.from_numpy(c).to(self.device),
                                        torch.zeros(1, dtype=torch.float64, device=self.device), radius=radius,
                                        input_size=1, min_segment_size=0, return_mask=True)
        return np.asarray(points)[mask[0].cpu().numpy()]

    def prepare_batch(self, points, det_centers, det_oris):
        """-> (x [S, input_size, D+1] float32 on the device, valid [S] bool): the network inputs of all
        detections of one frame (rows of invalid segments are zero)."""
        pts = torch.as_tensor(np.ascontiguousarray(np.asarray(points, dtype=np.float64)), device=self.device)
        ctr = torch.as_tensor(np.ascontiguousarray(np.asarray(det_centers, dtype=np.float64)).reshape(-1, pts.shape[1]),
                              device=self.device)
        ori = torch.as_tensor(np.asarray(det_oris, dtype=np.float64).reshape(-1), device=self.device)
        self._calls += 1
        x, count = ops.segment_inputs(pts, ctr, ori, radius=self.cfg["segment_radius"],
                                      input_size=self.cfg["input_size"],
                                      min_segment_size=self.cfg["min_segment_size"],
                                      seed=self._seed * 1000003 + self._calls)
        return x, count >= self.cfg["min_segment_size"]

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
        return self.regress_batch(points, [np.asarray(det_center)], [det_ori])[0]

    # ---- batched form ------------------------------------------------------------------
    def regress_batch(self, points, det_centers, det_oris):
        """All detections of one frame: one preparation launch, one forward pass.  Returns a list with
        one entry per detection (None where the segment is too small)."""
        out = [None] * len(det_centers)
        if len(det_centers) == 0:
            return out
        x, valid = self.prepare_batch(points, det_centers, det_oris)
        idx = torch.nonzero(valid).reshape(-1)
        if idx.numel():
            with torch.no_grad():
                pred = self.model(x[idx]).cpu().numpy()
            for j, i in enumerate(idx.cpu().tolist()):
                out[i] = self._finish(pred[j], det_centers[i], det_oris[i])
        return out
