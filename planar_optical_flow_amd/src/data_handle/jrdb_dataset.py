"""Box-regression data set on the device (reference: src/data_handle/jrdb_dataset.py).

The reference keeps one NumPy segment per sample and, per ``__getitem__``, canonicalises it,
appends a random input angle, randomly drops a quarter of the points (training) and resamples
to ``input_size`` rows -- in DataLoader workers, one sample at a time.  Here all segments live
in one point pool in HBM (CSR offsets) and a whole batch is ONE launch of
``pof_segment_resample``.  The constructor takes the frames ``JRDBHandle`` yields
(``dict(segments, boxes, dets_center)``: a ``jrdb_handle.JRDBHandle`` or any iterable of such dicts) and
repeats the reference's bookkeeping: size filter, orientation wrap, neighbour annotations, one
augmented copy per sample in training.

Two deliberate differences, both places where the reference mutates its stored targets in
place through a slice view (``target = self.targets[idx][2:]`` then ``target[0] = ...`` /
``target[-1] = ...``, :107-131) so that fetching a sample twice changes it: this class
computes the same first-access values from the untouched originals every time.
"""
import numpy as np
import torch

from ... import ops

pi = np.pi
_INPUT_WITH_ANGLE = True    # reference module constant (:14)


def _wrap(angle):
    """One-step wrap into (-pi, pi] as the reference does (:63-66, :222-225)."""
    if angle > pi:
        angle -= 2 * pi
    if angle < -pi:
        angle += 2 * pi
    return angle


def _rot2_f32(phi):
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[c, -s], [s, c]], dtype=np.float32)


class JRDBBoxRegressionDataset:
    def __init__(self, split, cfg, frames=None, device="cuda", rng=None, seed=0):
        if frames is None:              # the reference's two-argument form: read the JRDB tree under cfg["data_dir"]
            from .jrdb_handle import JRDBHandle
            frames = JRDBHandle(split, cfg, device=device, rng=rng)
        self.input_size, self.is_3d, self.mode = cfg["input_size"], cfg["is_3d"], split
        self.augmentation_kwargs = cfg["augmentation_kwargs"]
        self._rng = np.random if rng is None else rng
        self.device = torch.device(device)
        self._seed, self._calls = int(seed), 0
        self.inputs, self.targets, self.targets_neighbor, self.dets_center = [], [], [], []
        augment = self.augmentation_kwargs["use_data_augmentation"] and split == "train"
        for frame in frames:
            boxes = np.asarray(frame["boxes"], dtype=np.float64)
            for segment, box, det_center in zip(frame["segments"], boxes, frame["dets_center"]):
                if not len(segment) > cfg["min_segment_size"]:
                    continue
                box[-1] = _wrap(box[-1])                    # in place, like the reference: `boxes` sees it too
                self._append(np.array(segment), box, det_center, boxes)
                if augment:
                    self._append(*self.data_augmentation(np.array(segment), box, det_center), boxes)
        # device-resident pool
        dev = self.device
        D = 3 if self.is_3d else 2
        lens = [len(x) for x in self.inputs]
        self._max_seg = max(lens) if lens else 0
        pool = np.concatenate([np.asarray(x, np.float64).reshape(-1, D) for x in self.inputs]) if lens \
            else np.zeros((1, D))
        self._points = torch.from_numpy(np.ascontiguousarray(pool)).to(dev)
        self._off = torch.from_numpy(np.cumsum([0] + lens).astype(np.int64)).to(dev)
        self._targets = torch.from_numpy(np.array(self.targets, np.float64).reshape(len(lens), -1)).to(dev)
        self._centers = torch.from_numpy(np.array(self.dets_center, np.float64).reshape(len(lens), -1)).to(dev)

    def _append(self, segment, target, det_center, boxes):
        self.inputs.append(segment)
        self.targets.append(np.array(target, dtype=np.float64))
        self.targets_neighbor.append(self.get_nearby_annotations(np.asarray(target, np.float64), boxes))
        self.dets_center.append(np.asarray(det_center, np.float64))

    def __len__(self):
        return len(self.inputs)

    # ---- reference API -----------------------------------------------------------------
    def data_augmentation(self, input, target, det_center):
        """Random rotation about the box centre, translation and common scaling of the box dimensions
        (:158-230); draws rot, dim, trans in the reference's order."""
        kw, rnd = self.augmentation_kwargs, self._rng
        rot_z = rnd.uniform(-kw["rot_max"] * pi, kw["rot_max"] * pi)
        dim = 1.0 + rnd.uniform(-kw["dim_max"], kw["dim_max"])
        trans = rnd.uniform(-kw["dist_max"], kw["dist_max"], 2)
        rot, centre = _rot2_f32(rot_z), target[:2]
        move = lambda xy: np.matmul(xy - centre, rot.T) + centre + trans
        if self.is_3d:
            input_aug = input.copy()
            input_aug[:, :2] = move(input[:, :2])
            det_aug = np.append(move(det_center[:2]), det_center[-1])
            target_aug = np.hstack((centre + trans, [target[2]], np.asarray(target[3:6]) * dim, [target[-1] - rot_z]))
        else:
            input_aug, det_aug = move(input), move(det_center)
            target_aug = np.hstack((centre + trans, np.asarray(target[2:4]) * dim, [target[-1] - rot_z]))
        target_aug[-1] = _wrap(target_aug[-1])
        return input_aug, target_aug, det_aug

    def get_nearby_annotations(self, target, anns, radius=1.0):
        anns = np.asarray(anns, dtype=np.float64)
        near = anns[np.linalg.norm(anns[:, :3] - target[:3], axis=1) <= radius]
        return np.append(near, target.reshape(1, -1), axis=0)

    def collate_batch(self, batch):
        return {k: np.array([sample[k] for sample in batch]) for k in batch[0]}

    def __getitem__(self, idx):
        """One sample as NumPy arrays (the reference's dict); training code should use get_batch."""
        b = self.get_batch([idx])
        out = {k: (v[0].cpu().numpy() if torch.is_tensor(v) else v[0]) for k, v in b.items()}
        return out

    # ---- batched device path -------------------------------------------------------------
    def get_batch(self, indices):
        """-> dict(input [B, input_size, D+1] float32, target [B, 3|5], rot_z [B], det_center, box_center:
        device tensors; target_neighbor: list of arrays)."""
        dev = self.device
        idx = torch.as_tensor(np.asarray(indices, dtype=np.int64), device=dev)
        tg, ctr = self._targets[idx], self._centers[idx]
        target = tg[:, 2:].clone()
        target[:, 0] = target[:, 0] - ctr[:, -1]            # the reference's `target[0] - det_center[-1]`
        box_center = tg[:, :3].clone() if self.is_3d else tg[:, :2].clone()
        out = {"det_center": ctr, "box_center": box_center, "target_neighbor": [self.targets_neighbor[i] for i in indices]}
        extra = None
        if _INPUT_WITH_ANGLE:
            rot_z = tg[:, -1].clone()
            lim = self.augmentation_kwargs["rot_max"] * pi
            noise = torch.from_numpy(np.asarray(self._rng.uniform(-lim, lim, len(indices)), np.float64)).to(dev)
            extra = (rot_z + noise).contiguous()
            target[:, -1] = rot_z - extra
            out["rot_z"] = rot_z
        drop = self.augmentation_kwargs["random_drop"] \
            if self.augmentation_kwargs["use_data_augmentation"] and self.mode == "train" else 0.0
        # CSR of the batch inside the pool: one launch over (start, length) pairs
        # the batch's segments gathered into a compact pool with its own CSR (device index arithmetic)
        start = self._off[idx]
        lens = self._off[idx + 1] - start
        offs = torch.zeros(len(indices) + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens, 0, out=offs[1:])
        total = int(offs[-1])
        owner = torch.repeat_interleave(torch.arange(len(indices), device=dev), lens, output_size=total)
        rows = start[owner] + (torch.arange(total, device=dev) - offs[owner])
        self._calls += 1
        x, _ = ops.segment_resample(self._points[rows].contiguous(), offs.to(torch.int32), ctr.contiguous(), extra,
                                    random_drop=drop, input_size=self.input_size,
                                    seed=self._seed * 1000003 + self._calls, max_segment=self._max_seg)
        out["input"], out["target"] = x, target
        return out
