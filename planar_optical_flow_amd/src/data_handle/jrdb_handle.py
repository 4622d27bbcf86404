"""JRDB feeder pieces on the hot path (reference: src/data_handle/jrdb_handle.py).

``anns_to_segments`` (:178-256) selects, for every annotated box of a frame, the points within
``radius`` of a randomly perturbed box centre.  The reference loops over the annotations with one
NumPy norm each; here all annotations of the frame share ONE launch of ``pof_segment_inputs``
(its mask output), and ``segment_inputs`` additionally returns the fixed-size network inputs.
``JRDBHandle`` is the file side (:58-176, :266-315): the per-sequence ``frames_pc_laser.json`` /
``labels_3d/<sequence>.json`` indices, the upper-velodyne ``.pcd`` clouds (``pcd_io``; LZF payloads
decoded natively) and the 2-D laser ``.txt`` scans, brought to the base frame and cut into segments.
"""
import json
import os

import numpy as np
import torch

from ... import ops, pcd_io
from ..utils import jrdb_transforms as jt

_LASER_Z = 0.176   # height of the 2-D laser plane in the base frame (:203-206)


def box_is_on_ground(jrdb_ann_dict):
    """:258-264."""
    bottom_h = float(jrdb_ann_dict["box"]["cz"]) - 0.5 * float(jrdb_ann_dict["box"]["h"])
    return bottom_h < -0.69


def pseudo_centers(anns, perturb=0.1, is_3d=True, rng=None):
    """Perturbed box centres: cx + r cos(a), cy + r sin(a) with a ~ U(0, 2 pi), r ~ U(-perturb, perturb),
    drawn per annotation in the reference's order (alpha first).  rng=None uses the global NumPy state
    like the reference."""
    rnd = np.random if rng is None else rng
    out = []
    for ann in anns:
        cx, cy = ann["box"]["cx"], ann["box"]["cy"]
        alpha = rnd.uniform(0, 2 * np.pi)
        r = rnd.uniform(-perturb, perturb)
        c = [cx + r * np.cos(alpha), cy + r * np.sin(alpha)]
        out.append(c + [_LASER_Z] if is_3d else c)
    return np.array(out, dtype=np.float64).reshape(len(anns), 3 if is_3d else 2)


def anns_to_segments(points, anns, radius=0.7, perturb=0.1, is_3d=True, rng=None, device="cuda"):
    """-> (segments list[S] of point arrays, boxes [S, 7|5], dets_center [S, 3|2]).

    points [N, 3]; the query is on the xy plane in both modes (3-D mode keeps the z column of the
    selected points, 2-D mode drops it first)."""
    points = np.asarray(points)
    centers = pseudo_centers(anns, perturb, is_3d, rng)
    if is_3d:
        boxes = np.array([[a["box"][k] for k in ("cx", "cy", "cz", "l", "w", "h", "rot_z")] for a in anns])
        src = points
    else:
        boxes = np.array([[a["box"][k] for k in ("cx", "cy", "l", "w", "rot_z")] for a in anns])
        src = points[:, :2]
    if len(anns) == 0:
        return [], boxes, centers
    xy = torch.from_numpy(np.ascontiguousarray(points[:, :2], dtype=np.float64)).to(device)
    ctr = torch.from_numpy(np.ascontiguousarray(centers[:, :2])).to(device)
    _, _, mask = ops.segment_inputs(xy, ctr, torch.zeros(len(anns), dtype=torch.float64, device=device),
                                    radius=radius, input_size=1, min_segment_size=0, return_mask=True)
    mask = mask.cpu().numpy()
    return [src[m] for m in mask], boxes, centers


# The reference's own train / validation partition of the JRDB training set (:22-55); the order is
# part of the interface because it fixes the flat sample index.
_LOCATIONS_TRAIN = """packard-poster-session-2019-03-20_2 packard-poster-session-2019-03-20_1
clark-center-intersection-2019-02-28_0 huang-lane-2019-02-12_0 jordan-hall-2019-04-22_0
memorial-court-2019-03-16_0 packard-poster-session-2019-03-20_0 clark-center-2019-02-28_1
stlc-111-2019-04-19_0 clark-center-2019-02-28_0 tressider-2019-03-16_0 svl-meeting-gates-2-2019-04-08_1
forbes-cafe-2019-01-22_0 gates-159-group-meeting-2019-04-03_0 huang-basement-2019-01-25_0
svl-meeting-gates-2-2019-04-08_0 tressider-2019-03-16_1 nvidia-aud-2019-04-18_0"""
_LOCATIONS_VAL = """cubberly-auditorium-2019-04-22_0 tressider-2019-04-26_2 gates-to-clark-2019-02-28_1
meyer-green-2019-03-16_0 gates-basement-elevators-2019-01-17_1 huang-2-2019-01-25_0
bytes-cafe-2019-02-07_0 hewlett-packard-intersection-2019-01-24_0 gates-ai-lab-2019-02-08_0"""
_JRDB_TRAIN_SEQUENCES = _LOCATIONS_TRAIN.split()
_JRDB_VAL_SEQUENCES = _LOCATIONS_VAL.split()


def _velodyne_name(frame):
    return os.path.basename(frame["pointclouds"]["upper_velodyne"]["url"])


class JRDBHandle:
    """``JRDBHandle(split, cfg)[i]`` -> the frame's json record plus ``points`` [N, 3] in the base frame,
    ``segments`` (list of point arrays), ``boxes`` and ``dets_center``.

    cfg: data_dir, radius_segment, perturb, is_3d (as the reference); optional ``sequences`` overrides the
    built-in split lists (a subset of a JRDB download, or another data set in the same layout).  "test"
    reads the validation sequences because JRDB ships no test labels (:66-67).  Only frames whose
    velodyne file name has an entry in the sequence's label file are indexed."""

    def __init__(self, split, cfg, device="cuda", rng=None):
        if split not in ("train", "val", "test"):
            raise AssertionError('Invalid split "%s"' % split)
        self.radius_segment, self.perturb, self.is_3d = cfg["radius_segment"], cfg["perturb"], cfg["is_3d"]
        self.device, self._rng = device, rng
        root = os.path.join(os.path.abspath(os.path.expanduser(cfg["data_dir"])), "train_dataset")
        self.data_dir = root
        self.timestamp_dir = os.path.join(root, "timestamps")
        self.pc_label_dir = os.path.join(root, "labels", "labels_3d")
        names = cfg.get("sequences")
        if names is None:
            names = _JRDB_TRAIN_SEQUENCES if split == "train" else _JRDB_VAL_SEQUENCES
        self.sequence_names = list(names)
        print("{} dataset: {} sequences found".format("val" if split == "test" else split, len(self.sequence_names)))
        self.sequence_pc_frames, self.sequence_pc_labels = [], []
        index = []
        for q, name in enumerate(self.sequence_names):
            frames, labels = self._load_one_sequence(name)
            self.sequence_pc_frames.append(frames)
            self.sequence_pc_labels.append(labels)
            index += [(q, k) for k, fr in enumerate(frames) if _velodyne_name(fr) in labels]
        self._index = np.array(index, dtype=np.int64).reshape(-1, 2)

    def __len__(self):
        return len(self._index)

    def __getitem__(self, idx):
        if not -len(self) <= idx < len(self):
            raise IndexError(idx)            # also ends `for frame in handle`
        q, k = self._index[idx]
        frame = dict(self.sequence_pc_frames[q][k])          # shallow copy: the stored record stays as loaded
        points = self.load_points(frame)
        anns = self.sequence_pc_labels[q][_velodyne_name(frame)]
        segments, boxes, centers = self.anns_to_segments(points, anns, radius=self.radius_segment, perturb=self.perturb)
        frame.update(segments=segments, boxes=boxes, dets_center=centers, points=points)
        return frame

    def load_points(self, frame):
        """Sensor file of one frame record -> [N, 3] points in the base frame (:126-139): the velodyne cloud in
        3-D mode; in 2-D mode the laser ranges on a (-pi, pi) grid at z = -0.7 in the laser frame."""
        if self.is_3d:
            return jt.transform_pts_upper_velodyne_to_base(
                self._load_pointcloud(frame["pointclouds"]["upper_velodyne"]["url"])).T
        r = self._load_laser(frame["laser"]["url"])
        phi = np.linspace(-np.pi, np.pi, len(r), dtype=np.float32)
        x, y = r * np.cos(phi), r * np.sin(phi)     # float32 on the host, one scan per frame
        z = np.full(len(r), -0.7, dtype=np.float32)
        return jt.transform_pts_laser_to_base(np.stack((x, y, z), axis=0)).T

    def anns_to_segments(self, points, anns, radius=0.7, perturb=0.1):
        return anns_to_segments(points, anns, radius, perturb, self.is_3d, self._rng, self.device)

    box_is_on_ground = staticmethod(box_is_on_ground)

    # ---- files ---------------------------------------------------------------------------
    def _load_one_sequence(self, seq_name):
        """-> (frame records of frames_pc_laser.json["data"], {velodyne file name: annotations})."""
        with open(os.path.join(self.timestamp_dir, seq_name, "frames_pc_laser.json")) as f:
            frames = json.load(f)["data"]
        with open(os.path.join(self.pc_label_dir, seq_name + ".json")) as f:
            labels = json.load(f)["labels"]
        return frames, labels

    def _load_pointcloud(self, url):
        """-> float32 [3, N]."""
        return pcd_io.read_pcd_xyz(os.path.join(self.data_dir, url))

    def _load_laser(self, url):
        """Whitespace-separated ranges -> float32 [N] (np.loadtxt(dtype=float32) of the reference: text ->
        double -> float32)."""
        with open(os.path.join(self.data_dir, url)) as f:
            return np.array(f.read().split(), dtype=np.float64).astype(np.float32)
