"""JRDB feeder pieces on the hot path (reference: src/data_handle/jrdb_handle.py).

``anns_to_segments`` (:178-256) selects, for every annotated box of a frame, the points within
``radius`` of a randomly perturbed box centre.  The reference loops over the annotations with one
NumPy norm each; here all annotations of the frame share ONE launch of ``pof_segment_inputs``
(its mask output), and ``segment_inputs`` additionally returns the fixed-size network inputs.
File parsing (frames_pc_laser.json, labels_3d, PCD) stays host I/O and is not rebuilt.
"""
import numpy as np
import torch

from ... import ops

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
