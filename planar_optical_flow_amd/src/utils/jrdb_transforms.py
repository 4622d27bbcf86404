"""JRDB sensor frames (reference: src/utils/jrdb_transforms.py:13-67).

Every frame is x-forward / y-left / z-up and differs from ``base`` (the frame of the 3-D
annotations) by a yaw and a vertical offset only.  Points are ``[3, N]`` arrays; the
matrices are float32 like the reference's, so ``R @ pts`` promotes exactly as it does there.
"""
import numpy as np


def _yaw(rot_z):
    c, s = np.cos(rot_z), np.sin(rot_z)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)


def _lift(z):
    return np.array([0, 0, z], dtype=np.float32).reshape(3, 1)


# sensor -> (rotation to base, translation to base)
_TO_BASE = {
    "laser": (_yaw(np.pi / 120), None),
    "upper_velodyne": (_yaw(0.085), _lift(0.33529)),
    "lower_velodyne": (np.eye(3, dtype=np.float32), _lift(-0.13511)),
}


def _to_base(sensor, pts):
    rot, shift = _TO_BASE[sensor]
    out = rot @ pts
    return out if shift is None else out + shift


def _from_base(sensor, pts):
    rot, shift = _TO_BASE[sensor]
    return rot.T @ (pts if shift is None else pts - shift)


def transform_pts_upper_velodyne_to_base(pts):
    return _to_base("upper_velodyne", pts)


def transform_pts_lower_velodyne_to_base(pts):
    return _to_base("lower_velodyne", pts)


def transform_pts_laser_to_base(pts):
    return _to_base("laser", pts)


def transform_pts_base_to_upper_velodyne(pts):
    return _from_base("upper_velodyne", pts)


def transform_pts_base_to_lower_velodyne(pts):
    return _from_base("lower_velodyne", pts)


def transform_pts_base_to_laser(pts):
    return _from_base("laser", pts)
