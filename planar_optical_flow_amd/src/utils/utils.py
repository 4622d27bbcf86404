"""Drop-in for the reference's ``src/utils/utils.py`` scan-geometry functions.

Same names, argument order and defaults as the reference (citations are
``src/utils/utils.py:line`` of the reference checkout).  Every function runs on
the MI355X through libpof_hip.so:

* NumPy arguments are copied to the device, processed by the HIP kernel and the
  result is returned as NumPy with the reference's dtype (a convenience path
  for code that still calls per scan -- one launch per call, latency bound);
* torch device tensors are processed in place of residence and tensors are
  returned.  Batched inputs ([B,N] ranges, [B,T,N] windows) are accepted
  wherever the arithmetic is per point -- that is the fast path, see
  ``planar_optical_flow_amd.ops``.

There is no CPU fallback: without a HIP device these functions raise.
"""
import numpy as np
import torch

from planar_optical_flow_amd import ops

_DEFAULT_INC = np.radians(0.5)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("planar_optical_flow_amd needs a HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _is_t(x):
    return isinstance(x, torch.Tensor)


def _to_dev(x, dtype):
    if _is_t(x):
        return x.to(device=_device(), dtype=dtype).contiguous()
    arr = np.ascontiguousarray(x, dtype={torch.float32: np.float32, torch.float64: np.float64,
                                         torch.int32: np.int32}[dtype])
    if not arr.flags.writeable:
        arr = arr.copy()
    return torch.from_numpy(arr).to(_device())


def _grid_of(scan_phi):
    """(angle_inc, N) of a uniformly spaced angle grid given as array/tensor."""
    phi = scan_phi.detach().cpu().numpy() if _is_t(scan_phi) else np.asarray(scan_phi)
    n = phi.shape[-1]
    assert n >= 2, "angle grid needs at least two points"
    inc = float((phi[-1] - phi[0]) / (n - 1))
    return inc, n


def _table_for(scan_phi):
    """Device angle table matching `scan_phi`.  The kernels regenerate the grid
    from (angle_inc, N) exactly as get_laser_phi does; a grid that is not the
    linspace of its end points is rejected."""
    inc, n = _grid_of(scan_phi)
    tab = ops.phi_table(inc, n, _device())
    ref = tab[:n]
    got = _to_dev(scan_phi, torch.float64).reshape(-1)
    if not torch.allclose(ref, got, rtol=0, atol=1e-9):
        raise AssertionError("scan_phi must be a uniform angle grid (get_laser_phi)")
    return tab


# ------------------------------------------------------------------ A1
def get_laser_phi(angle_inc=np.radians(0.5), num_pts=450):
    """:25-29.  Evaluated on the device (bit-identical to numpy.linspace)."""
    return ops.laser_phi(angle_inc, num_pts, _device()).cpu().numpy()


# ------------------------------------------------------------------ A2
def rphi_to_xy(r, phi):
    """:47-48."""
    tensor_in = _is_t(r)
    rr = _to_dev(r, torch.float32 if (tensor_in and r.dtype == torch.float32) or
                 (not tensor_in and np.asarray(r).dtype == np.float32) else torch.float64)
    n = rr.shape[-1] if rr.dim() else 1
    phi_arr = phi.detach().cpu().numpy() if _is_t(phi) else np.asarray(phi, dtype=np.float64)
    if rr.dim() == 0 or phi_arr.ndim == 0 or rr.dtype != torch.float32 or phi_arr.shape[-1] != n or n < 2:
        # scalar / non-grid use (e.g. a single detection): r*cos, r*sin elementwise on the device
        p = _to_dev(phi_arr, torch.float64)
        x, y = rr.double() * torch.cos(p), rr.double() * torch.sin(p)
    else:
        tab = _table_for(phi_arr)
        flat = rr.reshape(-1, n)
        out = ops.scan_preprocess(flat, tab, out_dtype=torch.float64, want=("xy",))["xy"]
        x, y = out[..., 0].reshape(rr.shape), out[..., 1].reshape(rr.shape)
    if tensor_in:
        return x, y
    return x.cpu().numpy(), y.cpu().numpy()


def rphi_to_xy_torch(r, phi):
    """:51-52."""
    return rphi_to_xy(r, phi)


def scan_to_xy(scan, phi=None):
    """:32-36."""
    return rphi_to_xy(scan, get_laser_phi() if phi is None else phi)


def xy_to_rphi(x, y):
    """:39-43."""
    tensor_in = _is_t(x)
    r, p = ops.xy_to_rphi(_to_dev(x, torch.float64), _to_dev(y, torch.float64))
    return (r, p) if tensor_in else (r.cpu().numpy(), p.cpu().numpy())


# ------------------------------------------------------------------ A4
def _rotate(flow, scan_phi, to_canonical, force_f32=False):
    tensor_in = _is_t(flow)
    dt = torch.float32 if (force_f32 or (tensor_in and flow.dtype == torch.float32)) else torch.float64
    f = _to_dev(flow, dt)
    out = ops.rotate_flow(f, _table_for(scan_phi), to_canonical)
    return out if tensor_in else out.cpu().numpy()


def global_to_canonical_flow(flow, scan_phi):
    """:62-75."""
    return _rotate(flow, scan_phi, True)


def canonical_to_global_flow(flow_canonical, scan_phi):
    """:78-89."""
    return _rotate(flow_canonical, scan_phi, False)


def canonical_to_global_flow_torch(flow_canonical, scan_phi):
    """:92-105 (float32; the rotation table stays resident on the device instead
    of being rebuilt and copied every call)."""
    return _rotate(flow_canonical, scan_phi, False, force_f32=True)


# ------------------------------------------------------------------ A5
def global_to_canonical(scan_r, scan_phi, dets_r, dets_phi):
    """:55-59.  Per point: scan_r/scan_phi [N] (or [B,N]), dets_* broadcastable."""
    tensor_in = _is_t(scan_r)
    r = _to_dev(scan_r, torch.float32)
    r2 = r.reshape(-1, r.shape[-1])
    dr = _to_dev(np.broadcast_to(dets_r.cpu().numpy() if _is_t(dets_r) else dets_r, tuple(r.shape)), torch.float64)
    dp = _to_dev(np.broadcast_to(dets_phi.cpu().numpy() if _is_t(dets_phi) else dets_phi, tuple(r.shape)),
                 torch.float64)
    dx, dy = ops.det_to_canonical(r2, _table_for(scan_phi), dr.reshape(r2.shape), dp.reshape(r2.shape))
    dx, dy = dx.reshape(r.shape), dy.reshape(r.shape)
    return (dx, dy) if tensor_in else (dx.cpu().numpy(), dy.cpu().numpy())


def canonical_to_global(scan_r, scan_phi, dx, dy):
    """:109-116."""
    tensor_in = _is_t(scan_r)
    r = _to_dev(scan_r, torch.float32)
    r2 = r.reshape(-1, r.shape[-1])
    ddx = _to_dev(dx, torch.float64).reshape(r2.shape)
    ddy = _to_dev(dy, torch.float64).reshape(r2.shape)
    dr, dp = ops.canonical_to_det(r2, _table_for(scan_phi), ddx, ddy)
    dr, dp = dr.reshape(r.shape), dp.reshape(r.shape)
    return (dr, dp) if tensor_in else (dr.cpu().numpy(), dp.cpu().numpy())


def canonical_to_global_torch(scan_r, scan_phi, dx, dy):
    """:119-126."""
    return canonical_to_global(scan_r, scan_phi, dx, dy)


# ------------------------------------------------------------------ A3
def _flow(kind, r, scan_phi, odom0, odom1, canonical):
    tab = _table_for(scan_phi)
    rr = _to_dev(r, torch.float32).reshape(1, -1)
    o0 = _to_dev(np.asarray(odom0, dtype=np.float64).reshape(1, 3), torch.float64)
    o1 = _to_dev(np.asarray(odom1, dtype=np.float64).reshape(1, 3), torch.float64)
    out = ops.scan_preprocess(rr, tab, o0, o1, flow_kind=kind, canonical=canonical, out_dtype=torch.float64,
                              want=("flow",))
    return out["flow"][0].cpu().numpy()


def _flow_xy(kind, xy, odom0, odom1):
    tensor_in = _is_t(xy)
    p = _to_dev(xy, torch.float64).reshape(1, -1, 2)
    o0 = _to_dev(np.asarray(odom0, dtype=np.float64).reshape(1, 3), torch.float64)
    o1 = _to_dev(np.asarray(odom1, dtype=np.float64).reshape(1, 3), torch.float64)
    out = ops.flow_from_xy(p, o0, o1, kind)[0]
    return out if tensor_in else out.cpu().numpy()


def get_flow_target(scan, scan_phi, odom_0, odom_1, to_canonical=False):
    """:204-229."""
    return _flow(ops.FLOW_TARGET, scan, scan_phi, odom_0, odom_1, to_canonical)


def get_displacement_from_odometry(scan1_xy, odom0, odom1):
    """:639-662."""
    return _flow_xy(ops.FLOW_DISPLACEMENT, scan1_xy, odom0, odom1)


def get_velocity_from_odometry(scan1_xy, odom0, odom1):
    """:609-636."""
    return _flow_xy(ops.FLOW_VELOCITY, scan1_xy, odom0, odom1)


# ------------------------------------------------------------------ A6
def _csr_one(dets, cls_ids):
    d = np.asarray(dets, dtype=np.float64).reshape(-1, 2)
    return ops.DetCSR.from_numpy(np.array([0, len(d)], dtype=np.int32), d,
                                 np.asarray(cls_ids, dtype=np.uint8).reshape(-1), _device())


def closest_detection(scan, scan_phi, dets, radii):
    """:232-256.  `radii` may differ per detection: detections are grouped into at
    most three radius classes (the reference only ever uses three)."""
    if len(dets) == 0:
        return np.zeros_like(scan, dtype=int)
    assert len(dets) == len(radii), "Need to give a radius for each detection!"
    uniq = sorted(set(float(r) for r in radii))
    assert len(uniq) <= 3, "at most three distinct radii are supported per call"
    cls_ids = [uniq.index(float(r)) for r in radii]
    rad3 = (uniq + [uniq[-1]] * 3)[:3]
    out = ops.scan_preprocess(_to_dev(scan, torch.float32).reshape(1, -1), _table_for(scan_phi),
                              dets=_csr_one(dets, cls_ids), assoc_radius=rad3, want=("closest",))
    return out["closest"][0].cpu().numpy()


def get_regression_target(scan, scan_phi, wcs, was, wps, radius_wc=0.6, radius_wa=0.4, radius_wp=0.35,
                          label_wc=1, label_wa=2, label_wp=3, pedestrian_only=False):
    """:147-185 -> (target_cls int64 [N], target_reg float32 [N,2])."""
    if pedestrian_only:
        dets, cls_ids = list(wps), [2] * len(wps)
        labels = (1, 1, 1)
    else:
        dets = list(wcs) + list(was) + list(wps)
        cls_ids = [0] * len(wcs) + [1] * len(was) + [2] * len(wps)
        labels = (label_wc, label_wa, label_wp)
    out = ops.scan_preprocess(_to_dev(scan, torch.float32).reshape(1, -1), _table_for(scan_phi),
                              dets=_csr_one(dets, cls_ids), assoc_radius=(radius_wc, radius_wa, radius_wp),
                              labels=labels, want=("target_cls", "target_reg"))
    return out["target_cls"][0].cpu().numpy(), out["target_reg"][0].cpu().numpy()


# ------------------------------------------------------------------ A8
def scans_to_cutout(scans, scan_phi, stride=1, centered=True, fixed=False, window_width=1.66,
                    window_depth=1.0, num_cutout_pts=48, padding_val=29.99, area_mode=False):
    """:259-334.  scans (T,N) -> (N/stride, T, P) float32; a leading batch axis
    ([B,T,N] -> [B,N/stride,T,P]) is accepted."""
    tensor_in = _is_t(scans)
    s = _to_dev(scans, torch.float32)
    batched = s.dim() == 3
    out = ops.cutout(s if batched else s[None], _table_for(scan_phi), stride=stride, centered=centered,
                     fixed=fixed, window_width=window_width, window_depth=window_depth,
                     num_cutout_pts=num_cutout_pts, padding_val=padding_val, area_mode=area_mode)
    out = out if batched else out[0]
    return out if tensor_in else out.cpu().numpy()


def scans_to_polar_grid(scans, min_range=0.0, max_range=30.0, range_bin_size=1.0, tsdf_clip=1.0,
                        normalize=True):
    """:492-531.  scans (T,N) -> (T, R, N) float32 TSDF columns; a leading batch axis is accepted."""
    tensor_in = _is_t(scans)
    s = _to_dev(scans, torch.float32)
    batched = s.dim() == 3
    out = ops.polar_grid(s if batched else s[None], min_range, max_range, range_bin_size, tsdf_clip, normalize)
    out = out if batched else out[0]
    return out if tensor_in else out.cpu().numpy()


def scans_to_cutout_torch(scans, scan_phi, stride=1, centered=True, fixed=False, window_width=1.66,
                          window_depth=1.0, num_cutout_pts=48, padding_val=29.99, area_mode=False):
    """:337-420.  The reference's torch twin does its index math in float32 and
    disagrees with its own NumPy version; this follows the NumPy (float64) one."""
    return scans_to_cutout(scans, scan_phi, stride, centered, fixed, window_width, window_depth,
                           num_cutout_pts, padding_val, area_mode)


# ------------------------------------------------------------------ A11
def nms_predicted_center(scan_grid, phi_grid, pred_cls, pred_reg, min_dist=0.5):
    """:535-571 -> (det_xys [M,2], det_cls [M,1], instance_mask [N] int32)."""
    pc = pred_cls.detach().cpu().numpy() if _is_t(pred_cls) else np.asarray(pred_cls)
    assert pc.ndim == 2 and pc.shape[1] == 1
    xy, dc, num, inst = ops.nms_predicted_center(
        _to_dev(scan_grid, torch.float32).reshape(1, -1), _table_for(phi_grid),
        _to_dev(pc[:, 0], torch.float64).reshape(1, -1), _to_dev(pred_reg, torch.float64).reshape(1, -1, 2), min_dist)
    m = int(num[0].item())
    return xy[0, :m].cpu().numpy(), dc[0, :m].cpu().numpy().reshape(-1, 1).astype(pc.dtype), inst[0].cpu().numpy()


def flow_to_hsv(flow):
    """:574-584.  Flow vectors [..., 2] -> RGB colours [..., 3] (float64): hue from the direction, saturation from
    min(|flow|, 0.1) / 0.1, value 1.  The polar conversion is the HIP ``xy_to_rphi``; the HSV -> RGB step repeats
    ``colorsys.hsv_to_rgb``'s arithmetic for all vectors at once instead of one Python call per point."""
    fl = flow.detach().cpu().numpy() if _is_t(flow) else np.asarray(flow)
    r, phi = xy_to_rphi(np.ascontiguousarray(fl[..., 0], dtype=np.float64), np.ascontiguousarray(fl[..., 1], dtype=np.float64))
    h = (phi + 2.0 * np.pi) / np.pi / 2
    sat = np.minimum(r, 0.1) / 0.1
    v = np.ones_like(h)
    sector = (h * 6.0).astype(np.int64)             # int(): truncation; h > 0 here
    f = h * 6.0 - sector
    p, q, t = v * (1.0 - sat), v * (1.0 - sat * f), v * (1.0 - sat * (1.0 - f))
    sector = sector % 6
    table = np.stack([np.stack(c, axis=-1) for c in ((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))])
    rgb = np.take_along_axis(table, sector[None, ..., None], axis=0)[0]
    return np.where((sat == 0.0)[..., None], v[..., None], rgb)


def data_augmentation(sample_dict):
    """:129-144.  Host-side random left-right flip (uses the global NumPy RNG like
    the reference)."""
    scans, target_reg = sample_dict["scans"], sample_dict["target_reg"]
    if np.random.rand() < 0.5:
        scans = scans[:, ::-1]
        target_reg[:, 0] = -target_reg[:, 0]
    sample_dict.update({"target_reg": target_reg, "scans": scans})
    return sample_dict


def _phi_to_rotation_matrix(phi, is_3d=False):
    """:601-606.  Rotation about z by `phi` as a float32 matrix (2x2, or 3x3 when is_3d)."""
    c, s = np.cos(phi), np.sin(phi)
    rot = np.eye(3 if is_3d else 2, dtype=np.float32)
    rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1] = c, -s, s, c
    return rot
