"""Batched device API: torch tensors in HBM -> libpof_hip.so (C ABI) -> torch tensors.

torch is used only as the owner of device memory and streams.  Every function
validates shapes/dtypes on the host before a kernel sees them and launches on
torch's current HIP stream.  No CPU fallback exists: a missing library or a CPU
tensor raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_TAB_CACHE = {}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a tensor on the HIP device (no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def phi_table(angle_inc=np.radians(0.5), num_pts=450, device="cuda"):
    """A1: device table [3N] float64 = phi | (cos, sin) interleaved.  Cached."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (float(angle_inc), int(num_pts), device.index)
    tab = _TAB_CACHE.get(key)
    if tab is None:
        tab = torch.empty(3 * num_pts, dtype=torch.float64, device=device)
        with torch.cuda.device(device):
            _lib.call("pof_laser_phi", float(angle_inc), int(num_pts), _ptr(tab), _stream())
        _TAB_CACHE[key] = tab
    return tab


def laser_phi(angle_inc=np.radians(0.5), num_pts=450, device="cuda"):
    return phi_table(angle_inc, num_pts, device)[:num_pts]


class DetCSR:
    """Ragged per-sample detections on the device: offsets [B+1] int32,
    rphi [D,2] float64, cls [D] uint8 (0 wc, 1 wa, 2 wp)."""

    def __init__(self, offsets, rphi, cls):
        self.offsets = _dev(offsets, torch.int32, "det offsets")
        self.rphi = _dev(rphi, torch.float64, "det rphi")
        self.cls = _dev(cls, torch.uint8, "det cls")
        if self.rphi.dim() != 2 or self.rphi.shape[1] != 2 or self.cls.shape[0] != self.rphi.shape[0]:
            raise ValueError("det_rphi must be [D,2] and det_cls [D]")

    @staticmethod
    def from_numpy(offsets, rphi, cls, device="cuda"):
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        if offsets[0] != 0 or np.any(np.diff(offsets) < 0) or offsets[-1] != len(rphi):
            raise ValueError("CSR offsets must start at 0, be non-decreasing and end at D")
        # keep at least one element so data_ptr() is valid for D == 0
        r = np.ascontiguousarray(rphi, dtype=np.float64).reshape(-1, 2)
        c = np.ascontiguousarray(cls, dtype=np.uint8)
        if len(r) == 0:
            r, c = np.zeros((1, 2)), np.zeros(1, dtype=np.uint8)
            d = DetCSR(torch.from_numpy(offsets).to(device), torch.from_numpy(r).to(device),
                       torch.from_numpy(c).to(device))
            return d
        return DetCSR(torch.from_numpy(offsets).to(device), torch.from_numpy(r).to(device),
                      torch.from_numpy(c).to(device))


FLOW_DISPLACEMENT, FLOW_TARGET, FLOW_VELOCITY, FLOW_PREPARED, ALIGN_NEXT_SCAN = 0, 1, 2, 3, 4


def scan_preprocess_workspace_bytes(B, D):
    return int(_lib.load().pof_scan_preprocess_workspace_bytes(int(min(B, 65535)), int(D)))


def scan_preprocess(scans, tab, odom0=None, odom1=None, dets=None, flow_kind=FLOW_DISPLACEMENT,
                    canonical=True, out_dtype=torch.float32, want=("flow",),
                    assoc_radius=(0.6, 0.4, 0.35), labels=(1, 2, 3), dyn_radius=(2.5, 2.0, 2.0),
                    out=None, workspace=None, phases=3, next_batch=None):
    """A2-A7 fused, one launch.

    scans: [B,T,N] (the last row of each window is the current scan) or [B,N].
    want: subset of {"xy","flow","closest","target_cls","target_reg","dyn_mask",
          "valid_mask","exclude_mask"}.  `out` may hold preallocated tensors.
    phases: 3 = both launches; 1 = only the per-sample params launch into
          `workspace`, 2 = only the streaming launch (for two-stream pipelining of
          independent batches; B <= 65535 in that mode).
    next_batch: chained form -- dict(odom0, odom1, dets, workspace) of the NEXT batch; this
          call then streams the current batch (its params must already be in `workspace`,
          i.e. phases is forced to 2) and evaluates the next batch's params in the same launch.
    Returns a dict of device tensors.
    """
    scans = _dev(scans, torch.float32, "scans")
    if scans.dim() == 3:
        B, T, N = scans.shape
        stride, off = T * N, (T - 1) * N
    elif scans.dim() == 2:
        B, N = scans.shape
        stride, off = N, 0
    else:
        raise ValueError("scans must be [B,T,N] or [B,N]")
    tab = _dev(tab, torch.float64, "tab")
    if tab.numel() != 3 * N:
        raise ValueError("angle table is for %d points, scans have %d" % (tab.numel() // 3, N))
    want = set(want)
    known = {"xy", "flow", "closest", "target_cls", "target_reg", "dyn_mask", "valid_mask", "exclude_mask"}
    if not want <= known:
        raise ValueError("unknown outputs: %s" % sorted(want - known))
    need_det = bool(want & {"closest", "target_cls", "target_reg", "dyn_mask", "exclude_mask"})
    if need_det and dets is None:
        raise ValueError("association / dynamic-mask outputs need detections")
    if "flow" in want:
        odom0 = _dev(odom0, torch.float64, "odom0")
        odom1 = _dev(odom1, torch.float64, "odom1")
        if tuple(odom0.shape) != (B, 3) or tuple(odom1.shape) != (B, 3):
            raise ValueError("odometry must be [B,3]")
    if dets is not None and dets.offsets.numel() != B + 1:
        raise ValueError("detection offsets must have B+1 entries")
    if out_dtype not in (torch.float32, torch.float64):
        raise TypeError("out_dtype must be float32 or float64")
    dev = scans.device
    out = {} if out is None else out

    def buf(name, shape, dtype):
        if name not in want:
            return None
        t = out.get(name)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=dev)
            out[name] = t
        else:
            _dev(t, dtype, name)
            if tuple(t.shape) != tuple(shape):
                raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
        return t

    xy = buf("xy", (B, N, 2), out_dtype)
    flow = buf("flow", (B, N, 2), out_dtype)
    closest = buf("closest", (B, N), torch.int64)
    tcls = buf("target_cls", (B, N), torch.int64)
    treg = buf("target_reg", (B, N, 2), torch.float32)
    dyn = buf("dyn_mask", (B, N), torch.float32)
    val = buf("valid_mask", (B, N), torch.float32)
    exc = buf("exclude_mask", (B, N), torch.float32)
    ar = (C.c_double * 3)(*assoc_radius)
    lb = (C.c_int32 * 3)(*labels)
    dr = (C.c_double * 3)(*dyn_radius)
    base = scans.data_ptr() + 4 * off
    D = int(dets.rphi.shape[0]) if need_det else 0
    ws_bytes = _lib.load().pof_scan_preprocess_workspace_bytes(min(B, 65535), D)
    if workspace is None:
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    elif workspace.numel() * workspace.element_size() < ws_bytes or not workspace.is_cuda:
        raise ValueError("workspace must be a device tensor of at least %d bytes" % ws_bytes)
    nxt = None
    if next_batch is not None:
        if B > 65535:
            raise ValueError("chained form needs B <= 65535")
        nxt = _scan_inputs(next_batch, want, flow_kind, assoc_radius, labels, dyn_radius, need_det)
    with torch.cuda.device(dev):
        # grid.y carries the sample index: chunk very large batches
        step = 65535
        for s in range(0, max(B, 1), step):
            n = min(step, B - s)
            if n <= 0:
                break

            def sl(t, per):
                return None if t is None else C.c_void_p(t.data_ptr() + s * per * t.element_size())

            _lib.call(
                "pof_scan_preprocess_chained" if nxt is not None else "pof_scan_preprocess_phase",
                C.c_void_p(base + 4 * s * stride), stride, n, N, _ptr(tab),
                sl(odom0, 3) if "flow" in want else None, sl(odom1, 3) if "flow" in want else None,
                int(flow_kind), int(bool(canonical)), int(out_dtype == torch.float64),
                sl(xy, 2 * N), sl(flow, 2 * N),
                sl(dets.offsets, 1) if need_det else None,
                _ptr(dets.rphi) if need_det else None, _ptr(dets.cls) if need_det else None, D,
                ar, lb, dr, sl(closest, N), sl(tcls, N), sl(treg, 2 * N), sl(dyn, N), sl(val, N),
                sl(exc, N), _ptr(workspace), workspace.numel() * workspace.element_size(),
                C.byref(nxt) if nxt is not None else int(phases), _stream())
    return out


_TABF = {}


def phi_table_f32(tab):
    """[N][2] float32 copy of the angle table's (cos, sin) pairs, each rounded once from its float64 entry --
    exactly what the float32-output preprocess kernel would otherwise convert per point.  Cached per table."""
    key = (tab.data_ptr(), tab.numel(), tab.device)
    t = _TABF.get(key)
    if t is None or t[0]() is not tab:
        import weakref
        n = tab.numel() // 3
        tf = tab[n:].to(torch.float32).contiguous()
        _TABF[key] = (weakref.ref(tab), tf)
        return tf
    return t[1]


def _scan_inputs(nb, want, flow_kind, assoc_radius, labels, dyn_radius, need_det):
    """pof_scan_inputs of one batch whose params are to be evaluated: dict(odom0, odom1, dets, workspace)."""
    nd = nb.get("dets")
    n0, n1 = nb.get("odom0"), nb.get("odom1")
    nws = nb["workspace"]
    nxt = _lib.ScanInputs()
    nxt.want_flow = int("flow" in want)
    if nxt.want_flow:
        n0 = _dev(n0, torch.float64, "next odom0")
        n1 = _dev(n1, torch.float64, "next odom1")
        nxt.odom0, nxt.odom1, nxt.B = n0.data_ptr(), n1.data_ptr(), n0.shape[0]
    else:
        nxt.B = nd.offsets.numel() - 1 if nd is not None else 0
    if need_det and nd is not None:
        nxt.det_offsets, nxt.det_rphi, nxt.det_cls = nd.offsets.data_ptr(), nd.rphi.data_ptr(), nd.cls.data_ptr()
        nxt.D = int(nd.rphi.shape[0])
    nxt.flow_kind = int(flow_kind)
    nxt.assoc_radius = (C.c_double * 3)(*assoc_radius)
    nxt.labels = (C.c_int32 * 3)(*labels)
    nxt.dyn_radius = (C.c_double * 3)(*dyn_radius)
    nxt.workspace = nws.data_ptr()
    nxt.workspace_bytes = nws.numel() * nws.element_size()
    return nxt


SCAN_MAX_SLOTS = _lib.SCAN_MAX_SLOTS     # batches one launch of scan_preprocess_multi may stream (and evaluate params for)


def scan_preprocess_multi(batches, tab, next_batches=(), flow_kind=FLOW_DISPLACEMENT, canonical=True,
                          out_dtype=torch.float32, want=("flow",), assoc_radius=(0.6, 0.4, 0.35), labels=(1, 2, 3),
                          dyn_radius=(2.5, 2.0, 2.0), prepare=False):
    """A2-A7 fused for up to SCAN_MAX_SLOTS (8) batches in ONE launch (pof_scan_preprocess_multi): a loader that runs ahead hands
    over several ring slots at once.

    batches: list of dict(scans [B,T,N] | [B,N], dets (DetCSR or None), out {name: tensor}, workspace) -- the
             workspaces must already hold the batches' params (an earlier call's `next_batches`); every name in
             `want` must be preallocated in `out`.
    next_batches: list of dict(odom0, odom1, dets, workspace) whose params this launch evaluates on extra
             workgroups.  `batches` may be empty (params only: priming the ring).
    prepare: do not launch; return a PreparedScanLaunch that does (the arguments marshalled once).
    """
    want = set(want)
    known = {"xy", "flow", "closest", "target_cls", "target_reg", "dyn_mask", "valid_mask", "exclude_mask"}
    if not want <= known:
        raise ValueError("unknown outputs: %s" % sorted(want - known))
    if len(batches) > _lib.SCAN_MAX_SLOTS or len(next_batches) > _lib.SCAN_MAX_SLOTS:
        raise ValueError("at most %d batches per launch" % _lib.SCAN_MAX_SLOTS)
    need_det = bool(want & {"closest", "target_cls", "target_reg", "dyn_mask", "exclude_mask"})
    tab = _dev(tab, torch.float64, "tab")
    N = tab.numel() // 3
    cur = (_lib.ScanBatch * max(len(batches), 1))()
    keep = []
    for k, bt in enumerate(batches):
        scans = _dev(bt["scans"], torch.float32, "scans")
        if scans.dim() == 3:
            B, T, n = scans.shape
            stride, off = T * n, (T - 1) * n
        elif scans.dim() == 2:
            B, n = scans.shape
            stride, off = n, 0
        else:
            raise ValueError("scans must be [B,T,N] or [B,N]")
        if n != N:
            raise ValueError("angle table is for %d points, scans have %d" % (N, n))
        dets = bt.get("dets")
        if need_det and dets is None:
            raise ValueError("association / dynamic-mask outputs need detections")
        if dets is not None and dets.offsets.numel() != B + 1:
            raise ValueError("detection offsets must have B+1 entries")
        out, ws = bt["out"], bt["workspace"]
        c = cur[k]
        c.ranges, c.sample_stride, c.B = scans.data_ptr() + 4 * off, stride, B
        c.D = int(dets.rphi.shape[0]) if need_det else 0
        c.det_offsets = dets.offsets.data_ptr() if need_det else None
        shapes = {"xy": ((B, N, 2), out_dtype), "flow": ((B, N, 2), out_dtype), "closest": ((B, N), torch.int64),
                  "target_cls": ((B, N), torch.int64), "target_reg": ((B, N, 2), torch.float32),
                  "dyn_mask": ((B, N), torch.float32), "valid_mask": ((B, N), torch.float32),
                  "exclude_mask": ((B, N), torch.float32)}
        for name in known:
            if name in want:
                t = _dev(out[name], shapes[name][1], name)
                if tuple(t.shape) != shapes[name][0]:
                    raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), shapes[name][0]))
                setattr(c, name, t.data_ptr())
        c.workspace, c.workspace_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
        keep.append((scans, out, ws))
    nxt = (_lib.ScanInputs * max(len(next_batches), 1))()
    for k, nb in enumerate(next_batches):
        nxt[k] = _scan_inputs(nb, want, flow_kind, assoc_radius, labels, dyn_radius, need_det)
    tabf = phi_table_f32(tab) if out_dtype == torch.float32 else None
    args = (cur, len(batches), nxt, len(next_batches), N, _ptr(tab), _ptr(tabf), int(flow_kind), int(bool(canonical)),
            int(out_dtype == torch.float64), (C.c_double * 3)(*assoc_radius), (C.c_int32 * 3)(*labels),
            (C.c_double * 3)(*dyn_radius))
    if prepare:
        return PreparedScanLaunch(args, tab.device, (keep, list(next_batches), tab, tabf))
    with torch.cuda.device(tab.device):
        _lib.call("pof_scan_preprocess_multi", *args, _stream())


class PreparedScanLaunch:
    """A marshalled pof_scan_preprocess_multi call (``scan_preprocess_multi(..., prepare=True)``): a loader that cycles
    through a ring of fixed buffers builds the descriptor tables of its (current slots, next slots) combinations once
    and then pays one ctypes call per launch -- a few microseconds of host time instead of the ~100 us the checked,
    per-call marshalling of 8 + 8 batches costs.  The tensors it points at are kept alive by the object."""

    def __init__(self, args, device, keep):
        self._args, self._device, self._keep = args, device, keep
        self._fn = getattr(_lib.load(), "pof_scan_preprocess_multi")

    def __call__(self):
        code = self._fn(*self._args, torch.cuda.current_stream(self._device).cuda_stream)
        if code != _lib.POF_OK:
            _lib.call("pof_scan_preprocess_multi", *self._args, _stream())      # the checked path raises the exception


def flow_from_xy(xy, odom0, odom1, flow_kind=FLOW_DISPLACEMENT, canonical=False, tab=None):
    """A3 on scanner-frame points: xy [B,N,2] float64, odom [B,3] float64 -> flow [B,N,2] float64."""
    xy = _dev(xy, torch.float64, "xy")
    odom0 = _dev(odom0, torch.float64, "odom0")
    odom1 = _dev(odom1, torch.float64, "odom1")
    if xy.dim() != 3 or xy.shape[-1] != 2 or tuple(odom0.shape) != (xy.shape[0], 3) or odom0.shape != odom1.shape:
        raise ValueError("xy must be [B,N,2] and odometry [B,3]")
    if canonical and (tab is None or tab.numel() != 3 * xy.shape[1]):
        raise ValueError("canonical output needs the angle table of this N")
    out = torch.empty_like(xy)
    with torch.cuda.device(xy.device):
        _lib.call("pof_flow_from_xy", _ptr(xy), _ptr(odom0), _ptr(odom1), int(flow_kind), int(bool(canonical)),
                  _ptr(tab), _ptr(out), xy.shape[0], xy.shape[1], _stream())
    return out


def xy_to_rphi(x, y):
    """A2 inverse on float64 device tensors of equal shape."""
    x = _dev(x, torch.float64, "x")
    y = _dev(y, torch.float64, "y")
    if x.shape != y.shape:
        raise ValueError("x and y must have the same shape")
    r, phi = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.call("pof_xy_to_rphi", _ptr(x), _ptr(y), _ptr(r), _ptr(phi), x.numel(), _stream())
    return r, phi


def rotate_flow(flow, tab, to_canonical=True, out=None):
    """A4 on a [B,N,2] (or [N,2]) float32/float64 device tensor."""
    if flow.dtype not in (torch.float32, torch.float64):
        raise TypeError("flow must be float32 or float64")
    flow = _dev(flow, flow.dtype, "flow")
    N = flow.shape[-2]
    B = flow.numel() // (2 * N)
    if flow.shape[-1] != 2 or tab.numel() != 3 * N:
        raise ValueError("flow must be [...,N,2] matching the angle table")
    out = torch.empty_like(flow) if out is None else out
    with torch.cuda.device(flow.device):
        _lib.call("pof_rotate_flow", _ptr(flow), _ptr(out), _ptr(tab), B, N, int(bool(to_canonical)),
                  int(flow.dtype == torch.float64), _stream())
    return out


def det_to_canonical(ranges, tab, det_r, det_phi):
    """A5 forward, batched per point: ranges [B,N] float32, det_* [B,N] float64."""
    ranges = _dev(ranges, torch.float32, "ranges")
    B, N = ranges.shape
    det_r = _dev(det_r, torch.float64, "det_r")
    det_phi = _dev(det_phi, torch.float64, "det_phi")
    dx, dy = torch.empty_like(det_r), torch.empty_like(det_r)
    with torch.cuda.device(ranges.device):
        _lib.call("pof_det_to_canonical", _ptr(ranges), _ptr(tab), _ptr(det_r), _ptr(det_phi), _ptr(dx),
                  _ptr(dy), B, N, _stream())
    return dx, dy


def canonical_to_det(ranges, tab, dx, dy):
    """A5 inverse."""
    ranges = _dev(ranges, torch.float32, "ranges")
    B, N = ranges.shape
    dx = _dev(dx, torch.float64, "dx")
    dy = _dev(dy, torch.float64, "dy")
    r, p = torch.empty_like(dx), torch.empty_like(dx)
    with torch.cuda.device(ranges.device):
        _lib.call("pof_canonical_to_det", _ptr(ranges), _ptr(tab), _ptr(dx), _ptr(dy), _ptr(r), _ptr(p),
                  B, N, _stream())
    return r, p


def cutout(scans, tab, stride=1, centered=True, fixed=False, window_width=1.66, window_depth=1.0,
           num_cutout_pts=48, padding_val=29.99, area_mode=False, out=None, return_debug=False,
           exact_values=True, out_dtype=torch.float32, workspace=None):
    """A8 for a batch: scans [B,T,N] float32 -> [B, ceil(N/stride), T, P] float32.
    exact_values=False selects the float32 value path (exact indices, values within 1e-5);
    out_dtype=torch.float16 stores the result as float16 (BASELINE config 5).
    workspace: optional caller-owned int32 tensor of >= min(B, 65535) elements (the per-sample area maxima); a
    caller that captures this call in a hipGraph passes one it allocated BEFORE the capture, so that the graph
    holds no allocation of its own for it (streaming.StreamingDetector)."""
    if out_dtype not in (torch.float32, torch.float16):
        raise TypeError("out_dtype must be float32 or float16")
    scans = _dev(scans, torch.float32, "scans")
    if scans.dim() != 3:
        raise ValueError("scans must be [B,T,N]")
    B, T, N = scans.shape
    if tab.numel() != 3 * N:
        raise ValueError("angle table does not match N")
    Ns = (N + stride - 1) // stride
    P = int(num_cutout_pts)
    if out is None:
        out = torch.empty((B, Ns, T, P), dtype=out_dtype, device=scans.device)
    else:
        _dev(out, out_dtype, "out")
        if tuple(out.shape) != (B, Ns, T, P):
            raise ValueError("out has the wrong shape")
    entry = "pof_cutout_ex" if out_dtype == torch.float32 else "pof_cutout_f16"
    dbg = torch.empty((B, P, T, Ns), dtype=torch.int32, device=scans.device) if return_debug else None
    dbg_area = None
    with torch.cuda.device(scans.device):
        step = 65535
        for s in range(0, B, step):
            n = min(step, B - s)
            if workspace is not None:
                ws = _dev(workspace, torch.int32, "workspace")
                if ws.numel() < n:
                    raise ValueError("workspace must hold at least %d int32 elements" % n)
            else:
                ws = torch.empty(max(n, 1), dtype=torch.int32, device=scans.device)
            _lib.call(entry, _ptr(scans[s:s + n]), n, T, N, _ptr(tab), int(stride), int(bool(centered)),
                      int(bool(fixed)), float(window_width), float(window_depth), P, float(padding_val),
                      int(bool(area_mode)), 0 if exact_values else 1, _ptr(out[s:s + n]), _ptr(ws),
                      _ptr(dbg[s:s + n]) if dbg is not None else None, _stream())
            if return_debug:
                dbg_area = ws
    if return_debug:
        return out, {"lo": dbg, "s_area": dbg_area}
    return out


def conv3_bn_lrelu(x, wt, scale, shift, pool=False, negative_slope=0.1, out=None):
    """N2 trunk layer (inference): x [S,Ci,L] f32, wt [3,Ci,Co] f32 (conv weight transposed), scale/shift [Co]
    (folded BatchNorm + bias) -> [S, Co, L//2 if pool else L]."""
    x = _dev(x, torch.float32, "x")
    wt = _dev(wt, torch.float32, "wt")
    scale = _dev(scale, torch.float32, "scale")
    shift = _dev(shift, torch.float32, "shift")
    S, Ci, L = x.shape
    if wt.dim() != 3 or wt.shape[0] != 3 or wt.shape[1] != Ci:
        raise ValueError("wt must be [3, Ci, Co]")
    Co = wt.shape[2]
    if scale.numel() != Co or shift.numel() != Co:
        raise ValueError("scale / shift must have Co entries")
    Lout = L // 2 if pool else L
    if out is None:
        out = torch.empty((S, Co, Lout), dtype=torch.float32, device=x.device)
    else:
        _dev(out, torch.float32, "out")
    if S > 0:
        with torch.cuda.device(x.device):
            _lib.call("pof_conv3_bn_lrelu", _ptr(x), _ptr(wt), _ptr(scale), _ptr(shift), S, Ci, Co, L,
                      int(bool(pool)), float(negative_slope), _ptr(out), _stream())
    return out


def conv3_first_two(x, l1, wt, scale, shift, slope1=0.1, pool=False, negative_slope=0.1, out=None):
    """The trunk's first two units in one launch (inference): x [S,L] or [S,1,L] f32 (single-channel cutouts), l1 [C1,4]
    f32 = the first unit as {a0, a1, a2, b} per channel (taps x folded BatchNorm scale, shift), wt [3,C1,Co] /
    scale / shift [Co] = the second unit -> [S, Co, L//2 if pool else L]."""
    x = _dev(x, torch.float32, "x")
    if x.dim() == 3 and x.shape[1] == 1:
        x = x.view(x.shape[0], x.shape[2])
    if x.dim() != 2:
        raise ValueError("x must be [S, L] or [S, 1, L]")
    l1 = _dev(l1, torch.float32, "l1")
    wt = _dev(wt, torch.float32, "wt")
    scale = _dev(scale, torch.float32, "scale")
    shift = _dev(shift, torch.float32, "shift")
    S, L = x.shape
    if l1.dim() != 2 or l1.shape[1] != 4 or l1.shape[0] > 128:
        raise ValueError("l1 must be [C1 <= 128, 4]")
    C1 = l1.shape[0]
    if wt.dim() != 3 or wt.shape[0] != 3 or wt.shape[1] != C1:
        raise ValueError("wt must be [3, C1, Co]")
    Co = wt.shape[2]
    if scale.numel() != Co or shift.numel() != Co:
        raise ValueError("scale / shift must have Co entries")
    if pool and L % 2:
        raise ValueError("pooled output needs an even L")
    if out is None:
        out = torch.empty((S, Co, L // 2 if pool else L), dtype=torch.float32, device=x.device)
    else:
        _dev(out, torch.float32, "out")
    if S > 0:
        with torch.cuda.device(x.device):
            _lib.call("pof_conv3_first_two", _ptr(x), _ptr(l1), float(slope1), _ptr(wt), _ptr(scale), _ptr(shift), S, C1, Co,
                      L, int(bool(pool)), float(negative_slope), _ptr(out), _stream())
    return out


def conv1d_bn_lrelu(x, wt, scale, shift, stride=1, pool=False, negative_slope=0.1, out=None):
    """Conv1d(kernel 1 | 3, padding kernel // 2, stride 1 | 2) + folded BatchNorm + LeakyReLU [+ max_pool1d(2)] on the
    float32-MFMA implicit-GEMM kernel: x [S,Ci,L] f32, wt [kernel,Ci,Co] f32 (conv weight transposed), scale / shift
    [Co] -> [S, Co, Lc] with Lc = L (stride 1) | (L + 1) // 2 (stride 2), halved when pooled."""
    x = _dev(x, torch.float32, "x")
    wt = _dev(wt, torch.float32, "wt")
    scale = _dev(scale, torch.float32, "scale")
    shift = _dev(shift, torch.float32, "shift")
    S, Ci, L = x.shape
    if wt.dim() != 3 or wt.shape[0] not in (1, 3) or wt.shape[1] != Ci:
        raise ValueError("wt must be [1 | 3, Ci, Co]")
    K, Co = int(wt.shape[0]), int(wt.shape[2])
    if scale.numel() != Co or shift.numel() != Co:
        raise ValueError("scale / shift must have Co entries")
    Lc = L if stride == 1 else (L + 1) // 2
    Lout = Lc // 2 if pool else Lc
    if out is None:
        out = torch.empty((S, Co, Lout), dtype=torch.float32, device=x.device)
    else:
        _dev(out, torch.float32, "out")
    if S > 0:
        with torch.cuda.device(x.device):
            _lib.call("pof_conv1d_bn_lrelu", _ptr(x), _ptr(wt), _ptr(scale), _ptr(shift), S, Ci, Co, L, K, int(stride),
                      int(bool(pool)), float(negative_slope), _ptr(out), _stream())
    return out


def drow_heads(feat, w_cls, b_cls, w_reg, b_reg):
    """N2 heads (inference): feat [S,C,L] f32, w_cls [n_cls,C], w_reg [2,C] -> (pred_cls [S,n_cls], pred_reg [S,2]):
    mean over positions + both 1x1 convolutions in one launch."""
    feat = _dev(feat, torch.float32, "feat")
    S, C, L = feat.shape
    w_cls = _dev(w_cls.reshape(-1, C), torch.float32, "w_cls")
    w_reg = _dev(w_reg.reshape(-1, C), torch.float32, "w_reg")
    b_cls, b_reg = _dev(b_cls, torch.float32, "b_cls"), _dev(b_reg, torch.float32, "b_reg")
    n_cls = w_cls.shape[0]
    if w_reg.shape[0] != 2 or b_cls.numel() != n_cls or b_reg.numel() != 2:
        raise ValueError("heads: w_cls [n_cls, C], b_cls [n_cls], w_reg [2, C], b_reg [2]")
    pred_cls = torch.empty((S, n_cls), dtype=torch.float32, device=feat.device)
    pred_reg = torch.empty((S, 2), dtype=torch.float32, device=feat.device)
    if S > 0:
        with torch.cuda.device(feat.device):
            _lib.call("pof_drow_heads", _ptr(feat), S, C, L, _ptr(w_cls), _ptr(b_cls), n_cls, _ptr(w_reg), _ptr(b_reg),
                      _ptr(pred_cls), _ptr(pred_reg), _stream())
    return pred_cls, pred_reg


def conv3_wgrad_supported(S, Ci, Co, L, kernel_size=3):
    return S > 0 and int(_lib.load().pof_conv1d_wgrad_workspace_bytes(int(S), int(Ci), int(Co), int(L),
                                                                      int(kernel_size))) > 0


def conv3_wgrad(x, dy, kernel_size=3):
    """N2 training: weight gradient of Conv1d(k = 3 | 1, pad = k // 2): x [S,Ci,L] f32, dy [S,Co,L] f32 ->
    dw [Co,Ci,k] f32."""
    x = _dev(x, torch.float32, "x")
    dy = _dev(dy, torch.float32, "dy")
    S, Ci, L = x.shape
    if dy.dim() != 3 or dy.shape[0] != S or dy.shape[2] != L:
        raise ValueError("dy must be [S, Co, L]")
    if kernel_size not in (1, 3):
        raise ValueError("kernel_size must be 1 or 3")
    Co = dy.shape[1]
    nbytes = int(_lib.load().pof_conv1d_wgrad_workspace_bytes(S, Ci, Co, L, kernel_size))
    if nbytes == 0:
        raise ValueError("conv3_wgrad: unsupported shape S=%d Ci=%d Co=%d L=%d" % (S, Ci, Co, L))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((Co, Ci, kernel_size), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pof_conv1d_wgrad", _ptr(x), _ptr(dy), S, Ci, Co, L, kernel_size, _ptr(dw), _ptr(ws), nbytes,
                  _stream())
    return dw


def regression_loss2(pred, target, alpha=0.5, with_grad=True):
    """configs[3] loss: pred, target [B,3|5] f32 -> (loss [] f32, d loss / d pred [B,T] f32 or None) in one launch."""
    pred = _dev(pred, torch.float32, "pred")
    target = _dev(target, torch.float32, "target")
    if pred.dim() != 2 or tuple(target.shape) != tuple(pred.shape) or pred.shape[1] not in (3, 5) or pred.shape[0] < 1:
        raise ValueError("pred and target must be [B, 3] or [B, 5] with B >= 1")
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if with_grad else None
    with torch.cuda.device(pred.device):
        _lib.call("pof_regression_loss2", _ptr(pred), _ptr(target), pred.shape[0], pred.shape[1], float(alpha), _ptr(loss),
                  _ptr(dpred), _stream())
    return loss, dpred


def linear_bias(x, weight, bias=None, out=None):
    """configs[3] dense layers: x [B,K] f32, weight [N,K] f32 (torch.nn.Linear's), bias [N] f32 or None -> x @ weight.T
    + bias [B,N] on the small-batch MFMA kernel (K a multiple of 4)."""
    x = _dev(x, torch.float32, "x")
    weight = _dev(weight, torch.float32, "weight")
    if x.dim() != 2 or weight.dim() != 2 or weight.shape[1] != x.shape[1]:
        raise ValueError("x must be [B, K] and weight [N, K]")
    B, K = x.shape
    N = weight.shape[0]
    if K % 4:
        raise ValueError("linear_bias: K = %d is not a multiple of 4" % K)
    if bias is not None and _dev(bias, torch.float32, "bias").numel() != N:
        raise ValueError("bias must have N entries")
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, N) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError("out must be a contiguous float32 [B, N] tensor on x's device")
    if B == 0:
        return out
    with torch.cuda.device(x.device):
        _lib.call("pof_linear_bias", _ptr(x), _ptr(weight), _ptr(bias), B, K, N, _ptr(out), _stream())
    return out


def bn_lrelu_pool_supported(S, C, L, pool=False, groups=1):
    """True when the fused training tail covers this shape (L <= 256, C*L % 4 == 0, S % groups == 0; pool = True / 1:
    max over pairs, L even; pool = 2: max over the whole row, L a power of two >= 4)."""
    if int(pool) == 1 and (L & 1):
        return False
    if int(pool) == 2 and (L < 4 or L & (L - 1)):
        return False
    return S > 0 and int(_lib.load().pof_bn_lrelu_pool_workspace_bytes(int(S), int(C), int(L), int(groups))) > 0


def _bn_workspace(S, C, L, groups, device):
    nbytes = int(_lib.load().pof_bn_lrelu_pool_workspace_bytes(int(S), int(C), int(L), int(groups)))
    if nbytes == 0:
        raise ValueError("bn_lrelu_pool: unsupported shape S=%d C=%d L=%d groups=%d (needs L <= 256, C*L %% 4 == 0, "
                         "S %% groups == 0)" % (S, C, L, groups))
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device), nbytes


def bn_lrelu_pool_forward(y, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5,
                          negative_slope=0.1, pool=False, groups=1):
    """N2 training tail: y [S,C,L] f32 (convolution output) -> (z [S,C,L or L//2], save_mean [groups*C],
    save_invstd [groups*C]); z = max_pool1d?(leaky_relu(batch_norm_train(y))), the batch statistics taken separately
    over each of `groups` equal contiguous ranges of the sequences.  running_mean / running_var are updated in
    place (once per group, in order).  pool = 2: the maximum over the whole row, z [S,C] (the PointNet's max over
    points)."""
    y = _dev(y, torch.float32, "y")
    gamma = _dev(gamma, torch.float32, "gamma")
    beta = _dev(beta, torch.float32, "beta")
    S, C, L = y.shape
    if gamma.numel() != C or beta.numel() != C:
        raise ValueError("gamma / beta must have C entries")
    for name, t in (("running_mean", running_mean), ("running_var", running_var)):
        if t is not None and (_dev(t, torch.float32, name).numel() != C):
            raise ValueError("%s must have C entries" % name)
    ws, nbytes = _bn_workspace(S, C, L, groups, y.device)
    pool = int(pool)
    if not bn_lrelu_pool_supported(S, C, L, pool, groups):
        raise ValueError("bn_lrelu_pool: pool mode %d does not take L = %d" % (pool, L))
    out = torch.empty((S, C) if pool == 2 else (S, C, L // 2 if pool else L), dtype=torch.float32, device=y.device)
    mean = torch.empty(groups * C, dtype=torch.float32, device=y.device)
    invstd = torch.empty(groups * C, dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        _lib.call("pof_bn_lrelu_pool_forward", _ptr(y), S, C, L, int(groups), _ptr(gamma), _ptr(beta),
                  _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), float(negative_slope),
                  pool, _ptr(out), _ptr(mean), _ptr(invstd), _ptr(ws), nbytes, _stream())
    return out, mean, invstd


def bn_lrelu_pool_backward(y, dz, gamma, beta, save_mean, save_invstd, negative_slope=0.1, pool=False,
                           bias_grad=False, groups=1):
    """Backward of bn_lrelu_pool_forward: -> (dy [S,C,L], dgamma [C], dbeta [C]) and, with ``bias_grad``, also
    sum(dy) over (S, L) [C] -- the gradient of the convolution bias in front of the BatchNorm."""
    y = _dev(y, torch.float32, "y")
    dz = _dev(dz, torch.float32, "dz")
    S, C, L = y.shape
    pool = int(pool)
    if not bn_lrelu_pool_supported(S, C, L, pool, groups):
        raise ValueError("bn_lrelu_pool: pool mode %d does not take L = %d" % (pool, L))
    if tuple(dz.shape) != ((S, C) if pool == 2 else (S, C, L // 2 if pool else L)):
        raise ValueError("dz must be [S, C%s]" % ("" if pool == 2 else ", L//2" if pool else ", L"))
    for name, t, n in (("gamma", gamma, C), ("beta", beta, C), ("save_mean", save_mean, groups * C),
                       ("save_invstd", save_invstd, groups * C)):
        if _dev(t, torch.float32, name).numel() != n:
            raise ValueError("%s must have %d entries" % (name, n))
    ws, nbytes = _bn_workspace(S, C, L, groups, y.device)
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=y.device)
    dbias = torch.empty(C, dtype=torch.float32, device=y.device) if bias_grad else None
    with torch.cuda.device(y.device):
        _lib.call("pof_bn_lrelu_pool_backward", _ptr(y), _ptr(dz), S, C, L, int(groups), _ptr(gamma), _ptr(beta),
                  _ptr(save_mean), _ptr(save_invstd), float(negative_slope), pool, _ptr(dy), _ptr(dgamma),
                  _ptr(dbeta), _ptr(dbias), _ptr(ws), nbytes, _stream())
    return (dy, dgamma, dbeta, dbias) if bias_grad else (dy, dgamma, dbeta)


def segment_inputs(points, centers, oris, radius=0.4, input_size=64, min_segment_size=5, seed=0,
                   return_mask=False):
    """N3: points [Np,D] f64, centers [S,D] f64, oris [S] f64 -> (x [S,input_size,D+1] f32, count [S] i32
    [, mask [S,Np] bool]): radius query + fixed-size resampling of every detection's segment in one launch."""
    points = _dev(points, torch.float64, "points")
    centers = _dev(centers, torch.float64, "centers")
    oris = _dev(oris, torch.float64, "oris")
    if points.dim() != 2 or centers.dim() != 2 or points.shape[1] != centers.shape[1]:
        raise ValueError("points [Np,D] and centers [S,D] must share D")
    Np, D = points.shape
    S = centers.shape[0]
    if oris.numel() != S:
        raise ValueError("one orientation per detection")
    dev = points.device
    x = torch.zeros((S, int(input_size), D + 1), dtype=torch.float32, device=dev)
    count = torch.zeros((S,), dtype=torch.int32, device=dev)
    mask = torch.zeros((S, Np), dtype=torch.uint8, device=dev) if return_mask else None
    if S > 0 and Np > 0:
        with torch.cuda.device(dev):
            _lib.call("pof_segment_inputs", _ptr(points), Np, D, _ptr(centers), _ptr(oris), S, float(radius),
                      int(input_size), int(min_segment_size), int(seed) & 0xFFFFFFFF, _ptr(x), _ptr(count),
                      _ptr(mask) if mask is not None else None, _stream())
    if return_mask:
        return x, count, mask.bool()
    return x, count


def segment_resample(points, seg_offsets, centers, extra=None, random_drop=0.0, input_size=64, seed=0,
                     max_segment=None):
    """Box-head training feeder: points [P,D] f64 pool, seg_offsets [S+1] int32 CSR, centers [S,D] f64,
    extra [S] f64 or None -> (x [S,input_size,D(+1)] f32, count [S] int32)."""
    points = _dev(points, torch.float64, "points")
    seg_offsets = _dev(seg_offsets, torch.int32, "seg_offsets")
    centers = _dev(centers, torch.float64, "centers")
    if extra is not None:
        extra = _dev(extra, torch.float64, "extra")
    D = points.shape[1]
    S = seg_offsets.numel() - 1
    if centers.shape != (S, D):
        raise ValueError("centers must be [S, D]")
    if max_segment is None:
        max_segment = int((seg_offsets[1:] - seg_offsets[:-1]).max().item()) if S > 0 else 0
    W = D + (0 if extra is None else 1)
    x = torch.zeros((S, int(input_size), W), dtype=torch.float32, device=points.device)
    count = torch.zeros((S,), dtype=torch.int32, device=points.device)
    if S > 0:
        with torch.cuda.device(points.device):
            _lib.call("pof_segment_resample", _ptr(points), D, _ptr(seg_offsets), S, int(max_segment), _ptr(centers),
                      _ptr(extra) if extra is not None else None, float(random_drop), int(input_size),
                      int(seed) & 0xFFFFFFFF, _ptr(x), _ptr(count), _stream())
    return x, count


def polar_grid(scans, min_range=0.0, max_range=30.0, range_bin_size=1.0, tsdf_clip=1.0, normalize=True, out=None):
    """N4 for a batch: scans [B,T,N] float32 -> [B, T, R, N] float32, R = int((max-min)/bin) + 1."""
    scans = _dev(scans, torch.float32, "scans")
    if scans.dim() != 3:
        raise ValueError("scans must be [B,T,N]")
    B, T, N = scans.shape
    R = int((max_range - min_range) / range_bin_size) + 1
    if out is None:
        out = torch.empty((B, T, R, N), dtype=torch.float32, device=scans.device)
    else:
        _dev(out, torch.float32, "out")
        if tuple(out.shape) != (B, T, R, N):
            raise ValueError("out has the wrong shape")
    with torch.cuda.device(scans.device):
        _lib.call("pof_polar_grid", _ptr(scans), B, T, N, float(min_range), float(max_range), float(range_bin_size),
                  float(tsdf_clip), int(bool(normalize)), _ptr(out), _stream())
    return out


def nms_predicted_center(ranges, tab, pred_cls, pred_reg, min_dist=0.5):
    """A11 batched: ranges [B,N] f32, pred_cls [B,N] f64, pred_reg [B,N,2] f64 ->
    (det_xy [B,N,2], det_cls [B,N], num_det [B] int32, instance_mask [B,N] int32)."""
    ranges = _dev(ranges, torch.float32, "ranges")
    B, N = ranges.shape
    pred_cls = _dev(pred_cls, torch.float64, "pred_cls")
    pred_reg = _dev(pred_reg, torch.float64, "pred_reg")
    if tuple(pred_cls.shape) != (B, N) or tuple(pred_reg.shape) != (B, N, 2):
        raise AssertionError("pred_cls must be [B,N] and pred_reg [B,N,2]")
    dev = ranges.device
    det_xy = torch.zeros((B, N, 2), dtype=torch.float64, device=dev)
    det_cls = torch.zeros((B, N), dtype=torch.float64, device=dev)
    num = torch.zeros(B, dtype=torch.int32, device=dev)
    inst = torch.zeros((B, N), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pof_nms_predicted_center", _ptr(ranges), _ptr(tab), _ptr(pred_cls), _ptr(pred_reg),
                  float(min_dist), B, N, _ptr(det_xy), _ptr(det_cls), _ptr(num), _ptr(inst), None, 0,
                  _stream())
    return det_xy, det_cls, num, inst


def flow_errors(pred, target, mask=None):
    """A12 reductions: returns (epe_sum [B], aae_sum [B] radians, count [B]) float64."""
    pred = _dev(pred, torch.float32, "pred")
    target = _dev(target, torch.float32, "target")
    if pred.shape != target.shape or pred.dim() != 3 or pred.shape[-1] != 2:
        raise ValueError("pred/target must be [B,N,2]")
    B, N = pred.shape[:2]
    if mask is not None:
        mask = _dev(mask, torch.float32, "mask")
        if tuple(mask.shape) != (B, N):
            raise ValueError("mask must be [B,N]")
    dev = pred.device
    e = torch.empty(B, dtype=torch.float64, device=dev)
    a = torch.empty(B, dtype=torch.float64, device=dev)
    c = torch.empty(B, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pof_flow_errors", _ptr(pred), _ptr(target), _ptr(mask), B, N, _ptr(e), _ptr(a), _ptr(c),
                  _stream())
    return e, a, c


def band_correlation(feat1, feat2, kernel_size=3, max_displacement=5, out=None):
    """A9: [B,C,n] x2 float32 (or float16 storage, BASELINE config 5) -> [B, 2*max_displacement+1, n] float32."""
    half = isinstance(feat1, torch.Tensor) and feat1.dtype == torch.float16
    feat1 = _dev(feat1, torch.float16 if half else torch.float32, "feat1")
    feat2 = _dev(feat2, torch.float16 if half else torch.float32, "feat2")
    entry = "pof_band_correlation_f16" if half else "pof_band_correlation"
    if feat1.shape != feat2.shape or feat1.dim() != 3:
        raise ValueError("features must be two [B,C,n] tensors of equal shape")
    B, Cc, n = feat1.shape
    D = 2 * max_displacement + 1
    if out is None:
        out = torch.empty((B, D, n), dtype=torch.float32, device=feat1.device)
    with torch.cuda.device(feat1.device):
        step = 65535
        for s in range(0, B, step):
            m = min(step, B - s)
            _lib.call(entry, _ptr(feat1[s:s + m]), _ptr(feat2[s:s + m]), _ptr(out[s:s + m]),
                      m, Cc, n, int(kernel_size), int(max_displacement), _stream())
    return out


def spatial_attention(emb_x, emb_t, x, tmpl, alpha=0.5, window_size=11, out=None):
    """A10 after the embedding: emb_* [B,N,E] float32, x/tmpl [B,N,...] float32 (or float16 storage,
    BASELINE config 5) -> (out like x, band [B,N,w], prob [B,N,w])."""
    emb_x = _dev(emb_x, torch.float32, "emb_x")
    emb_t = _dev(emb_t, torch.float32, "emb_t")
    half = isinstance(x, torch.Tensor) and x.dtype == torch.float16
    x = _dev(x, torch.float16 if half else torch.float32, "x")
    tmpl = _dev(tmpl, torch.float16 if half else torch.float32, "tmpl")
    entry = "pof_spatial_attention_f16" if half else "pof_spatial_attention"
    if emb_x.shape != emb_t.shape or emb_x.dim() != 3 or x.shape != tmpl.shape:
        raise ValueError("shape mismatch")
    B, N, E = emb_x.shape
    if x.shape[0] != B or x.shape[1] != N:
        raise ValueError("x must be [B,N,...]")
    F = x.numel() // (B * N)
    W = 2 * int(window_size / 2) + 1
    dev = x.device
    band = torch.empty((B, N, W), dtype=torch.float32, device=dev)
    prob = torch.empty((B, N, W), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(dev):
        step = 65535
        for s in range(0, B, step):
            m = min(step, B - s)
            _lib.call(entry, _ptr(emb_x[s:s + m]), _ptr(emb_t[s:s + m]), _ptr(x[s:s + m]),
                      _ptr(tmpl[s:s + m]), m, N, E, F, int(window_size), float(alpha), _ptr(band[s:s + m]),
                      _ptr(prob[s:s + m]), _ptr(out[s:s + m]), _stream())
    return out, band, prob


def band_correlation_backward(feat1, feat2, g_out, kernel_size=3, max_displacement=5):
    """Gradients of band_correlation wrt feat1 / feat2."""
    feat1 = _dev(feat1, torch.float32, "feat1")
    feat2 = _dev(feat2, torch.float32, "feat2")
    g_out = _dev(g_out, torch.float32, "g_out")
    B, Cc, n = feat1.shape
    if tuple(g_out.shape) != (B, 2 * max_displacement + 1, n):
        raise ValueError("g_out must be [B, 2*max_displacement+1, n]")
    d1, d2 = torch.empty_like(feat1), torch.empty_like(feat2)
    with torch.cuda.device(feat1.device):
        for s in range(0, B, 65535):
            m = min(65535, B - s)
            _lib.call("pof_band_correlation_backward", _ptr(feat1[s:s + m]), _ptr(feat2[s:s + m]),
                      _ptr(g_out[s:s + m]), _ptr(d1[s:s + m]), _ptr(d2[s:s + m]), m, Cc, n, int(kernel_size),
                      int(max_displacement), _stream())
    return d1, d2


def spatial_attention_backward(emb_x, emb_t, tmpl, prob, g_out, g_band, alpha, window_size, fused=True):
    """Gradients of spatial_attention: -> (d_emb_x, d_emb_t, d_x, d_tmpl).  ``fused``: one walk over g and tmpl
    (pof_spatial_attention_backward_fused); False: the two-pass form (band product on the MFMA, transposed merge)."""
    emb_x = _dev(emb_x, torch.float32, "emb_x")
    emb_t = _dev(emb_t, torch.float32, "emb_t")
    tmpl = _dev(tmpl, torch.float32, "tmpl")
    prob = _dev(prob, torch.float32, "prob")
    g_out = _dev(g_out, torch.float32, "g_out")
    if g_band is not None:
        g_band = _dev(g_band, torch.float32, "g_band")
    B, N, E = emb_x.shape
    F = tmpl.numel() // (B * N)
    dev = tmpl.device
    dsim = torch.empty_like(prob)
    dex, det = torch.empty_like(emb_x), torch.empty_like(emb_t)
    dx, dt = torch.empty_like(tmpl), torch.empty_like(tmpl)
    with torch.cuda.device(dev):
        for s in range(0, B, 65535):
            m = min(65535, B - s)
            args = (_ptr(emb_x[s:s + m]), _ptr(emb_t[s:s + m]), _ptr(tmpl[s:s + m]), _ptr(prob[s:s + m]),
                    _ptr(g_out[s:s + m]), _ptr(g_band[s:s + m]) if g_band is not None else None, m, N, E, F,
                    int(window_size), float(alpha), _ptr(dsim[s:s + m]), _ptr(dex[s:s + m]), _ptr(det[s:s + m]),
                    _ptr(dx[s:s + m]), _ptr(dt[s:s + m]))
            if fused:
                nbytes = int(_lib.load().pof_spatial_attention_backward_workspace_bytes(m, N, F, int(window_size)))
                ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
                _lib.call("pof_spatial_attention_backward_fused", *args, _ptr(ws), nbytes, _stream())
            else:
                _lib.call("pof_spatial_attention_backward", *args, _stream())
    return dex, det, dx, dt


def segment_features(ranges, tab, jump_dist=0.5, max_seg=None):
    """A13: ranges [B,N] float32 -> (seg_id [B,N] int32, num_seg [B] int32, feat [B,max_seg,16] f64)."""
    ranges = _dev(ranges, torch.float32, "ranges")
    B, N = ranges.shape
    max_seg = N if max_seg is None else int(max_seg)
    dev = ranges.device
    seg_id = torch.empty((B, N), dtype=torch.int32, device=dev)
    num = torch.empty(B, dtype=torch.int32, device=dev)
    feat = torch.full((B, max_seg, 16), float("nan"), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pof_segment_features", _ptr(ranges), _ptr(tab), B, N, float(jump_dist), max_seg,
                  _ptr(seg_id), _ptr(num), _ptr(feat), _stream())
    return seg_id, num, feat


def segment_features_reference(ranges, tab, next_ranges=None, odom_dt=None, wp_offsets=None, wp_xy=None,
                               radius_wp=0.5, jump_dist=0.5, max_seg=None, want_plain=False):
    """A13 in the reference's own form (compute_feature, adaboost_person_det.py:102-210): ranges [B,N] float32
    -> (seg_id [B,N], num_seg [B], num_kept [B], ref_feat [B,max_seg,15] float64[, feat [B,max_seg,16]]).
    next_ranges [B,N] float32 and odom_dt [B] float64 feed the mean-speed column, the annotation CSR
    (wp_offsets [B+1] int32, wp_xy [W,2] float64) the labels."""
    ranges = _dev(ranges, torch.float32, "ranges")
    B, N = ranges.shape
    if next_ranges is not None:
        next_ranges = _dev(next_ranges, torch.float32, "next_ranges")
        if next_ranges.shape != ranges.shape:
            raise ValueError("next_ranges must have the shape of ranges")
    if odom_dt is not None:
        odom_dt = _dev(odom_dt, torch.float64, "odom_dt")
        if odom_dt.numel() != B:
            raise ValueError("odom_dt must be [B]")
    if wp_offsets is not None:
        wp_offsets = _dev(wp_offsets, torch.int32, "wp_offsets")
        wp_xy = _dev(wp_xy, torch.float64, "wp_xy")
        if wp_offsets.numel() != B + 1 or wp_xy.dim() != 2 or wp_xy.shape[1] != 2:
            raise ValueError("annotations must be CSR: wp_offsets [B+1], wp_xy [W,2]")
    max_seg = N if max_seg is None else int(max_seg)
    dev = ranges.device
    seg_id = torch.empty((B, N), dtype=torch.int32, device=dev)
    num = torch.empty(B, dtype=torch.int32, device=dev)
    kept = torch.empty(B, dtype=torch.int32, device=dev)
    ref = torch.full((B, max_seg, 15), float("nan"), dtype=torch.float64, device=dev)
    feat = torch.full((B, max_seg, 16), float("nan"), dtype=torch.float64, device=dev) if want_plain else None
    with torch.cuda.device(dev):
        _lib.call("pof_segment_features_ex", _ptr(ranges), _ptr(next_ranges), _ptr(tab), B, N, float(jump_dist),
                  _ptr(odom_dt), _ptr(wp_offsets), _ptr(wp_xy), float(radius_wp), max_seg, _ptr(seg_id), _ptr(num),
                  _ptr(kept), _ptr(feat), _ptr(ref), _stream())
    return (seg_id, num, kept, ref, feat) if want_plain else (seg_id, num, kept, ref)


_PERM_3D = [0, 1, 3, 4, 6, 2, 5]


def rotate_iou(boxes, query_boxes, criterion=-1, is_3d=False, n_valid=None, k_valid=None):
    """A16: boxes [N,s] / [G,N,s], query [K,s] / [G,K,s] float32 device tensors in the
    REFERENCE column order (3-D: x,y,z,l,w,h,rot) -> iou [N,K] / [G,N,K] float32."""
    if boxes.dim() == 2:
        return rotate_iou(boxes[None], query_boxes[None], criterion, is_3d)[0]
    s = 7 if is_3d else 5
    if boxes.shape[-1] != s or query_boxes.shape[-1] != s or boxes.shape[0] != query_boxes.shape[0]:
        raise ValueError("boxes must be [G,N,%d] and query [G,K,%d]" % (s, s))
    b = boxes.to(torch.float32)
    q = query_boxes.to(torch.float32)
    if is_3d:
        b, q = b[..., _PERM_3D], q[..., _PERM_3D]
    b = _dev(b.contiguous(), torch.float32, "boxes")
    q = _dev(q.contiguous(), torch.float32, "query_boxes")
    G, N, K = b.shape[0], b.shape[1], q.shape[1]
    out = torch.zeros((G, N, K), dtype=torch.float32, device=b.device)
    if n_valid is not None:
        n_valid = _dev(n_valid, torch.int32, "n_valid")
    if k_valid is not None:
        k_valid = _dev(k_valid, torch.int32, "k_valid")
    with torch.cuda.device(b.device):
        _lib.call("pof_rotate_iou", _ptr(b), _ptr(q), _ptr(out), G, N, K, _ptr(n_valid), _ptr(k_valid),
                  int(criterion), int(bool(is_3d)), _stream())
    return out


def gather_windows(scans_all, seq_first, scan_idx, num_scans, distance=5, stride=1, out=None):
    """N1: scans_all [S,N] f32, seq_first / scan_idx [B] int32 -> (windows [B,num_scans+1,N],
    row_cur [B], row_prev [B])."""
    scans_all = _dev(scans_all, torch.float32, "scans_all")
    seq_first = _dev(seq_first, torch.int32, "seq_first")
    scan_idx = _dev(scan_idx, torch.int32, "scan_idx")
    B, N = scan_idx.shape[0], scans_all.shape[1]
    dev = scans_all.device
    if out is None:
        out = torch.empty((B, num_scans + 1, N), dtype=torch.float32, device=dev)
    row_cur = torch.empty(B, dtype=torch.int32, device=dev)
    row_prev = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        for s in range(0, B, 65535):
            m = min(65535, B - s)
            _lib.call("pof_gather_windows", _ptr(scans_all), _ptr(seq_first[s:s + m]), _ptr(scan_idx[s:s + m]), m,
                      int(num_scans), int(distance), int(stride), N, _ptr(out[s:s + m]), _ptr(row_cur[s:s + m]),
                      _ptr(row_prev[s:s + m]), _stream())
    return out, row_cur, row_prev


def associate_odometry(scans_t, odoms_t, odoms, odom_lo, odom_hi, row_cur, row_prev):
    """N1: time association -> (odom0 [B,3] f64, odom1 [B,3] f64, idx0 [B], idx1 [B])."""
    scans_t = _dev(scans_t, torch.float32, "scans_t")
    odoms_t = _dev(odoms_t, torch.float32, "odoms_t")
    odoms = _dev(odoms, torch.float32, "odoms")
    B = row_cur.shape[0]
    dev = scans_t.device
    o0 = torch.empty((B, 3), dtype=torch.float64, device=dev)
    o1 = torch.empty((B, 3), dtype=torch.float64, device=dev)
    i0 = torch.empty(B, dtype=torch.int32, device=dev)
    i1 = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pof_associate_odometry", _ptr(scans_t), _ptr(odoms_t), _ptr(odoms),
                  _ptr(_dev(odom_lo, torch.int32, "odom_lo")), _ptr(_dev(odom_hi, torch.int32, "odom_hi")),
                  _ptr(_dev(row_cur, torch.int32, "row_cur")), _ptr(_dev(row_prev, torch.int32, "row_prev")), B,
                  _ptr(o0), _ptr(o1), _ptr(i0), _ptr(i1), _stream())
    return o0, o1, i0, i1
