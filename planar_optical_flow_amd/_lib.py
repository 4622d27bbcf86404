"""ctypes binding of libpof_hip.so (include/pof_abi.h).

There is no CPU fallback: if the shared library is missing or a call returns a
non-zero code this module raises.  The library is located in-tree
(planar_optical_flow_amd/lib/libpof_hip.so, produced by
``python -m planar_optical_flow_amd.build`` / ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpof_hip.so")

POF_OK, POF_E_BADARG, POF_E_SHAPE, POF_E_LAUNCH, POF_E_WORKSPACE = 0, -1, -2, -3, -4

_p = C.c_void_p
_i = C.c_int
_d = C.c_double
_ll = C.c_longlong
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/pof_abi.h one to one
SIGNATURES = {
    "pof_abi_version": (_i, []),
    "pof_error_string": (C.c_char_p, [_i]),
    "pof_take_stale_error": (_i, []),
    "pof_laser_phi": (_i, [_d, _i, _p, _p]),
    "pof_scan_preprocess_workspace_bytes": (_sz, [_i, _i]),
    "pof_scan_preprocess_phase": (_i, [_p, _ll, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p,
                                       _p, _p, _p, _p, _p, _p, _p, _sz, _i, _p]),
    "pof_scan_preprocess_chained": (_i, [_p, _ll, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p,
                                         _p, _p, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "pof_scan_preprocess": (_i, [_p, _ll, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p,
                                 _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pof_scan_preprocess_multi": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "pof_flow_from_xy": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _i, _p]),
    "pof_xy_to_rphi": (_i, [_p, _p, _p, _p, _ll, _p]),
    "pof_rotate_flow": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "pof_det_to_canonical": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "pof_canonical_to_det": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "pof_cutout": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _d, _d, _i, _d, _i, _p, _p, _p, _p]),
    "pof_cutout_ex": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _d, _d, _i, _d, _i, _i, _p, _p, _p, _p]),
    "pof_cutout_f16": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _d, _d, _i, _d, _i, _i, _p, _p, _p, _p]),
    "pof_nms_workspace_bytes": (_sz, [_i, _i]),
    "pof_nms_predicted_center": (_i, [_p, _p, _p, _p, _d, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "pof_flow_errors": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "pof_band_correlation": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pof_band_correlation_f16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pof_spatial_attention_f16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p, _p, _p]),
    "pof_band_correlation_backward": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pof_spatial_attention_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pof_spatial_attention_backward_fused": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pof_spatial_attention_backward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p, _p, _p, _p, _p]),
    "pof_spatial_attention": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p, _p, _p]),
    "pof_segment_features": (_i, [_p, _p, _i, _i, _d, _i, _p, _p, _p, _p]),
    "pof_segment_features_ex": (_i, [_p, _p, _p, _i, _i, _d, _p, _p, _p, _d, _i, _p, _p, _p, _p, _p, _p]),
    "pof_gather_windows": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "pof_associate_odometry": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "pof_rotate_iou": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _i, _i, _p]),
    "pof_conv3_bn_lrelu": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p]),
    "pof_conv1d_bn_lrelu": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _d, _p, _p]),
    "pof_bn_lrelu_pool_workspace_bytes": (_sz, [_ll, _i, _i, _i]),
    "pof_bn_lrelu_pool_forward": (_i, [_p, _ll, _i, _i, _i, _p, _p, _p, _p, _d, _d, _d, _i, _p, _p, _p, _p, _sz, _p]),
    "pof_bn_lrelu_pool_backward": (_i, [_p, _p, _ll, _i, _i, _i, _p, _p, _p, _p, _d, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "pof_conv3_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pof_conv3_wgrad": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "pof_conv3_first_two": (_i, [_p, _p, _d, _p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p]),
    "pof_regression_loss2": (_i, [_p, _p, _ll, _i, _d, _p, _p, _p]),
    "pof_linear_bias": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "pof_conv1d_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "pof_conv1d_wgrad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "pof_drow_heads": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p]),
    "pof_segment_inputs": (_i, [_p, _i, _i, _p, _p, _i, _d, _i, _i, C.c_uint32, _p, _p, _p, _p]),
    "pof_segment_resample": (_i, [_p, _i, _p, _i, _i, _p, _p, _d, _i, C.c_uint32, _p, _p, _p]),
    "pof_polar_grid": (_i, [_p, _i, _i, _i, _d, _d, _d, _d, _i, _p, _p]),
    "pof_csv_shape": (_i, [C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "pof_csv_read_f64": (_i, [C.c_char_p, _ll, _i, _p, _i]),
    "pof_lzf_decompress": (_ll, [_p, _ll, _p, _ll]),
    "pof_stump_search": (_i, [_p, _p, _ll, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "pof_stump_vote": (_i, [_p, _ll, _i, _p, _p, _p, _i, _p, _p, _p]),
}


class ScanInputs(C.Structure):
    """pof_scan_inputs of include/pof_abi.h."""
    _fields_ = [("odom0", C.c_void_p), ("odom1", C.c_void_p), ("det_offsets", C.c_void_p),
                ("det_rphi", C.c_void_p), ("det_cls", C.c_void_p), ("B", C.c_int32), ("D", C.c_int32),
                ("flow_kind", C.c_int32), ("want_flow", C.c_int32), ("assoc_radius", C.c_double * 3),
                ("labels", C.c_int32 * 3), ("pad_", C.c_int32), ("dyn_radius", C.c_double * 3),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class ScanBatch(C.Structure):
    """pof_scan_batch of include/pof_abi.h."""
    _fields_ = [("ranges", C.c_void_p), ("sample_stride", C.c_longlong), ("B", C.c_int32), ("D", C.c_int32),
                ("det_offsets", C.c_void_p), ("xy", C.c_void_p), ("flow", C.c_void_p), ("closest", C.c_void_p),
                ("target_cls", C.c_void_p), ("target_reg", C.c_void_p), ("dyn_mask", C.c_void_p),
                ("valid_mask", C.c_void_p), ("exclude_mask", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


SCAN_MAX_SLOTS = 8


class PofError(RuntimeError):
    def __init__(self, fn, code, msg):
        super().__init__("%s failed with code %d: %s" % (fn, code, msg))
        self.code = code


_lib = None


def load(path=None):
    """Load the library and declare every prototype.  Raises if it is absent:
    the product path has no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("POF_LIB_PATH") or LIB_PATH      # POF_LIB_PATH: experimental builds (tools/)
    # torch must be imported BEFORE the library is dlopen'ed: PyTorch-ROCm ships its own
    # libamdhip64, and the process has to end up with ONE HIP runtime -- the one that owns
    # torch's devices and streams.  Loaded the other way round, libpof_hip.so binds to the
    # system runtime and every launch on a torch stream fails (hipErrorInvalidResourceHandle).
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise ImportError(
            "libpof_hip.so not found at %s -- build it with "
            "`python -m planar_optical_flow_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    lib = C.CDLL(path)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError("libpof_hip.so lacks exports declared in include/pof_abi.h: %s" % missing)
    if lib.pof_abi_version() != 1:
        raise ImportError("libpof_hip.so ABI version mismatch")
    _lib = lib
    return lib


class StaleHipError(RuntimeWarning):
    """An earlier HIP call of this thread (not one of this library's) had failed when an entry point was
    entered; the library took the sticky error out of the way and reports it here instead of swallowing it."""


_STRICT = os.environ.get("POF_STRICT_ERRORS", "0") == "1"


def take_stale_error():
    """hipError_t code the library found pending at an entry point since the last query (0 = none)."""
    return int(load().pof_take_stale_error())


def call(name, *args):
    """Invoke an int-returning entry point; non-zero -> exception.
    POF_E_BADARG maps to AssertionError like the reference's input guards.  With POF_STRICT_ERRORS=1 a HIP
    error that was pending when the entry point was entered raises StaleHipError (as a warning) here."""
    lib = load()
    code = getattr(lib, name)(*args)
    if _STRICT:
        stale = lib.pof_take_stale_error()
        if stale:
            import warnings
            warnings.warn("%s: HIP error %d was pending from an earlier call of this thread" % (name, stale),
                          StaleHipError, stacklevel=2)
    if code != POF_OK:
        msg = lib.pof_error_string(code).decode()
        if code == POF_E_BADARG:
            raise AssertionError("%s: %s" % (name, msg))
        raise PofError(name, code, msg)
    return code
