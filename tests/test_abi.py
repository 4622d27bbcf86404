"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a
GPU and exports every symbol include/pof_abi.h declares; host-side argument
validation rejects bad input before anything is launched."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from planar_optical_flow_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "pof_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pof_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 15
    raw = ctypes.CDLL(os.path.join(REPO, "planar_optical_flow_amd", "lib", "libpof_hip.so"))
    for n in names:
        assert hasattr(raw, n), "libpof_hip.so does not export %s" % n


def test_binding_matches_header(lib):
    from planar_optical_flow_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert lib.pof_abi_version() == 1
    assert lib.pof_error_string(-2).decode().startswith("inconsistent")


def test_bad_arguments_are_rejected_without_launch(lib):
    """NULL pointers / bad sizes return POF_E_BADARG before any HIP call."""
    from planar_optical_flow_amd import _lib
    with pytest.raises(AssertionError):
        _lib.call("pof_laser_phi", 0.01, 450, None, None)
    with pytest.raises(AssertionError):
        _lib.call("pof_cutout", None, 1, 1, 450, None, 1, 1, 1, 1.0, 1.0, 56, 29.99, 0, None, None, None, None)
    with pytest.raises(AssertionError):
        _lib.call("pof_rotate_iou", None, None, None, 1, 1, 1, None, None, -1, 0, None)


def test_every_entry_point_rejects_null_arguments(lib):
    """Every int-returning device entry point called with all-NULL pointers and zero sizes returns a
    negative status (no launch, no fault); the wrapper maps POF_E_BADARG to AssertionError."""
    import ctypes as C
    from planar_optical_flow_amd import _lib
    skip = {"pof_abi_version", "pof_error_string", "pof_scan_preprocess_workspace_bytes", "pof_nms_workspace_bytes",
            "pof_take_stale_error"}
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        if name in skip or restype is not C.c_int:
            continue
        args = []
        for t in argtypes:
            if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
                args.append(None)
            elif t is C.c_double:
                args.append(0.0)
            else:
                args.append(0)
        rc = getattr(lib, name)(*args)
        assert rc < 0, "%s accepted NULL arguments (rc=%d)" % (name, rc)


def test_new_ops_refuse_cpu_tensors():
    from planar_optical_flow_amd import ops
    with pytest.raises(TypeError):
        ops.polar_grid(torch.zeros(1, 2, 450))
    with pytest.raises(TypeError):
        ops.conv3_bn_lrelu(torch.zeros(2, 1, 56), torch.zeros(3, 1, 64), torch.ones(64), torch.zeros(64))
    with pytest.raises(TypeError):
        ops.segment_inputs(torch.zeros(10, 2, dtype=torch.float64), torch.zeros(1, 2, dtype=torch.float64),
                           torch.zeros(1, dtype=torch.float64))
    with pytest.raises(TypeError):
        ops.band_correlation(torch.zeros(1, 4, 57), torch.zeros(1, 4, 57))


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing on the host."""
    from planar_optical_flow_amd import ops
    with pytest.raises(TypeError):
        ops.scan_preprocess(torch.zeros(2, 450), torch.zeros(1350, dtype=torch.float64), want=("xy",))
    with pytest.raises(TypeError):
        ops.cutout(torch.zeros(1, 2, 450), torch.zeros(1350, dtype=torch.float64))


def test_missing_library_fails_loudly(tmp_path):
    from planar_optical_flow_amd import _lib
    with pytest.raises(ImportError):
        _lib.load(str(tmp_path / "nope.so"))


def test_synth_is_deterministic():
    from planar_optical_flow_amd import synth
    a = synth.make_batch(5, 4, T=3)
    b = synth.make_batch(5, 4, T=3)
    assert np.array_equal(a.scans, b.scans) and np.array_equal(a.odom1, b.odom1)
    o, r, c = a.det_csr()
    assert o[0] == 0 and o[-1] == len(r) == len(c)


def test_wgrad_shape_predicate_mirrors_the_library(lib):
    """torch_ops._wgrad_supported restates make_wgrad()'s shape test in Python (traceable by torch.compile); the library's
    own answer -- a non-zero workspace size -- is the referee."""
    from planar_optical_flow_amd import torch_ops
    import itertools
    lengths = list(range(1, 90)) + [96, 100, 112, 127, 128, 200, 255, 256, 257, 300, 450]
    for L, (ci, co), S, k in itertools.product(lengths, ((1, 64), (64, 64), (64, 128), (128, 1024), (3, 5), (700, 65)),
                                               (1, 7, 4096), (1, 3)):
        want = lib.pof_conv1d_wgrad_workspace_bytes(S, ci, co, L, k) > 0
        assert torch_ops._wgrad_supported(S, ci, co, L, k) == want, (S, ci, co, L, k)
    assert not torch_ops._wgrad_supported(0, 4, 4, 8, 3) and not torch_ops._wgrad_supported(4, 4, 4, 8, 2)
