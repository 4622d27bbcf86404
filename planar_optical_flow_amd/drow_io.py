"""DROW on-disk formats -> parsed sequences -> device-resident scan store.

Reference loaders restated here (same return tuples, same dtypes):

  load_scan_file   src/utils/dataset_dr_spaam.py:473-478   `<seq>.csv`    seq,t,r0..r{N-1}
  load_odom        src/utils/dataset_dr_spaam.py:504-509   `<seq>.odom2`  seq,t,x,y,phi
  load_odom_file   src/utils/dataset_dr_spaam.py:497-502   `<seq>.difodom` dt,dx,dy,dphi
  load_det_file    src/utils/dataset_dr_spaam.py:480-495   `<seq>.wc/.wa/.wp`  seq,<json [[r,phi],...]>
  load_flow_file   bin/data_prepare.py:70-72               `<seq>.flow`
  prepare_sequence bin/data_prepare.py:82-114              writes `.difodom` and `.flow`

The numeric files go through the library's CSV reader (`pof_csv_read_f64`, strtod =
Python float(): bit-identical to np.genfromtxt, ~100x faster); there is no NumPy fallback.
`pack_split` / `load_pack` keep a parsed split as one `.npz` so that start-up after the
first run is a single read.
"""
import ctypes as C
import json
import os
from glob import glob

import numpy as np

from . import _lib


def read_csv(path, threads=0):
    """float64 [rows, cols] of a numeric comma-separated file (np.genfromtxt(path, delimiter=","))."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    rows, cols = C.c_longlong(0), C.c_int(0)
    bpath = os.fsencode(path)
    _lib.call("pof_csv_shape", bpath, C.byref(rows), C.byref(cols))
    out = np.empty((rows.value, max(cols.value, 0)), dtype=np.float64)
    if rows.value:
        _lib.call("pof_csv_read_f64", bpath, rows.value, cols.value, out.ctypes.data_as(C.c_void_p), int(threads))
    return out


def load_scan_file(seq_name):
    data = read_csv(seq_name + ".csv")
    seqs = data[:, 0].astype(np.uint32)
    times = data[:, 1].astype(np.float32)
    scans = data[:, 2:].astype(np.float32)
    return seqs, times, scans


def load_odom(seq_name):
    data = read_csv(seq_name + ".odom2")
    return data[:, 0].astype(np.uint32), data[:, 1].astype(np.float32), data[:, 2:5].astype(np.float32)


def load_odom_file(seq_name):
    odoms = read_csv(seq_name + ".difodom")
    return odoms[:, 0], odoms[:, 1:]


def load_flow_file(seq_name, shape=(-1, 450, 2)):
    return read_csv(seq_name + ".flow").reshape(shape)


def load_det_file(seq_name):
    def do_load(f_name):
        seqs, dets = [], []
        with open(f_name) as f:
            for line in f:
                seq, tail = line.split(",", 1)
                seqs.append(int(seq))
                dets.append(json.loads(tail))
        return seqs, dets

    s1, wcs = do_load(seq_name + ".wc")
    s2, was = do_load(seq_name + ".wa")
    s3, wps = do_load(seq_name + ".wp")
    assert all(a == b == c for a, b, c in zip(s1, s2, s3))
    return np.array(s1), wcs, was, wps


def sequence_names(data_path, split, max_sequences=5):
    """Stems of `<data_path>/<split>/*.csv`.  The reference keeps the first five names in
    glob order (dataset_dr_spaam.py:269-271); here the names are sorted first so that the
    selection does not depend on the directory order.  max_sequences=None keeps all."""
    names = sorted(f[:-4] for f in glob(os.path.join(data_path, split, "*.csv")))
    return names if max_sequences is None else names[:max_sequences]


def load_sequences(data_path, split="train", max_sequences=5):
    """The per-sequence dicts DROWDeviceDataset takes, parsed from a DROW split directory."""
    seqs = []
    for name in sequence_names(data_path, split, max_sequences):
        _, odoms_t, odoms = load_odom(name)
        scans_ns, scans_t, scans = load_scan_file(name)
        dets_ns, wc, wa, wp = load_det_file(name)
        seqs.append(dict(name=name, scans_ns=scans_ns, scans_t=scans_t, scans=scans, odoms_t=odoms_t,
                         odoms=odoms, dets_ns=dets_ns, dets_wc=wc, dets_wa=wa, dets_wp=wp))
    if not seqs:
        raise FileNotFoundError("{}: No valid data".format(split))
    return seqs


# ---- one-file binary pack -------------------------------------------------------------
_DET_KEYS = ("dets_wc", "dets_wa", "dets_wp")


def pack_sequences(sequences, path):
    """All sequences in one .npz: numeric arrays concatenated with row offsets, detection
    lists as CSR (frame offsets + [r, phi] rows) per class."""
    out = {"names": np.array([s.get("name", str(i)) for i, s in enumerate(sequences)])}
    for key, dt in (("scans_ns", np.uint32), ("scans_t", np.float32), ("scans", np.float32),
                    ("odoms_t", np.float32), ("odoms", np.float32), ("dets_ns", np.int64)):
        arrs = [np.asarray(s[key], dt) for s in sequences]
        out[key] = np.concatenate(arrs) if arrs else np.zeros((0,), dt)
        out[key + "_rows"] = np.cumsum([0] + [len(a) for a in arrs]).astype(np.int64)
    for key in _DET_KEYS:
        frames = [f for s in sequences for f in s[key]]
        out[key + "_off"] = np.cumsum([0] + [len(f) for f in frames]).astype(np.int64)
        out[key] = (np.array([d for f in frames for d in f], np.float64).reshape(-1, 2))
    np.savez(path, **out)


def load_pack(path):
    z = np.load(path, allow_pickle=False)
    n_seq = len(z["names"])
    seqs = []
    frame0 = 0
    for i in range(n_seq):
        s = {"name": str(z["names"][i])}
        for key in ("scans_ns", "scans_t", "scans", "odoms_t", "odoms", "dets_ns"):
            r = z[key + "_rows"]
            s[key] = z[key][r[i]:r[i + 1]]
        nf = len(s["dets_ns"])
        for key in _DET_KEYS:
            off, rows = z[key + "_off"], z[key]
            s[key] = [rows[off[frame0 + f]:off[frame0 + f + 1]].tolist() for f in range(nf)]
        frame0 += nf
        seqs.append(s)
    return seqs


def pack_split(data_path, split, out_path=None, max_sequences=5):
    out_path = out_path or os.path.join(data_path, split + ".pofpack.npz")
    pack_sequences(load_sequences(data_path, split, max_sequences), out_path)
    return out_path


# ---- bin/data_prepare.py ---------------------------------------------------------------
def prepare_sequence(seq_name, angle_inc=np.radians(0.5), device="cuda"):
    """Writes `<seq>.difodom` (row-to-row odometry differences, last row zero) and `<seq>.flow`
    (get_flow_target of every scan under its difference) exactly as bin/data_prepare.py:82-114
    does; the per-scan flow runs as one batched launch (ops.scan_preprocess, FLOW_PREPARED:
    odom0 = (dx, dy, dphi), odom1[0] = dt)."""
    import torch
    from . import ops
    _, odom_t, odom = load_odom(seq_name)
    diff_t = np.concatenate((odom_t[1:] - odom_t[:-1], [0]))
    diff = np.concatenate((odom[1:] - odom[:-1], [[0] * 3]))
    np.savetxt(seq_name + ".difodom", np.hstack([diff_t.reshape(-1, 1), diff]), fmt="%8.6f", delimiter=",")
    _, _, scans = load_scan_file(seq_name)
    odoms_t, odoms = load_odom_file(seq_name)          # the rounded text is what the reference re-reads
    n = min(len(scans), len(odoms))
    dev = torch.device(device)
    tab = ops.phi_table(angle_inc, scans.shape[1], device=dev)
    o0 = torch.from_numpy(np.ascontiguousarray(odoms[:n], np.float64)).to(dev)
    o1 = torch.zeros_like(o0)
    o1[:, 0] = torch.from_numpy(np.ascontiguousarray(odoms_t[:n], np.float64)).to(dev)
    res = ops.scan_preprocess(torch.from_numpy(scans[:n]).to(dev).unsqueeze(1), tab, o0, o1, None,
                              flow_kind=ops.FLOW_PREPARED, canonical=False, out_dtype=torch.float64,
                              want=("flow",))
    flow = res["flow"].cpu().numpy()
    np.savetxt(seq_name + ".flow", flow.reshape(-1, scans.shape[1] * 2), fmt="%10.8f", delimiter=",")
    return flow
