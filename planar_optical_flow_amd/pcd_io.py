"""PCL ``.pcd`` point-cloud files for the JRDB feeder (reference: the vendored pypcd in
src/data_handle/_pypcd.py, used only through ``point_cloud_from_path(...).pc_data[x|y|z]`` by
src/data_handle/jrdb_handle.py:293-305).

Host I/O only.  The three DATA encodings are read without per-point Python work: ``binary`` rows
are one ``np.frombuffer`` over the file, ``ascii`` is ``np.loadtxt`` on the structured row type (what
the reference does, so the text -> float conversion is the same), and ``binary_compressed`` (LZF,
stored column by column) is decoded by ``pof_lzf_decompress`` in the native library.
``write_pcd`` exists for tests and for exporting segments.
"""
import ctypes
import struct

import numpy as np

_NP_TYPE = {("F", 4): np.float32, ("F", 8): np.float64,
            ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32, ("U", 8): np.uint64,
            ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32, ("I", 8): np.int64}
_PCD_TYPE = {np.dtype(v).str[1:]: k for k, v in _NP_TYPE.items()}


class PCDFormatError(ValueError):
    pass


def _parse_header(f):
    """Reads up to and including the DATA line -> dict(fields, size, type, count, width, height, points, data)."""
    meta = {}
    while True:
        raw = f.readline()
        if not raw:
            raise PCDFormatError("PCD header ends before a DATA line")
        line = raw.decode("ascii", errors="replace").strip()
        if not line or line.startswith("#"):
            continue
        key, _, value = line.partition(" ")
        key, words = key.lower(), value.split()
        if key in ("fields", "type"):
            meta[key] = words
        elif key in ("size", "count"):
            meta[key] = [int(w) for w in words]
        elif key in ("width", "height", "points"):
            meta[key] = int(words[0])
        elif key == "data":
            meta["data"] = value.strip().lower()
            break
        else:
            meta[key] = value          # VERSION, VIEWPOINT: carried, unused
    for need in ("fields", "size", "type"):
        if need not in meta:
            raise PCDFormatError("PCD header lacks %s" % need.upper())
    meta.setdefault("count", [1] * len(meta["fields"]))
    if not (len(meta["fields"]) == len(meta["size"]) == len(meta["type"]) == len(meta["count"])):
        raise PCDFormatError("PCD header: FIELDS / SIZE / TYPE / COUNT lengths differ")
    if "points" not in meta:
        meta["points"] = meta.get("width", 0) * meta.get("height", 1)
    meta.setdefault("width", meta["points"])
    meta.setdefault("height", 1)
    return meta


def _row_dtype(meta):
    names, types = [], []
    for name, size, kind, count in zip(meta["fields"], meta["size"], meta["type"], meta["count"]):
        try:
            t = _NP_TYPE[(kind, size)]
        except KeyError:
            raise PCDFormatError("PCD field %s: unsupported TYPE %s SIZE %d" % (name, kind, size))
        if count == 1:
            names.append(name)
            types.append(t)
        else:                           # flattened like the reference: name_0000, name_0001, ...
            names += ["%s_%04d" % (name, i) for i in range(count)]
            types += [t] * count
    return np.dtype(list(zip(names, types)))


def lzf_decompress(data, out_size):
    """LZF block -> bytes of exactly ``out_size`` (native decoder; raises on malformed input)."""
    from . import _lib
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(out_size, dtype=np.uint8)
    got = _lib.load().pof_lzf_decompress(src.ctypes.data_as(ctypes.c_void_p), src.size,
                                         out.ctypes.data_as(ctypes.c_void_p), out.size)
    if got != out_size:
        raise PCDFormatError("LZF stream is malformed or does not decode to %d bytes" % out_size)
    return out


def lzf_compress(data):
    """Greedy LZF encoder (hash of 3-byte prefixes, back references up to 8 KiB).  Pure Python: meant for
    test files and small exports, not for bulk data."""
    data = bytes(data)
    n, out, lit, i, table = len(data), bytearray(), bytearray(), 0, {}

    def flush():
        for s in range(0, len(lit), 32):
            run = lit[s:s + 32]
            out.append(len(run) - 1)
            out.extend(run)
        lit.clear()

    while i < n:
        ref = -1
        if i + 2 < n:
            key = data[i:i + 3]
            ref = table.get(key, -1)
            table[key] = i
        if ref >= 0 and 0 < i - ref <= 8192:
            length = 3
            while i + length < n and length < 264 and data[ref + length] == data[i + length]:
                length += 1
            flush()
            off, ln = i - ref - 1, length - 2
            if ln < 7:
                out.append((ln << 5) | (off >> 8))
            else:
                out.append((7 << 5) | (off >> 8))
                out.append(ln - 7)
            out.append(off & 0xFF)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def read_pcd(path):
    """-> (structured array [points], header dict)."""
    with open(path, "rb") as f:
        meta = _parse_header(f)
        dtype, n = _row_dtype(meta), meta["points"]
        kind = meta["data"]
        if kind == "binary":
            buf = f.read(n * dtype.itemsize)
            if len(buf) < n * dtype.itemsize:
                raise PCDFormatError("%s: binary payload is shorter than POINTS rows" % path)
            rows = np.frombuffer(buf, dtype=dtype, count=n).copy()
        elif kind == "ascii":
            rows = np.loadtxt(f, dtype=dtype, delimiter=" ", ndmin=1) if n else np.zeros(0, dtype)
        elif kind == "binary_compressed":
            head = f.read(8)
            if len(head) < 8:
                raise PCDFormatError("%s: missing compressed-size words" % path)
            csize, usize = struct.unpack("<II", head)
            if usize != n * dtype.itemsize:
                raise PCDFormatError("%s: uncompressed size %d != POINTS x row size %d" % (path, usize, n * dtype.itemsize))
            raw = lzf_decompress(f.read(csize), usize)
            rows, at = np.zeros(n, dtype=dtype), 0
            for name in dtype.names:                     # stored column by column
                t = dtype[name]
                rows[name] = raw[at:at + n * t.itemsize].view(t)
                at += n * t.itemsize
        else:
            raise PCDFormatError("%s: DATA %r is not ascii / binary / binary_compressed" % (path, kind))
    return rows, meta


def read_pcd_xyz(path):
    """-> float32 [3, N] (x, y, z rows), the array jrdb_handle.py:293-305 builds."""
    rows, _ = read_pcd(path)
    for k in "xyz":
        if k not in rows.dtype.names:
            raise PCDFormatError("%s: no %s field" % (path, k))
    return np.array([rows["x"], rows["y"], rows["z"]], dtype=np.float32)


def write_pcd(path, columns, data="binary"):
    """columns: dict name -> 1-D array (equal lengths, insertion order = field order)."""
    names = list(columns)
    arrs = [np.ascontiguousarray(columns[k]) for k in names]
    n = len(arrs[0]) if arrs else 0
    if any(a.ndim != 1 or len(a) != n for a in arrs):
        raise ValueError("write_pcd: columns must be 1-D and of equal length")
    kinds = [_PCD_TYPE[a.dtype.str[1:]] for a in arrs]
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
            "FIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
            % (" ".join(names), " ".join(str(k[1]) for k in kinds), " ".join(k[0] for k in kinds),
               " ".join("1" for _ in names), n, n, data))
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if data == "ascii":
            for i in range(n):
                f.write((" ".join(repr(a[i].item()) if a.dtype.kind == "f" else str(a[i].item()) for a in arrs) + "\n").encode())
        elif data == "binary":
            rows = np.zeros(n, dtype=np.dtype(list(zip(names, [a.dtype for a in arrs]))))
            for k, a in zip(names, arrs):
                rows[k] = a
            f.write(rows.tobytes())
        elif data == "binary_compressed":
            raw = b"".join(a.tobytes() for a in arrs)
            comp = lzf_compress(raw)
            f.write(struct.pack("<II", len(comp), len(raw)))
            f.write(comp)
        else:
            raise ValueError("write_pcd: data must be ascii, binary or binary_compressed")
