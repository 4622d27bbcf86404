"""JRDB file side: .pcd reader (ascii / binary / LZF), the handle's index and sensor loading against the
reference's own JRDBHandle run on the same synthetic tree (tests/golden/jrdb_files.npz, made by
tools/gen_golden.py), and -- on the GPU -- the segments cut by the HIP radius query."""
import os
import struct

import numpy as np
import pytest

from planar_optical_flow_amd import pcd_io

import jrdb_tree

GOLD = os.path.join(os.path.dirname(__file__), "golden", "jrdb_files.npz")
CFG = {"radius_segment": 0.7, "perturb": 0.1, "is_3d": True}


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def tree(tmp_path_factory, gold):
    root = str(tmp_path_factory.mktemp("JRDB"))
    labelled = jrdb_tree.make_tree(root, [str(s) for s in gold["val_sequences"]], seed=11)
    return root, labelled


# ---------------------------------------------------------------------------------------------
# .pcd
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 7, 2000])
@pytest.mark.parametrize("kind", jrdb_tree.PCD_KINDS)
def test_pcd_round_trip(tmp_path, n, kind):
    rng = np.random.default_rng(n + 3)
    cols = {"x": np.round(rng.normal(size=n), 1).astype(np.float32), "y": np.zeros(n, np.float32),
            "z": rng.normal(size=n).astype(np.float32), "ring": rng.integers(0, 16, n).astype(np.uint16),
            "t": rng.normal(size=n)}
    path = str(tmp_path / "c.pcd")
    pcd_io.write_pcd(path, cols, data=kind)
    rows, meta = pcd_io.read_pcd(path)
    assert meta["points"] == n and meta["data"] == kind
    for k, v in cols.items():
        assert rows[k].dtype == v.dtype and np.array_equal(rows[k], v), (k, kind)
    xyz = pcd_io.read_pcd_xyz(path)
    assert xyz.dtype == np.float32 and xyz.shape == (3, n) and np.array_equal(xyz[2], cols["z"])


def test_lzf_known_streams():
    # literal run, then a 3-byte back reference, then an overlapping (run-length) reference with the long form
    assert bytes(pcd_io.lzf_decompress(b"\x02abc" + b"\x20\x02", 6)) == b"abcabc"
    assert bytes(pcd_io.lzf_decompress(b"\x00z" + b"\xe0\x0a\x00", 20)) == b"z" * 20
    raw = np.repeat(np.random.default_rng(0).integers(0, 200, 3000), 10).astype(np.uint8).tobytes()   # runs of 10
    comp = pcd_io.lzf_compress(raw)
    assert len(comp) < len(raw) // 2 and bytes(pcd_io.lzf_decompress(comp, len(raw))) == raw


@pytest.mark.parametrize("bad", [b"\x05abc", b"\x20", b"\x20\x00", b"\xe0", b"\x00a\x40\x05", b"\x00a\x00b"])
def test_lzf_rejects_malformed(bad):
    """Truncated literal run, truncated reference, reference before the start, output overrun / underrun."""
    with pytest.raises(pcd_io.PCDFormatError):
        pcd_io.lzf_decompress(bad, 1 if bad == b"\x00a\x00b" else 16)


def test_pcd_header_errors(tmp_path):
    p = str(tmp_path / "bad.pcd")
    open(p, "wb").write(b"VERSION .7\nFIELDS x y z\nSIZE 4 4 4\n")
    with pytest.raises(pcd_io.PCDFormatError):
        pcd_io.read_pcd(p)
    open(p, "wb").write(b"FIELDS x y\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n" + b"\0" * 8)
    with pytest.raises(pcd_io.PCDFormatError):
        pcd_io.read_pcd_xyz(p)                    # no z field
    open(p, "wb").write(b"FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n" + b"\0" * 20)
    with pytest.raises(pcd_io.PCDFormatError):
        pcd_io.read_pcd(p)                        # short payload
    open(p, "wb").write(b"FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA binary_compressed\n"
                        + struct.pack("<II", 3, 99) + b"\x01ab")
    with pytest.raises(pcd_io.PCDFormatError):
        pcd_io.read_pcd(p)                        # declared size does not match POINTS
    open(p, "wb").write(b"FIELDS x\nSIZE 3\nTYPE F\nCOUNT 1\nWIDTH 1\nPOINTS 1\nDATA ascii\n1\n")
    with pytest.raises(pcd_io.PCDFormatError):
        pcd_io.read_pcd(p)                        # unsupported field type


def test_pcd_multi_count_fields_and_height(tmp_path):
    p = str(tmp_path / "m.pcd")
    body = np.arange(2 * 3 * 5, dtype=np.float32)       # 6 points x (x, y, z, n[2])
    open(p, "wb").write(b"# organised cloud\nFIELDS x y z n\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 2\n"
                        b"WIDTH 3\nHEIGHT 2\nDATA binary\n" + body.tobytes())
    rows, meta = pcd_io.read_pcd(p)
    assert meta["points"] == 6 and rows.dtype.names == ("x", "y", "z", "n_0000", "n_0001")
    assert np.array_equal(rows["n_0001"], body[4::5])


# ---------------------------------------------------------------------------------------------
# handle: index and sensor loading (host only)
# ---------------------------------------------------------------------------------------------
def test_split_lists_are_the_references(gold):
    from planar_optical_flow_amd.src.data_handle import jrdb_handle as jh
    assert jh._JRDB_VAL_SEQUENCES == [str(s) for s in gold["val_sequences"]]
    assert jh._JRDB_TRAIN_SEQUENCES == [str(s) for s in gold["train_sequences"]]


def test_handle_index_and_points_match_reference(tree, gold):
    from planar_optical_flow_amd.src.data_handle.jrdb_handle import JRDBHandle
    root, labelled = tree
    h = JRDBHandle("val", dict(CFG, data_dir=root))
    assert len(h) == int(gold["len"]) == len(labelled)
    assert [s for s, _, _ in labelled] == [str(s) for s in gold["labelled_seq"]]
    for i in range(len(h)):
        q, k = h._index[i]
        frame = h.sequence_pc_frames[q][k]
        assert frame["pointclouds"]["upper_velodyne"]["url"] == str(gold["f%d_url" % i])
        pts = h.load_points(frame)
        ref = gold["f%d_points" % i]
        assert pts.dtype == ref.dtype and pts.shape == ref.shape
        assert np.array_equal(pts, ref), i                   # ascii and binary clouds, float32 transform
    with pytest.raises(IndexError):
        h._index[len(h)]
    with pytest.raises(AssertionError):
        JRDBHandle("training", dict(CFG, data_dir=root))


def test_handle_laser_points_match_reference(tree, gold):
    from planar_optical_flow_amd.src.data_handle.jrdb_handle import JRDBHandle
    root, _ = tree
    h = JRDBHandle("test", dict(CFG, data_dir=root, is_3d=False))      # "test" reads the validation sequences
    for i in gold["empty_index"]:
        q, k = h._index[int(i)]
        pts = h.load_points(h.sequence_pc_frames[q][k])
        ref = gold["laser%d_points" % int(i)]
        assert pts.shape == ref.shape == (180, 3) and pts.dtype == ref.dtype
        assert np.array_equal(pts, ref)


def test_handle_sequences_override_and_lzf_clouds(tmp_path):
    from planar_optical_flow_amd.src.data_handle.jrdb_handle import JRDBHandle
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    jrdb_tree.make_tree(a, ["s0", "s1"], seed=4, pcd_kinds=("binary",))
    jrdb_tree.make_tree(b, ["s0", "s1"], seed=4, pcd_kinds=("binary_compressed",))
    ha = JRDBHandle("train", dict(CFG, data_dir=a, sequences=["s0", "s1"]))
    hb = JRDBHandle("train", dict(CFG, data_dir=b, sequences=["s0", "s1"]))
    assert len(ha) == len(hb) == 4
    for i in range(len(ha)):
        qa, ka = ha._index[i]
        assert np.array_equal(ha.load_points(ha.sequence_pc_frames[qa][ka]), hb.load_points(hb.sequence_pc_frames[qa][ka]))


# ---------------------------------------------------------------------------------------------
# handle[i] end to end: segments through the HIP radius query
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_handle_frames_match_reference(tree, gold):
    from planar_optical_flow_amd.src.data_handle.jrdb_handle import JRDBHandle
    root, labelled = tree
    h = JRDBHandle("val", dict(CFG, data_dir=root))
    seen = 0
    for i in range(len(h)):
        np.random.seed(1000 + i)                               # the reference draws from the global state
        fr = h[i]
        assert np.array_equal(fr["points"], gold["f%d_points" % i])
        assert np.array_equal(np.asarray(fr["boxes"], np.float64).reshape(-1, 7), gold["f%d_boxes" % i].reshape(-1, 7))
        assert np.array_equal(np.asarray(fr["dets_center"]).reshape(-1, 3), gold["f%d_centers" % i].reshape(-1, 3))
        lens = [len(s) for s in fr["segments"]]
        assert lens == list(gold["f%d_seg_len" % i])
        if lens:
            got = np.concatenate(fr["segments"])
            assert got.dtype == gold["f%d_seg_pts" % i].dtype and np.array_equal(got, gold["f%d_seg_pts" % i])
        seen += len(lens)
        assert "segments" not in h.sequence_pc_frames[h._index[i][0]][h._index[i][1]]     # stored record untouched
    assert seen > 20
    assert sum(1 for _ in h) == len(h)                         # iteration protocol ends on IndexError


@pytest.mark.gpu
def test_handle_feeds_the_device_dataset(tree):
    from planar_optical_flow_amd.src.data_handle.jrdb_handle import JRDBHandle
    from planar_optical_flow_amd.src.data_handle.jrdb_dataset import JRDBBoxRegressionDataset
    root, _ = tree
    h = JRDBHandle("val", dict(CFG, data_dir=root), rng=np.random.default_rng(2))
    cfg = {"input_size": 64, "is_3d": True, "min_segment_size": 5,
           "augmentation_kwargs": {"use_data_augmentation": False, "rot_max": 0.1, "dim_max": 0.1, "dist_max": 0.1,
                                   "random_drop": 0.0}}
    ds = JRDBBoxRegressionDataset("val", cfg, h, rng=np.random.default_rng(3))
    assert len(ds) > 10
    # the reference's two-argument constructor builds its own handle from cfg["data_dir"]
    ds2 = JRDBBoxRegressionDataset("val", dict(cfg, data_dir=root, **CFG), rng=np.random.default_rng(2))
    assert len(ds2) == len(ds)
    from planar_optical_flow_amd.src.data_handle.get_dataloader import get_dataloader
    loader = get_dataloader("val", 16, 0, True, dict(cfg, data_dir=root, **CFG))       # the reference's call, from files
    sizes = [b["input"].shape[0] for b in loader]
    assert sum(sizes) == len(ds) and max(sizes) == 16
    batch = ds.get_batch(list(range(8)))
    assert batch["input"].shape == (8, 64, 4) and batch["input"].is_cuda
    assert bool(np.isfinite(batch["input"].cpu().numpy()).all())
