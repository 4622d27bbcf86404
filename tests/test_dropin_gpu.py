"""The reference's own call sites, spelled the reference's way (``import
src.utils.utils as u``), running on the HIP path: NumPy in -> NumPy out."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from planar_optical_flow_amd import synth

pytestmark = pytest.mark.gpu

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planar_optical_flow_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


@pytest.fixture(scope="module")
def u():
    import src.utils.utils as _u
    return _u


@pytest.fixture(scope="module")
def geo(golden):
    g = golden("scan_geometry")
    return g, synth.make_batch(seed=int(g["seed"]), B=int(g["B"]), T=2, mixed_classes=True)


def test_getitem_sequence_like_the_reference(u, geo):
    """dataset_dr_spaam.py:384-409, one sample at a time."""
    g, sb = geo
    scan_phi = u.get_laser_phi()
    assert np.array_equal(scan_phi, R.laser_phi())
    for b in range(3):
        cur = sb.scans[b, -1]
        d = sb.dets[b]
        wc, wa, wp = [list(map(tuple, d[k])) for k in ("wc", "wa", "wp")]
        target_cls, target_reg = u.get_regression_target(cur, scan_phi, wc, wa, wp)
        assert target_cls.dtype == np.int64 and target_reg.dtype == np.float32
        assert np.array_equal(target_cls, g["target_cls"][b])
        np.testing.assert_allclose(target_reg, g["target_reg"][b], atol=1e-6)
        cur_scan_xy = np.array(u.rphi_to_xy(cur, scan_phi)).T
        np.testing.assert_allclose(cur_scan_xy, g["xy"][b], atol=1e-14)
        flow = u.get_displacement_from_odometry(cur_scan_xy, sb.odom0[b], sb.odom1[b])
        np.testing.assert_allclose(flow, g["disp"][b], atol=1e-12)
        flow_c = u.global_to_canonical_flow(flow, scan_phi)
        np.testing.assert_allclose(flow_c, g["disp_canonical"][b], atol=1e-12)
        np.testing.assert_allclose(u.canonical_to_global_flow(flow_c, scan_phi), g["disp_back"][b], atol=1e-12)
        np.testing.assert_allclose(u.get_flow_target(cur, scan_phi, sb.odom0[b], sb.odom1[b], True),
                                   g["flow_target_canonical"][b], atol=1e-12)
        np.testing.assert_allclose(u.get_velocity_from_odometry(cur_scan_xy, sb.odom0[b], sb.odom1[b]),
                                   g["velocity"][b], atol=1e-12)
        dets = wc + wa + wp
        radii = [0.6] * len(wc) + [0.4] * len(wa) + [0.35] * len(wp)
        assert np.array_equal(u.closest_detection(cur, scan_phi, dets, radii), g["closest"][b])
    assert np.array_equal(u.closest_detection(sb.scans[0, -1], scan_phi, [], []), np.zeros(450, dtype=int))
    with pytest.raises(AssertionError):
        u.closest_detection(sb.scans[0, -1], scan_phi, [(1.0, 0.0)], [])


def test_a5_and_polar_helpers(u, geo):
    g, sb = geo
    phi = u.get_laser_phi()
    r, p = u.canonical_to_global(sb.scans[0, -1], phi, g["a5_dx"], g["a5_dy"])
    np.testing.assert_allclose(r, g["a5_det_r"], rtol=1e-14)
    x, y = u.global_to_canonical(sb.scans[0, -1], phi, r, p)
    np.testing.assert_allclose(x, g["a5_back_x"], atol=1e-13)
    rr, pp = u.xy_to_rphi(g["xy"][0][:, 0], g["xy"][0][:, 1])
    np.testing.assert_allclose(rr, sb.scans[0, -1].astype(np.float64), rtol=1e-15)
    np.testing.assert_allclose(pp, phi, atol=1e-15)


def test_cutout_and_nms_dropin(u, golden):
    from cases import CUTOUT_CASES
    g = golden("cutout")
    inc, n, kw = CUTOUT_CASES["dr_spaam"]
    phi = u.get_laser_phi(np.radians(inc), n)
    got = u.scans_to_cutout(g["dr_spaam_scans"][0], phi, **kw)
    assert got.dtype == np.float32 and got.shape == g["dr_spaam_out"][0].shape
    assert np.array_equal(got, R.cutout(g["dr_spaam_scans"][0], phi, atan_mode="cr", **kw))
    gn = golden("nms")
    xy, cls, inst = u.nms_predicted_center(gn["scan0"], u.get_laser_phi(), gn["cls0"], gn["reg0"], 0.5)
    assert np.array_equal(inst, gn["inst0"]) and inst.dtype == np.int32
    np.testing.assert_allclose(xy, gn["xy0"], atol=1e-12)
    assert np.array_equal(cls, gn["keepcls0"])
    with pytest.raises(AssertionError):
        u.nms_predicted_center(gn["scan0"], u.get_laser_phi(), gn["cls0"][:, 0], gn["reg0"])


def test_batch_preprocessor_matches_per_sample_oracle():
    """DROWDataset2.__getitem__ + collate_batch for a whole batch in three launches."""
    from planar_optical_flow_amd.preprocess import DROWBatchPreprocessor
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=56,
              padding_val=29.99, area_mode=True)
    pre = DROWBatchPreprocessor(cutout_kwargs=kw)
    sb = synth.make_batch(seed=21, B=16, T=6)
    dets = pre.make_detections([d["wc"] for d in sb.dets], [d["wa"] for d in sb.dets], [d["wp"] for d in sb.dets])
    batch = pre(torch.from_numpy(sb.scans).cuda(), torch.from_numpy(sb.odom0).cuda(),
                torch.from_numpy(sb.odom1).cuda(), dets)
    assert batch["input"].shape == (16, 450, 6, 56) and batch["target_flow"].shape == (16, 450, 2)
    phi = R.laser_phi()
    for b in (0, 7, 15):
        cur = sb.scans[b, -1]
        d = sb.dets[b]
        cls, reg = R.regression_target(cur, phi, d["wc"], d["wa"], d["wp"])
        xy = np.array(R.polar_to_xy(cur, phi)).T
        flow = R.flow_to_canonical(R.displacement_from_odometry(xy, sb.odom0[b], sb.odom1[b]), phi)
        mask = R.dynamic_mask(xy, d["wc"], d["wa"], d["wp"]) * R.valid_point_mask(cur)
        assert np.array_equal(batch["target_cls"][b].cpu().numpy(), cls)
        np.testing.assert_allclose(batch["target_reg"][b].cpu().numpy(), reg, atol=1e-6)
        np.testing.assert_allclose(batch["target_flow"][b].cpu().numpy(), flow, atol=5e-6)   # float32 output
        assert np.array_equal(batch["exclude_mask"][b].cpu().numpy().astype(np.float64), mask)
        assert np.array_equal(batch["input"][b].cpu().numpy(), R.cutout(sb.scans[b], phi, atan_mode="cr", **kw))


def test_rotate_iou_and_eval_metrics_dropin(golden):
    from src.utils.rotate_iou import rotate_iou_batched, rotate_iou_gpu_eval
    from src.utils.eval_utils import flow_epe, loss_fn_eval
    b1 = np.array([[0, 0, 0.7, 1, 1, 1, 0]])
    b2 = np.array([[0, 0, 0, 1, 1, 1, 0]])
    np.testing.assert_allclose(rotate_iou_gpu_eval(b1, b2, is_3d=True)[0, 0], 0.3 / 1.7, rtol=1e-6)  # reference __main__
    assert rotate_iou_gpu_eval(np.zeros((0, 5)), np.zeros((3, 5))).shape == (0, 3)
    rng = np.random.default_rng(4)
    boxes = np.concatenate([rng.uniform(-1, 1, (5, 2)), rng.uniform(0.4, 1.2, (5, 2)), rng.uniform(-3, 3, (5, 1))], 1)
    qs = [np.concatenate([rng.uniform(-1, 1, (k, 2)), rng.uniform(0.4, 1.2, (k, 2)), rng.uniform(-3, 3, (k, 1))], 1)
          for k in (3, 0, 5, 1, 2)]
    got = rotate_iou_batched(boxes, qs)
    for i in range(5):
        want = R.rotate_iou(boxes[i:i + 1], qs[i])[0] if len(qs[i]) else np.zeros(0)
        np.testing.assert_allclose(got[i], want, atol=1e-5)
    g = golden("losses")
    epe, aae = loss_fn_eval(g["pred"], g["target"])
    np.testing.assert_allclose(epe.cpu().numpy(), g["epe"], rtol=1e-5)
    np.testing.assert_allclose(aae.cpu().numpy(), g["aae"], rtol=1e-4)
    np.testing.assert_allclose(flow_epe(g["pred"], g["target"], g["mask"]), g["masked"], rtol=1e-5)


def test_spatial_attention_module_with_reference_weights(golden):
    """The nn.Module mirror loaded with the reference module's state dict."""
    from src.depracted.model.dr_spaam import _SpatialAttention
    from src.depracted.model.prototype import fusion
    g = golden("spatial_attn")
    att = _SpatialAttention(n_pts=14, n_channel=32, alpha=0.5, window_size=11)
    sd = {k[3:].replace("conv_0_", "conv.0.").replace("conv_1_", "conv.1."): torch.from_numpy(g[k])
          for k in g.files if k.startswith("sd_")}
    att.load_state_dict(sd)
    att = att.cuda().eval()
    with torch.no_grad():
        out, band = att(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["tmpl"]).cuda())
    np.testing.assert_allclose(band.cpu().numpy(), g["band"], rtol=1e-3, atol=1e-3)   # MIOpen conv vs MKL
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-3, atol=1e-4)
    gc = golden("band_corr")
    np.testing.assert_allclose(fusion(torch.from_numpy(gc["f1"]).cuda(), torch.from_numpy(gc["f2"]).cuda()).cpu().numpy(),
                               gc["out"], rtol=1e-4, atol=1e-3)


def test_box_regressor_dropin(golden):
    """BoxRegressor(ckpt)(points, centre, ori): the segment it feeds the network is
    the reference's (as a set), the forward equals the module's."""
    from planar_optical_flow_amd.box_regressor import BoxRegressor
    from src.model.get_model import get_model
    torch.manual_seed(61)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3})
    br = BoxRegressor({"model_state": model.state_dict()}, gpu=True, seed=5)
    rng = np.random.default_rng(6)
    centre = np.array([2.0, 1.0])
    pts = np.concatenate([centre + rng.normal(0, 0.15, (40, 2)), rng.uniform(-8, 8, (300, 2))])
    seg = br.generate_segment(pts, centre, 0.4)
    want = R.radius_query(pts, centre, 0.4)
    assert np.array_equal(seg, want)
    out = br(pts, centre, 0.3)
    assert out.shape == (5,) and np.allclose(out[:2], centre)
    assert br(pts, np.array([50.0, 50.0]), 0.0) is None            # fewer than 5 points
    outs = br.regress_batch(pts, [centre, np.array([50.0, 50.0])], [0.3, 0.0])
    assert outs[1] is None and outs[0].shape == (5,)
    # max-pool over points makes the prediction independent of the resampling order and
    # of how often a point is repeated: compare with a direct forward on the unique set
    x = torch.from_numpy(np.hstack([want - centre, np.full((len(want), 1), 0.3)])).float()[None].cuda()
    with torch.no_grad():
        direct = br.model(x)[0].cpu().numpy()
    np.testing.assert_allclose(out[2:4], direct[:2], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out[4], direct[2] + 0.3, rtol=1e-4, atol=1e-5)


def test_device_dataset_equals_reference_getitem(golden):
    """N1: the reference's DROWDataset2.__getitem__ + collate_batch (golden, produced by the
    reference itself) vs the device-resident data set: four launches for the whole batch."""
    from dataset_fixture import CUTOUT_KW, load_sequences
    from planar_optical_flow_amd.scan_store import DROWDeviceDataset
    g = golden("dataset_items")
    seqs = load_sequences(g)
    ds = DROWDeviceDataset(seqs, num_scans=5, cutout_kwargs=CUTOUT_KW, drop_static=False)
    n = int(g["n_items"])
    assert len(ds) == n
    batch = ds.get_batch(list(range(n)))
    assert torch.equal(batch["scans"].cpu(), torch.from_numpy(g["out_scans"]))            # window gather
    assert np.array_equal(batch["odom1"].cpu().numpy().astype(np.float32), g["out_odom1"])  # time association
    assert np.array_equal(batch["odom1_t"].cpu().numpy(), g["out_odom1_t"])
    assert np.array_equal(batch["scans_ns"].cpu().numpy(), g["out_scans_ns"].astype(np.int64))
    assert np.array_equal(np.array(batch["dets_ns"]), g["out_dets_ns"].astype(np.int64))
    assert np.array_equal(batch["target_cls"].cpu().numpy(), g["out_target_cls"])
    np.testing.assert_allclose(batch["target_reg"].cpu().numpy(), g["out_target_reg"], atol=1e-6)
    np.testing.assert_allclose(batch["target_flow"].cpu().numpy(), g["out_target_flow"], atol=5e-6)
    assert np.array_equal(batch["exclude_mask"].cpu().numpy().astype(np.float64), g["out_exclude_mask"])
    got = batch["input"][:3].cpu().numpy()
    frac = np.mean(np.abs(got - g["out_input_first3"]) > 1e-4)      # np.arctan last-bit effect only
    assert frac < 2e-3, frac
    phi = R.laser_phi()
    assert np.array_equal(got[0], R.cutout(g["out_scans"][0], phi, atan_mode="cr", **CUTOUT_KW))
    # a shuffled subset gives the same rows
    sub = ds.get_batch([5, 0, n - 1])
    assert torch.equal(sub["scans"], batch["scans"][[5, 0, n - 1]])
    assert torch.equal(sub["target_cls"], batch["target_cls"][[5, 0, n - 1]])


def test_device_dataset_static_filter():
    from planar_optical_flow_amd.scan_store import DROWDeviceDataset
    rng = np.random.default_rng(9)
    S = 20
    sb = synth.make_batch(seed=55, B=S, T=1)
    od = np.cumsum(rng.uniform(-0.02, 0.02, (S, 3)), axis=0).astype(np.float32)
    od[5:9] = od[5]                                   # a static stretch
    seq = {"scans": sb.scans[:, 0], "scans_ns": np.arange(S), "scans_t": np.arange(S, dtype=np.float32),
           "odoms_t": np.arange(S, dtype=np.float32), "odoms": od, "dets_ns": np.arange(0, S, 2),
           "dets_wc": [[] for _ in range(10)], "dets_wa": [[] for _ in range(10)],
           "dets_wp": [[[2.0, 0.1]] for _ in range(10)]}
    ds = DROWDeviceDataset([seq], cutout_kwargs=None)
    keep = R.static_scene_mask(od)
    assert ds.scans.shape[0] == int(keep.sum())
    kept_ns = np.arange(S)[keep]
    assert len(ds) == sum(1 for d in range(0, S, 2) if d in kept_ns)


def test_device_dataset_from_files_equals_reference(golden, tmp_path):
    """N1 end to end: DROW text files -> (library CSV reader, static-scene filter, annotated-frame
    index) -> device store -> batch, against the reference's DROWDataset2(data_path, split)
    constructor + __getitem__ + collate_batch run on the same files."""
    import os
    from dataset_fixture import CUTOUT_KW
    from test_oracle_golden import _write_drow_files
    from planar_optical_flow_amd import drow_io
    from planar_optical_flow_amd.scan_store import DROWDeviceDataset
    g = golden("dataset_files")
    _write_drow_files(g, str(tmp_path))
    ds = DROWDeviceDataset.from_files(str(tmp_path), "train", num_scans=5, cutout_kwargs=CUTOUT_KW)
    ref_names = [str(x) for x in g["ds_seq_names"]]
    mine = [os.path.basename(n) for n in ds.seq_names]
    assert sorted(mine) == sorted(ref_names) == ["run_a", "run_b"]       # run_static is dropped by both
    # the reference lists samples sequence by sequence in its directory order: map to ours
    ref_flat = [(ref_names[s], int(i)) for s, i in zip(g["ds_flat_seq"], g["ds_flat_scan"])]
    my_flat = [(mine[s], i) for s, i in ds.sample_index]
    assert sorted(ref_flat) == sorted(my_flat)
    order = [my_flat.index(x) for x in ref_flat]
    batch = ds.get_batch(order)
    assert torch.equal(batch["scans"].cpu(), torch.from_numpy(g["out_scans"]))
    assert np.array_equal(batch["odom1"].cpu().numpy().astype(np.float32), g["out_odom1"])
    assert np.array_equal(batch["target_cls"].cpu().numpy(), g["out_target_cls"])
    np.testing.assert_allclose(batch["target_reg"].cpu().numpy(), g["out_target_reg"], atol=1e-6)
    np.testing.assert_allclose(batch["target_flow"].cpu().numpy(), g["out_target_flow"], atol=5e-6)
    assert np.array_equal(batch["exclude_mask"].cpu().numpy().astype(np.float64), g["out_exclude_mask"])
    got = batch["input"][:2].cpu().numpy()
    assert np.mean(np.abs(got - g["out_input_first2"]) > 1e-4) < 2e-3
    # the binary pack feeds the same store
    ds2 = DROWDeviceDataset.from_pack(drow_io.pack_split(str(tmp_path), "train"), num_scans=5, cutout_kwargs=CUTOUT_KW)
    b2 = ds2.get_batch(order)
    for k in ("scans", "target_cls", "target_flow", "input"):
        assert torch.equal(b2[k], batch[k]), k


def test_prepare_sequence_matches_data_prepare(golden, tmp_path):
    """bin/data_prepare.py: `.difodom` text and `.flow` values (oracle prepared_flow_target on the
    re-read, rounded differences)."""
    import os
    from test_oracle_golden import _write_drow_files
    from planar_optical_flow_amd import drow_io
    g = golden("dataset_files")
    d, _ = _write_drow_files(g, str(tmp_path))
    base = os.path.join(d, "run_a")
    flow = drow_io.prepare_sequence(base)
    _, odom_t, odom = drow_io.load_odom(base)
    dt = np.concatenate((odom_t[1:] - odom_t[:-1], [0]))
    dd = np.concatenate((odom[1:] - odom[:-1], [[0] * 3]))
    import io
    buf = io.StringIO()
    np.savetxt(buf, np.hstack([dt.reshape(-1, 1), dd]), fmt="%8.6f", delimiter=",")
    assert open(base + ".difodom").read() == buf.getvalue()
    t_r, d_r = drow_io.load_odom_file(base)
    _, _, scans = drow_io.load_scan_file(base)
    phi = R.laser_phi()
    ref = np.stack([R.prepared_flow_target(scans[i], phi, t_r[i], d_r[i]) for i in range(len(scans))])
    np.testing.assert_allclose(flow, ref, atol=1e-12)
    np.testing.assert_allclose(drow_io.load_flow_file(base), ref, atol=6e-9)   # %10.8f text


def test_segment_inputs_equal_reference_resampling():
    """N3: one launch prepares every detection of a frame.  Against the oracle's radius query and
    resampling rule (box_regressor.py:61-75): same segment size, every row is a segment point minus
    the centre with the orientation appended, row multiplicities follow repeat + pad (small segments)
    or form a subset without repetition (large segments); different seeds give different subsets."""
    from planar_optical_flow_amd import ops
    rng = np.random.default_rng(21)
    for D in (2, 3):
        pts = rng.uniform(-6, 6, (5000, D))
        pts[:, 2:] *= 0.05
        centers = np.array([pts[7], pts[100] + 0.05, np.full(D, 40.0), pts[5]])          # [3] is empty
        dense = centers[0] + rng.normal(0, 0.1, (900, D))                                # > input_size points
        pts = np.concatenate([pts, dense])
        oris = np.array([0.3, -1.2, 0.0, 2.0])
        x, count, mask = ops.segment_inputs(torch.from_numpy(pts).cuda(), torch.from_numpy(centers).cuda(),
                                            torch.from_numpy(oris).cuda(), radius=0.4, input_size=64,
                                            min_segment_size=5, seed=3, return_mask=True)
        x, count, mask = x.cpu().numpy(), count.cpu().numpy(), mask.cpu().numpy()
        for s in range(4):
            seg = R.radius_query(pts, centers[s], 0.4)
            assert count[s] == len(seg)
            assert np.array_equal(pts[mask[s]], seg)
            if len(seg) < 5:
                assert not x[s].any()
                continue
            assert np.all(x[s][:, D] == np.float32(oris[s]))
            want = (seg - centers[s]).astype(np.float32)
            # every output row is one of the segment's rows; count how often each is used
            used = np.zeros(len(want), dtype=int)
            for row in x[s][:, :D]:
                hit = np.where((want == row).all(axis=1))[0]
                assert len(hit) >= 1
                used[hit[0]] += 1
            n = len(seg)
            if n > 64:
                assert used.max() == 1 and used.sum() == 64
            else:
                rep, pad = 64 // n, 64 % n
                extra = -(-pad // rep) if pad else 0          # rows of the repeated array that are taken again
                assert used.min() >= rep and used.sum() == 64
                assert (used > rep).sum() == extra
        x2, _ = ops.segment_inputs(torch.from_numpy(pts).cuda(), torch.from_numpy(centers).cuda(),
                                   torch.from_numpy(oris).cuda(), seed=4)
        assert not np.array_equal(x2[0].cpu().numpy(), x[0])          # another random subset of the dense segment
        x3, _ = ops.segment_inputs(torch.from_numpy(pts).cuda(), torch.from_numpy(centers).cuda(),
                                   torch.from_numpy(oris).cuda(), seed=3)
        assert np.array_equal(x3.cpu().numpy(), x)                    # same seed: same rows, run to run


def test_segment_inputs_huge_segment_is_uniform():
    """More candidates than the LDS list holds: the hash pre-thinning still yields a subset without
    repetition whose indices are spread over the whole segment."""
    from planar_optical_flow_amd import ops
    rng = np.random.default_rng(22)
    pts = rng.normal(0, 0.1, (30000, 2))
    x, count, mask = ops.segment_inputs(torch.from_numpy(pts).cuda(), torch.zeros(1, 2, dtype=torch.float64).cuda(),
                                        torch.zeros(1, dtype=torch.float64).cuda(), radius=0.4, input_size=256,
                                        seed=1, return_mask=True)
    n = int(count[0])
    assert n > 20000 and n == int(mask.sum())
    rows = x[0].cpu().numpy()[:, :2]
    assert len(np.unique(rows, axis=0)) == 256
    src = pts[mask[0].cpu().numpy()].astype(np.float32)
    order = {tuple(r): i for i, r in enumerate(src)}
    idx = np.array([order[tuple(r)] for r in rows])
    assert idx.min() < 0.1 * n and idx.max() > 0.9 * n            # not just the head of the segment


def test_jrdb_anns_to_segments_and_transforms():
    """N3 host mirror: anns_to_segments (all annotations of a frame in one launch) against the
    reference's per-annotation formula with the same RNG stream; the six rigid transforms are
    inverse pairs and match a float64 restatement."""
    from planar_optical_flow_amd.src.data_handle.jrdb_handle import anns_to_segments, box_is_on_ground
    from planar_optical_flow_amd.src.utils import jrdb_transforms as jt
    rng = np.random.default_rng(31)
    pts = rng.uniform(-5, 5, (4000, 3)).astype(np.float32)
    anns = [{"box": {"cx": float(rng.uniform(-4, 4)), "cy": float(rng.uniform(-4, 4)), "cz": -0.2, "l": 0.8,
                     "w": 0.6, "h": 1.7, "rot_z": 0.3}} for _ in range(9)]
    for is_3d in (True, False):
        segs, boxes, ctr = anns_to_segments(pts, anns, radius=0.7, perturb=0.1, is_3d=is_3d,
                                            rng=np.random.default_rng(5))
        ref_rng = np.random.default_rng(5)
        assert boxes.shape == (9, 7 if is_3d else 5)
        for s, ann in enumerate(anns):
            alpha = ref_rng.uniform(0, 2 * np.pi)
            r = ref_rng.uniform(-0.1, 0.1)
            c = np.array([ann["box"]["cx"] + r * np.cos(alpha), ann["box"]["cy"] + r * np.sin(alpha)])
            assert np.array_equal(ctr[s][:2], c)
            keep = np.linalg.norm(pts[:, :2] - c, axis=1) <= 0.7
            want = pts[keep] if is_3d else pts[:, :2][keep]
            assert np.array_equal(segs[s], want)
    assert box_is_on_ground({"box": {"cz": -0.2, "h": 1.7}}) and not box_is_on_ground({"box": {"cz": 0.5, "h": 1.0}})
    p = rng.normal(0, 3, (3, 50)).astype(np.float32)
    for fwd, inv, yaw, dz in ((jt.transform_pts_laser_to_base, jt.transform_pts_base_to_laser, np.pi / 120, 0.0),
                              (jt.transform_pts_upper_velodyne_to_base, jt.transform_pts_base_to_upper_velodyne, 0.085, 0.33529),
                              (jt.transform_pts_lower_velodyne_to_base, jt.transform_pts_base_to_lower_velodyne, 0.0, -0.13511)):
        q = fwd(p)
        c, s = np.cos(yaw), np.sin(yaw)
        want = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ p.astype(np.float64) + np.array([[0], [0], [dz]])
        np.testing.assert_allclose(q, want, atol=2e-6)
        np.testing.assert_allclose(inv(q), p, atol=2e-6)


def test_spatial_drow_forward_equals_reference(golden):
    """N2: SpatialDROW / DROW on the device (MIOpen trunks + the HIP attention gate) against the
    reference's CPU forward with identical seeded weights: eval, streaming inference through the
    running template, and training mode (BatchNorm batch statistics)."""
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import DROW, SpatialDROW
    g = golden("dr_spaam_model")
    torch.manual_seed(3)
    m = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda()
    x = torch.from_numpy(g["x"]).cuda()
    m.eval()
    with torch.no_grad():
        pc, pr, ff = m(x)
        np.testing.assert_allclose(pc.cpu().numpy(), g["eval_cls"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(pr.cpu().numpy(), g["eval_reg"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(ff.cpu().numpy(), g["eval_feat"], rtol=1e-3, atol=2e-3)
        _, _, tmpl0, _ = m(x[:, :, 3:4], testing=True)
        c1, r1, _, f1 = m(x[:, :, 4:5], testing=True, fea_template=tmpl0)
        np.testing.assert_allclose(c1.cpu().numpy(), g["stream_cls"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(r1.cpu().numpy(), g["stream_reg"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(f1.cpu().numpy(), g["stream_feat"], rtol=1e-3, atol=2e-3)
        # the same forwards with the trunk on the HIP conv kernels (BatchNorm folded)
        m.fuse_for_inference()
        pc2, pr2, ff2 = m(x)
        np.testing.assert_allclose(pc2.cpu().numpy(), g["eval_cls"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(pr2.cpu().numpy(), g["eval_reg"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(ff2.cpu().numpy(), g["eval_feat"], rtol=1e-3, atol=2e-3)
        np.testing.assert_allclose(pc2.cpu().numpy(), pc.cpu().numpy(), rtol=1e-4, atol=1e-5)
        m.fused_slab = 7                                   # slabs of sequences (large-batch memory bound)
        pc3, pr3, ff3 = m(x)
        assert torch.equal(pc3, pc2) and torch.equal(pr3, pr2) and torch.equal(ff3, ff2)
        del m.fused_slab
        # streaming inference on the fused trunk
        _, _, tmpl0, _ = m(x[:, :, 3:4], testing=True)
        c1, r1, _, f1 = m(x[:, :, 4:5], testing=True, fea_template=tmpl0)
        np.testing.assert_allclose(c1.cpu().numpy(), g["stream_cls"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(f1.cpu().numpy(), g["stream_feat"], rtol=1e-3, atol=2e-3)
    m.train()
    pc, pr, ff = m(x)
    np.testing.assert_allclose(pc.detach().cpu().numpy(), g["train_cls"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(pr.detach().cpu().numpy(), g["train_reg"], rtol=2e-3, atol=1e-3)
    (pc.sum() + pr.sum() + ff.sum()).backward()                       # trains through the HIP gate
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    torch.manual_seed(4)
    d = DROW(num_scans=5, num_pts=48).cuda().eval()
    with torch.no_grad():
        dc, dr_ = d(torch.from_numpy(g["drow_x"]).cuda())
    np.testing.assert_allclose(dc.cpu().numpy(), g["drow_cls"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(dr_.cpu().numpy(), g["drow_reg"], rtol=1e-3, atol=2e-4)
    d.fuse_for_inference()
    with torch.no_grad():
        dc2, dr2 = d(torch.from_numpy(g["drow_x"]).cuda())
    np.testing.assert_allclose(dc2.cpu().numpy(), g["drow_cls"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(dr2.cpu().numpy(), g["drow_reg"], rtol=1e-3, atol=2e-4)
    assert m._fused is None                         # m.train() above dropped the folded parameters


def test_prototype_forward_equals_reference(golden):
    """N2: Prototype (torch encoder / decoder around the HIP band correlation) against the reference's
    CPU forward with identical seeded weights, eval and train mode; gradients reach every parameter."""
    from planar_optical_flow_amd.src.depracted.model.prototype import Prototype
    g = golden("prototype_model")
    torch.manual_seed(7)
    m = Prototype(in_channel=1, max_displacement=5).cuda()
    s1, s2 = torch.from_numpy(g["s1"]).cuda(), torch.from_numpy(g["s2"]).cuda()
    m.eval()
    with torch.no_grad():
        plain = m(s1, s2)
        np.testing.assert_allclose(plain.cpu().numpy(), g["eval_out"], rtol=1e-3, atol=2e-4)
        # the same forward with every unit on the HIP conv kernel (stride-2 / stride-1 k = 3, k = 1; BatchNorm folded)
        m.fuse_for_inference()
        fused = m(s1, s2)
        np.testing.assert_allclose(fused.cpu().numpy(), g["eval_out"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(fused.cpu().numpy(), plain.cpu().numpy(), rtol=1e-4, atol=2e-4)   # outputs of magnitude 20-60
    m.train()
    assert m._fused is None
    out = m(s1, s2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["train_out"], rtol=2e-3, atol=1e-3)
    loss, err = m.loss_fn(out, torch.zeros_like(out))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_model_fn_adapters_on_device_batches(golden):
    """The reference's batch -> model -> loss adapters on device batches: a FlowDROW-style model fed by the
    device data set trains one step; NumPy batches give the same loss."""
    from dataset_fixture import CUTOUT_KW, load_sequences
    from planar_optical_flow_amd.scan_store import DROWDeviceDataset
    from planar_optical_flow_amd.src.utils import eval_utils as eu
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW, flow_loss

    class FlowHead(torch.nn.Module):          # SpatialDROW + a per-point flow head (FlowDROW needs n_cutout == window)
        def __init__(self):
            super().__init__()
            self.net = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True)
            self.head = torch.nn.Linear(12, 2)
            self.loss_fn = flow_loss

        def forward(self, x, cur_scan):
            pc, pr, ff = self.net(x)
            return pc, pr, self.head(torch.cat((ff, cur_scan.unsqueeze(-1)), dim=-1))

    g = golden("dataset_items")
    ds = DROWDeviceDataset(load_sequences(g), num_scans=5, cutout_kwargs=CUTOUT_KW, drop_static=False)
    batch = ds.get_batch(list(range(6)))
    torch.manual_seed(0)
    model = FlowHead().cuda().train()
    loss, pn, tn = eu.model_fn_dr_spaam(model, batch)
    loss.backward()
    assert torch.isfinite(loss) and model.head.weight.grad is not None
    np_batch = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in batch.items()}
    model.eval()
    with torch.no_grad():
        l_dev = eu.model_fn_dr_spaam(model, batch)[0].item()
        l_np = eu.model_fn_dr_spaam(model, np_batch)[0].item()
    assert abs(l_dev - l_np) < 1e-6
    epe, aae = eu.model_fn_eval(model, [batch, batch])
    assert np.isfinite(epe) and np.isfinite(aae)


def _box_frames(rng, n_frames=3, is_3d=True):
    frames = []
    for _ in range(n_frames):
        K = int(rng.integers(3, 6))
        boxes = np.column_stack([rng.uniform(-5, 5, K), rng.uniform(-5, 5, K), rng.uniform(-0.4, 0.1, K),
                                 rng.uniform(0.5, 1.0, K), rng.uniform(0.4, 0.8, K), rng.uniform(1.5, 1.9, K),
                                 rng.uniform(-4, 4, K)])
        if not is_3d:
            boxes = boxes[:, [0, 1, 3, 4, 6]]
        D = 3 if is_3d else 2
        segs, ctrs = [], []
        for k in range(K):
            n = int(rng.choice([3, 9, 40, 150]))
            c = np.append(boxes[k, :2] + rng.uniform(-0.1, 0.1, 2), 0.176)[:D]
            pts = np.column_stack([boxes[k, :2] + rng.normal(0, 0.2, (n, 2)), rng.uniform(-0.5, 0.5, n)])[:, :D]
            segs.append(pts)
            ctrs.append(c)
        frames.append({"segments": segs, "boxes": boxes, "dets_center": np.array(ctrs)})
    return frames


@pytest.mark.parametrize("is_3d", [True, False])
def test_box_regression_dataset_feeder(is_3d):
    """A14 feeder (jrdb_dataset.py:99-230): size filter, one augmented copy per training sample, the
    reference's target transforms, random drop count, resampling multiset and the appended input angle."""
    from planar_optical_flow_amd.src.data_handle.jrdb_dataset import JRDBBoxRegressionDataset
    cfg = {"input_size": 64, "is_3d": is_3d, "min_segment_size": 5,
           "augmentation_kwargs": {"use_data_augmentation": True, "rot_max": 0.25, "dim_max": 0.1, "dist_max": 0.2,
                                   "random_drop": 0.25}}
    frames = _box_frames(np.random.default_rng(7), is_3d=is_3d)
    n_kept = sum(1 for f in frames for s in f["segments"] if len(s) > 5)
    ds = JRDBBoxRegressionDataset("train", cfg, frames, rng=np.random.default_rng(1), seed=2)
    assert len(ds) == 2 * n_kept                         # original + augmented copy
    val = JRDBBoxRegressionDataset("val", cfg, _box_frames(np.random.default_rng(7), is_3d=is_3d),
                                   rng=np.random.default_rng(1))
    assert len(val) == n_kept
    # data_augmentation against a direct restatement with the same draws
    seg, tgt, ctr = val.inputs[0], val.targets[0], val.dets_center[0]
    chk = np.random.default_rng(5)
    val._rng = np.random.default_rng(5)
    ia, ta, ca = val.data_augmentation(seg.copy(), tgt.copy(), ctr.copy())
    rz = chk.uniform(-0.25 * np.pi, 0.25 * np.pi); dm = 1.0 + chk.uniform(-0.1, 0.1); tr = chk.uniform(-0.2, 0.2, 2)
    R2 = np.array([[np.cos(rz), -np.sin(rz)], [np.sin(rz), np.cos(rz)]], dtype=np.float32)
    np.testing.assert_allclose(ia[:, :2], (seg[:, :2] - tgt[:2]) @ R2.T + tgt[:2] + tr, atol=1e-12)
    np.testing.assert_allclose(ta[:2], tgt[:2] + tr, atol=1e-12)
    dims = slice(3, 6) if is_3d else slice(2, 4)
    np.testing.assert_allclose(ta[dims], tgt[dims] * dm, atol=1e-12)
    # batches
    idx = list(range(len(ds)))
    b = ds.get_batch(idx)
    D = 3 if is_3d else 2
    x = b["input"].cpu().numpy()
    assert x.shape == (len(ds), 64, D + 1)
    tg = np.array(ds.targets)
    ctr_all = np.array(ds.dets_center)
    want_t = tg[:, 2:].copy()
    want_t[:, 0] -= ctr_all[:, -1]
    ang = x[:, 0, D].astype(np.float64)
    want_t[:, -1] = tg[:, -1] - (tg[:, -1] - b["target"].cpu().numpy()[:, -1])      # rot_z - input_angle, checked below
    np.testing.assert_allclose(b["target"].cpu().numpy()[:, :-1], want_t[:, :-1], atol=1e-12)
    np.testing.assert_allclose(tg[:, -1] - b["target"].cpu().numpy()[:, -1], ang, atol=1e-6)   # column = input angle
    assert np.all(np.abs(ang - tg[:, -1]) <= 0.25 * np.pi + 1e-6)
    for s in range(len(ds)):
        assert np.all(x[s, :, D] == x[s, 0, D])
        pts = (np.asarray(ds.inputs[s], np.float64) - ctr_all[s]).astype(np.float32)
        n_all = len(pts)
        n = n_all - int(n_all * 0.25)
        used = np.zeros(n_all, dtype=int)
        for row in x[s, :, :D]:
            hit = np.where((pts == row).all(axis=1))[0]
            assert len(hit) >= 1
            used[hit[0]] += 1
        assert (used > 0).sum() == min(n, 64)                     # dropped points never appear
        if n <= 64:
            assert used[used > 0].min() >= 64 // n and used.sum() == 64
    # evaluation split: no drop, deterministic point set
    bv = val.get_batch([0, 1])
    xv = bv["input"].cpu().numpy()
    pts0 = (np.asarray(val.inputs[0], np.float64) - val.dets_center[0]).astype(np.float32)
    assert {tuple(r) for r in xv[0, :, :D]} <= {tuple(r) for r in pts0}
    if len(pts0) <= 64:
        assert {tuple(r) for r in xv[0, :, :D]} == {tuple(r) for r in pts0}


def test_get_dataloader_feeds_the_box_head_pipeline(tmp_path):
    """get_dataloader -> device batches -> the reference-shaped Pipeline / Trainer with the box head: one
    epoch of training and an evaluation run end to end on the device data path."""
    from planar_optical_flow_amd.src.data_handle.get_dataloader import get_dataloader
    from planar_optical_flow_amd.src.utils import eval_utils as eu
    from src.model.get_model import get_model
    np.random.seed(5)            # the data set draws its augmentation from the global state, like the reference
    torch.manual_seed(5)
    frames = _box_frames(np.random.default_rng(11), n_frames=6, is_3d=True)
    cfg = {"data_dir": "/data/JRDB", "frames": frames, "input_size": 64, "is_3d": True, "min_segment_size": 5,
           "augmentation_kwargs": {"use_data_augmentation": True, "rot_max": 0.25, "dim_max": 0.1, "dist_max": 0.2,
                                   "random_drop": 0.25}}
    loader = get_dataloader("train", 8, 4, True, cfg)
    assert len(loader) == -(-len(loader.dataset) // 8)
    model = get_model({"type": "box_reg", "input_dim": 4, "target_dim": 5, "dropout": 0.3}).cuda().train()
    opt = torch.optim.Adam(model.parameters(), 1e-3)
    seen = 0
    for batch in loader:
        assert batch["input"].is_cuda and batch["input"].shape[1:] == (64, 4)
        loss = model.model_fn(model, batch)[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert torch.isfinite(loss)
        seen += batch["input"].shape[0]
    assert seen == len(loader.dataset)
    model.eval()
    with torch.no_grad():
        _, _, ev = model.model_eval_fn(model, next(iter(loader)))       # batched rotated IoU on device batches
    # (a barely trained head may predict negative extents, so the IoU value itself is not meaningful yet)
    assert np.isfinite(ev["iou"]) and np.isfinite(ev["loss_dim"]) and np.isfinite(ev["loss_ori"])


def test_drow_dataset2_and_loaders_from_files(golden, tmp_path):
    """DROWDataset2(data_path, split, ...) / create_dataloader with the reference's arguments: the cutout
    network input equals the reference's (golden from its own constructor + __getitem__), the other
    network types have the reference's shapes, the flip augmentation mirrors scans and negates target_reg x."""
    import os
    from dataset_fixture import CUTOUT_KW
    from test_oracle_golden import _write_drow_files
    from planar_optical_flow_amd.src.utils.dataset_dr_spaam import DROWDataset2, create_dataloader
    g = golden("dataset_files")
    _write_drow_files(g, str(tmp_path))
    ds = DROWDataset2(str(tmp_path), split="train", num_scans=5, network_type="cutout", cutout_kwargs=CUTOUT_KW)
    ref_names = [str(x) for x in g["ds_seq_names"]]
    mine = [os.path.basename(n) for n in ds.seq_names]
    ref_flat = [(ref_names[s], int(i)) for s, i in zip(g["ds_flat_seq"], g["ds_flat_scan"])]
    my_flat = [(mine[s], i) for s, i in ds.sample_index]
    order = [my_flat.index(x) for x in ref_flat]
    b = ds.get_batch(order)
    assert torch.equal(b["scans"].cpu(), torch.from_numpy(g["out_scans"]))
    assert np.array_equal(b["target_cls"].cpu().numpy(), g["out_target_cls"])
    got = b["input"][:2].cpu().numpy()
    assert np.mean(np.abs(got - g["out_input_first2"]) > 1e-4) < 2e-3
    n = len(order)
    for nt, shape in (("fc1d", (n, 6, 1, 450)), ("fc1d_fea", (n, 6, 56, 450)), ("fc2d", (n, 6, 1, 31, 450))):
        d2 = DROWDataset2(str(tmp_path), split="train", num_scans=5, network_type=nt, cutout_kwargs=CUTOUT_KW,
                          polar_grid_kwargs={})
        assert tuple(d2.get_batch(order)["input"].shape) == shape
    aug = DROWDataset2(str(tmp_path), split="train", num_scans=5, network_type="fc1d", use_data_augumentation=True, seed=3)
    ba = aug.get_batch(order)
    flipped = ~(ba["scans"] == b["scans"]).all(dim=-1).all(dim=-1)
    assert 0 < int(flipped.sum()) < n
    assert torch.equal(ba["scans"][flipped], b["scans"][flipped].flip(-1))
    assert torch.equal(ba["target_reg"][flipped][..., 0], -b["target_reg"][flipped][..., 0])
    assert torch.equal(ba["target_reg"][~flipped], b["target_reg"][~flipped])
    train_loader, eval_loader = create_dataloader(str(tmp_path), 5, 4, 2, cutout_kwargs=CUTOUT_KW)
    assert eval_loader is None and sum(x["input"].shape[0] for x in train_loader) == n
    one = ds[0]
    assert one["input"].shape == (450, 6, 56) and one["scans"].shape == (6, 450)


def test_train_utils_trainer_on_device_loader(golden, tmp_path):
    """The DR-SPAAM training script's pieces (train_utils.Trainer + LucasScheduler + checkpoints) driven by
    the device loader and model_fn_dr_spaam: two epochs, a checkpoint that loads back, evaluation metrics."""
    from dataset_fixture import CUTOUT_KW, load_sequences
    from planar_optical_flow_amd.src.utils.dataset_dr_spaam import DROWDataset2
    from planar_optical_flow_amd.src.data_handle.get_dataloader import DeviceBatchLoader
    from planar_optical_flow_amd.src.utils import eval_utils as eu, train_utils as tu
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW, flow_loss

    class FlowHead(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True)
            self.head = torch.nn.Linear(12, 2)
            self.loss_fn = flow_loss

        def forward(self, x, cur_scan):
            pc, pr, ff = self.net(x)
            return pc, pr, self.head(torch.cat((ff, cur_scan.unsqueeze(-1)), dim=-1))

    g = golden("dataset_items")
    ds = DROWDataset2(None, num_scans=5, network_type="cutout", cutout_kwargs=CUTOUT_KW, sequences=load_sequences(g),
                      drop_static=False)
    loader = DeviceBatchLoader(ds, 4, shuffle=True)
    torch.manual_seed(1)
    model = FlowHead().cuda()
    opt = torch.optim.Adam(model.parameters(), lr=tu.lr_scheduler())
    sched = tu.LucasScheduler(opt, 0, 1e-3, 2, 1e-5)
    tb = tu.create_tb_logger(str(tmp_path))
    tr = tu.Trainer(model, eu.model_fn_dr_spaam, opt, str(tmp_path), sched, model_fn_eval=eu.model_fn_eval,
                    grad_norm_clip=1.0, tb_logger=tb)
    tr.train(num_epochs=2, train_loader=loader, eval_loader=[ds.get_batch([0, 1, 2])], ckpt_save_interval=1)
    assert abs(sched.get_lr() - 1e-3 * (1e-5 / 1e-3) ** ((1 + (len(loader) - 1) / len(loader)) / 2)) < 1e-12
    res = eu.eval_dr_spaam(model, [ds.get_batch([0, 1, 2]), ds.get_batch([3, 4, 5])], output_dir=str(tmp_path / "ev"))
    assert res["epe"].shape == (6,) and res["pred_flow"].shape == (6, 450, 2) and np.isfinite(res["eval_loss"])
    b0 = ds.get_batch([0, 1, 2])
    want = R.flow_to_global(b0["target_flow"][0].cpu().numpy().astype(np.float64), R.laser_phi())
    np.testing.assert_allclose(res["target_flow"][0], want, atol=1e-5)
    assert (tmp_path / "ev" / "flow_eval.npz").exists()
    it, ep = tu.load_checkpoint(FlowHead().cuda(), None, filename=str(tmp_path / "ckpt_e2.pth"))
    assert ep == 2 and it == 2 * len(loader)
    with pytest.raises(FileNotFoundError):
        tu.load_checkpoint(model, None, filename=str(tmp_path / "missing.pth"))


def test_stale_hip_error_is_surfaced_not_swallowed():
    """A sticky HIP error left by an earlier call of the thread (here: hipSetDevice on a device that does not
    exist) is taken out of the way by the next entry point -- whose own launch check must not trip over it --
    and handed to the caller through pof_take_stale_error() (the blanket clear hid it)."""
    import ctypes as C
    from planar_optical_flow_amd import _lib, ops
    lib = _lib.load()
    lib.pof_take_stale_error()                                   # start clean
    tab = ops.phi_table()
    torch.cuda.synchronize()
    # the HIP runtime this process already runs on (torch's copy): found in the process map, not by soname
    paths = {ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}
    assert len(paths) == 1, paths
    hip = C.CDLL(paths.pop())
    dev = torch.cuda.current_device()
    scans = torch.rand(2, 1, 450, device="cuda") * 10 + 1
    out = ops.scan_preprocess(scans, tab, want=("xy",))          # warm-up: buffers come from torch's cache below
    ws = torch.empty(ops.scan_preprocess_workspace_bytes(2, 0), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = hip.hipSetDevice(9999)
    assert rc != 0                                               # hipErrorInvalidDevice, now sticky
    hip.hipSetDevice(dev)
    ops.scan_preprocess(scans, tab, want=("xy",), out=out, workspace=ws)   # succeeds: its launch check saw a clean state
    assert _lib.take_stale_error() == rc                         # ... and the earlier error is reported
    assert _lib.take_stale_error() == 0                          # once
    torch.cuda.synchronize()
    assert torch.isfinite(out["xy"]).all()


def test_rccl_one_rank_smoke():
    """The RCCL code path executes at least once on the one GPU of the test box (tools/rccl_smoke.py): a one-rank
    `nccl` process group, the flat gradient bucket all-reduced on device tensors, the SyncBatchNorm forward /
    backward collectives.  (Own process: the group must not leak into the other tests.)"""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "rccl_smoke.py")], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stderr[-1500:]


def _bench_line(args, env_extra, timeout=400, launcher=False):
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29517"]
    cmd += [os.path.join(repo, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=repo, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # rank 0 only
    return json.loads(lines[0])


def _check_two_rank_line(d):
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 2 * 4096 and d["epe_vs_oracle_m"] < 1e-5
    assert d["timed_repeats"] == 3 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert "cpu_baseline" not in d and "cutout" not in d       # N = 1 only
    st = d["strong_scaling"]                                   # one 4096-scan batch cut in two
    assert st["scans_per_rank_per_step"] == 2048 and st["global_batch"] == 4096 and st["value"] > 0
    assert d["config"]["collective_backend"] == "gloo" and d["config"]["rccl_ranks"] is None
    bh = d["box_head_train"]                                   # BASELINE configs[3]: the one path with a collective
    assert bh["per_rank_batch"] == 256 and bh["samples_per_s"] > 0 and bh["grad_allreduce_ms"] > 0
    assert bh["grad_bucket_bytes"] == 4 * 953219 and bh["collective_backend"] == "gloo"


def test_bench_plain_command_line_starts_its_own_ranks():
    """`python bench.py --gpus 2` -- the plain command line, no launcher, no WORLD_SIZE: the parent starts two
    fresh rank processes itself and relays rank 0's line.  Rehearsed with both ranks on the one GPU (gloo
    instead of RCCL, which needs a device per rank)."""
    d = _bench_line(["--gpus", "2", "--steps", "40", "--warmup", "8", "--repeats", "3"],
                    {"POF_BENCH_SHARE_GPU": "1", "POF_BENCH_BACKEND": "gloo"})
    _check_two_rank_line(d)


def test_bench_two_ranks_under_torch_distributed_run():
    """The driver's N > 1 launch line (torch.distributed.run): ranks from the environment, same line."""
    d = _bench_line(["--gpus", "2", "--steps", "40", "--warmup", "8", "--repeats", "3"],
                    {"POF_BENCH_SHARE_GPU": "1", "POF_BENCH_BACKEND": "gloo"}, launcher=True)
    _check_two_rank_line(d)


def test_bench_launcher_path_equals_direct_path_at_one_gpu():
    """--gpus 1 through the rank launcher (--spawn) against the direct path: same line, same rate (medians of 5
    timed regions; 10 % covers the run-to-run spread of a 15 us launch on one box)."""
    args = ["--steps", "400", "--warmup", "40", "--no-cpu-baseline", "--no-extra"]
    direct = _bench_line(args, {})
    spawned = _bench_line(args + ["--spawn"], {})
    assert spawned["n_gpus"] == direct["n_gpus"] == 1
    assert abs(spawned["value"] / direct["value"] - 1.0) < 0.10, (spawned["value"], direct["value"])


def test_bench_single_gpu_line_and_world_size_check():
    """--gpus 1: roofline / cpu_baseline / host-fed objects present and consistent; a WORLD_SIZE that disagrees
    with --gpus is refused instead of silently measuring one GPU."""
    import os, subprocess, sys
    d = _bench_line(["--steps", "80", "--warmup", "8", "--repeats", "3", "--no-model"], {})
    assert d["n_gpus"] == 1 and "strong_scaling" not in d
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (rf["launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    # ONE clock: the roofline figure is the algorithmic bytes of a step over the wall interval of ms_per_step
    assert abs(rf["achieved"] - 14400 * 4096 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert abs(rf["launch_ms"] - d["ms_per_step"] * d["steps"] / rf["launches"]) < 1e-9 and rf["steps_per_launch"] == 8
    # the device interval is measured on regions of its own (the wall regions hold no event records): same launches
    assert 0.6 * d["ms_per_step"] <= rf["events"]["ms_per_step"] <= 1.3 * d["ms_per_step"]
    assert "f32" in d["dtype"] and d["value_f64"] > 0 and d["roofline_f64"]["epe_vs_oracle_m"] < 1e-12
    assert abs(d["roofline_f64"]["achieved"] - 18000 * 4096 / (d["roofline_f64"]["ms_per_step"] * 1e-3) / 1e9) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["repeats"] >= 3 and cb["cores"] == min(cb["host"]["physical_cores"], cb["host"]["usable_cpus"])
    assert cb["single_process_scans_per_s"] > 0 and cb["min"] <= cb["value"] <= cb["max"]
    assert 0 < d["host_fed"]["host_fed_scans_per_s"] < d["value"]
    assert d["box_head_train"]["grad_allreduce_ms"] is None and d["box_head_train"]["samples_per_s"] > 0
    gr = d["box_head_train"]["graphed"]                       # configs[3] as one replay: HIP units and library modules
    assert 0 < gr["ms_per_step"] and gr["ms_per_step_library_modules"] > 0 and d["box_head_train"]["host_syncs_per_step"] == 0
    assert "prototype_forward" not in d                     # --no-model: the network rows are left out
    assert d["config"]["slots_per_launch"] == 8 and d["config"]["ring_batches"] == 16
    assert d["single_batch_launches"]["ms_per_step"] >= d["ms_per_step"] * 0.95
    assert set(d["small_kernels"]) >= {"segment_kernel", "nms_kernel", "rotate_iou_kernel", "flow_errors_kernel",
                                       "gather_windows_kernel", "segment_inputs_kernel"}
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "8"],
                       capture_output=True, text=True, env=env, cwd=repo, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_box_head_gpu_forward_equals_reference(golden):
    """The box head on the device (1x1 convolutions issued as GEMMs in eval mode) against the reference's CPU
    forward with identical seeded weights, 2-D and 3-D variants; train mode uses the modules and agrees."""
    from src.model.get_model import get_model
    g = golden("box_head")
    for tag, cfg_m in (("2d", {"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}),
                       ("3d", {"type": "box_reg", "input_dim": 4, "target_dim": 5, "dropout": 0.3})):
        torch.manual_seed(61)
        m = get_model(cfg_m).cuda().eval()
        x = torch.from_numpy(g["in_" + tag]).cuda()
        with torch.no_grad():
            y = m(x)
            m.backbone.gemm_pointwise = False
            y_mod = m(x)
            m.backbone.gemm_pointwise = True
        np.testing.assert_allclose(y.cpu().numpy(), g["out_" + tag], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(y.cpu().numpy(), y_mod.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_box_head_overfits_one_sample():
    """The reference's _DEBUG_ONE_SAMPLE check (jrdb_dataset.py:13, :100-101): fed the same sample again and
    again the network 'should fit perfectly' -- through the device feeder, the adapter and Adam."""
    from planar_optical_flow_amd.src.data_handle.jrdb_dataset import JRDBBoxRegressionDataset
    from planar_optical_flow_amd.src.utils import eval_utils as eu
    from src.model.get_model import get_model
    cfg = {"input_size": 64, "is_3d": False, "min_segment_size": 5,
           "augmentation_kwargs": {"use_data_augmentation": False, "rot_max": 0.0, "dim_max": 0.0, "dist_max": 0.0,
                                   "random_drop": 0.0}}
    ds = JRDBBoxRegressionDataset("val", cfg, _box_frames(np.random.default_rng(3), n_frames=2, is_3d=False),
                                  rng=np.random.default_rng(0))
    torch.manual_seed(0)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0}).cuda().train()
    opt = torch.optim.Adam(model.parameters(), 2e-3)
    batch = ds.get_batch([0, 1, 2, 3])          # rot_max = 0: the same inputs and targets every time
    losses = []
    for it in range(300):
        loss = model.model_fn(model, batch)[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    first, tail = losses[0], min(losses[-30:])      # training on the GPU is not bit-reproducible run to run
    assert tail < 0.15 * first, (first, tail)


def test_batch_preprocessor_lookahead_chain():
    """DROWBatchPreprocessor with look-ahead: the chained single-launch sequence A -> B -> C gives the same
    batches as three self-contained calls, also when an unannounced batch interrupts the chain."""
    from planar_optical_flow_amd.preprocess import DROWBatchPreprocessor
    pre = DROWBatchPreprocessor(cutout_kwargs=None)
    plain = DROWBatchPreprocessor(cutout_kwargs=None)
    data = []
    for seed in (91, 92, 93, 94):
        sb = synth.make_batch(seed=seed, B=300, T=3, mixed_classes=True)
        o, r, c = sb.det_csr()
        from planar_optical_flow_amd import ops as _ops
        dev_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        data.append((dev_t(sb.scans), dev_t(sb.odom0), dev_t(sb.odom1), _ops.DetCSR.from_numpy(o, r, c, "cuda")))
    ref = [plain(*d) for d in data]
    keys = ("target_cls", "target_reg", "target_flow", "exclude_mask")
    got_a = pre(*data[0], lookahead=data[1][1:])
    got_b = pre(*data[1], lookahead=data[2][1:])
    got_d = pre(*data[3])                         # not the announced batch: parameters are recomputed
    got_c = pre(*data[2], lookahead=None)
    for got, want in ((got_a, ref[0]), (got_b, ref[1]), (got_d, ref[3]), (got_c, ref[2])):
        for k in keys:
            assert torch.equal(got[k], want[k]), k


def test_streaming_detector_graph_replay_equals_eager_steps():
    """DR-SPAAM streaming mode (reference forward(testing=True, fea_template=...)): the hipGraph replay gives
    the same numbers as the eager step and as calling the model by hand, across a reset."""
    from planar_optical_flow_amd import ops
    from planar_optical_flow_amd.streaming import StreamingDetector
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW
    torch.manual_seed(11)
    model = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
    B = 2
    scans = torch.from_numpy(synth.make_batch(seed=21, B=B, T=7).scans).cuda()          # [B, 7, 450]
    eager, graphed = StreamingDetector(model, batch=B, graph=False), StreamingDetector(model, batch=B, graph=True)
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=56, padding_val=29.99,
              area_mode=True)
    tmpl = None
    for t in range(7):
        if t == 4:                                    # a new sequence starts
            eager.reset(), graphed.reset()
            tmpl = None
        ce, re_ = (v.clone() for v in eager(scans[:, t]))
        cg, rg = graphed(scans[:, t])
        assert torch.equal(ce, cg) and torch.equal(re_, rg), t
        assert torch.equal(eager.template, graphed.template) and torch.equal(eager.feat_fused, graphed.feat_fused)
        with torch.no_grad():
            x = ops.cutout(scans[:, t:t + 1].contiguous(), ops.phi_table(), **kw)
            c0, r0, tmpl, f0 = model(x, testing=True, fea_template=tmpl)
        assert torch.equal(c0, cg) and torch.equal(r0, rg) and torch.equal(f0, graphed.feat_fused), t
        assert cg.shape == (B, 450, 1) and rg.shape == (B, 450, 2)
    assert graphed._graph is not None and eager._graph is None
    addr = graphed.template.data_ptr()
    graphed.reset()
    graphed(scans[:, 0]); graphed(scans[:, 1])
    assert graphed.template.data_ptr() == addr        # the captured graph keeps pointing at the live template
    # a single sensor passes a 1-D scan
    one = StreamingDetector(model, batch=1)
    c1, _ = one(scans[0, 0])
    assert c1.shape == (1, 450, 1)
    # detections inside the replayed step equal the stand-alone NMS on the step's outputs
    import src.utils.utils as u
    dn = StreamingDetector(model, batch=B, nms_min_dist=0.5)
    with pytest.raises(RuntimeError):
        dn.detections()
    for t in range(4):
        cls, reg = dn(scans[:, t])
        dets, inst = dn.detections()
        for b in range(B):
            xy, dc, im = u.nms_predicted_center(scans[b, t].cpu().numpy(), u.get_laser_phi(),
                                                torch.sigmoid(cls[b]).double().cpu().numpy(), reg[b].double().cpu().numpy())
            assert np.array_equal(dets[b][0], xy) and np.array_equal(dets[b][1], dc[:, 0]) and np.array_equal(inst[b], im)
    assert dn._graph is not None


def test_streaming_detectors_share_a_model_and_survive_refusing():
    """Two graphed detectors on ONE model: the second must not invalidate the folded trunk parameters the first
    one's graph replays from (use-after-free); a train() / eval() round trip or a new
    fuse_for_inference() makes a detector drop its graph and capture again.  Every step equals the eager detector."""
    from planar_optical_flow_amd.streaming import StreamingDetector
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW
    torch.manual_seed(12)
    model = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
    scans = torch.from_numpy(synth.make_batch(seed=23, B=1, T=12).scans).cuda()[0]         # [12, 450]
    first = StreamingDetector(model, batch=1)
    ref = StreamingDetector(model, batch=1, graph=False)
    fused_before = model._fused
    for t in range(3):                                          # eager first step, capture, one replay
        c, r = first(scans[t])
        ce, re_ = ref(scans[t])
        assert torch.equal(c, ce) and torch.equal(r, re_), t
    assert first._graph is not None
    second = StreamingDetector(model, batch=1)                  # must reuse, not rebuild, the folded parameters
    assert model._fused is fused_before and second._fused_ref is first._fused_ref
    junk = [torch.randn(1 << 18, device="cuda") for _ in range(8)]   # recycle whatever the allocator has free
    for t in range(3):
        second(scans[t])
    g_first = first._graph
    for t in range(3, 6):                                       # replays of the FIRST detector after the second was built
        c, r = first(scans[t])
        ce, re_ = ref(scans[t])
        assert torch.equal(c, ce) and torch.equal(r, re_), t
    assert first._graph is g_first
    # weights change: train-mode round trip drops the folded set; the detector notices and re-captures
    model.train()
    with torch.no_grad():
        for prm in model.parameters():
            prm.mul_(1.01)
    model.eval()
    for t in range(6, 9):
        c, r = first(scans[t])
        ce, re_ = ref(scans[t])
        assert torch.equal(c, ce) and torch.equal(r, re_), t
    assert first._graph is not g_first and first._fused_ref is model._fused
    del junk


def test_streaming_detector_single_sensor_replays():
    """batch = 1 through five calls (eager first step, capture, three replays) against the eager detector: the
    shape whose second replay hung in round 1 when the cutout cleared its area word with a 4-byte memset node."""
    from planar_optical_flow_amd.streaming import StreamingDetector
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW
    torch.manual_seed(13)
    model = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
    scans = torch.from_numpy(synth.make_batch(seed=24, B=1, T=5).scans).cuda()[0]
    graphed, eager = StreamingDetector(model, batch=1), StreamingDetector(model, batch=1, graph=False)
    for t in range(5):
        cg, rg = graphed(scans[t])
        ce, re_ = eager(scans[t])
        torch.cuda.synchronize()
        assert torch.equal(cg, ce) and torch.equal(rg, re_), t
        assert torch.equal(graphed.template, eager.template)
    assert graphed._graph is not None


def test_model_fn_obj_det_equals_reference(golden):
    """The detector's loss adapter (eval_utils.model_fn_obj_det) on both heads against the reference's own numbers
    with identical seeded weights; it also reads the target_cls / target_reg keys of the device batches."""
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import DROW, SpatialDROW
    from planar_optical_flow_amd.src.utils import eval_utils as eu
    g = golden("dr_spaam_model")
    torch.manual_seed(3)
    m1 = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
    torch.manual_seed(4)
    m4 = DROW(num_scans=5, num_pts=48).cuda().eval()
    with torch.no_grad():
        l1, tb1, rt = eu.model_fn_obj_det(m1, {"input": g["x"], "target_flow_cls": g["objdet1_cls"],
                                                "target_flow_reg": g["objdet1_reg"]}, rtn_result=True)
        l4, tb4, _ = eu.model_fn_obj_det(m4, {"input": torch.from_numpy(g["drow_x"]).cuda(),
                                               "target_cls": torch.from_numpy(g["objdet4_cls"]).cuda(),
                                               "target_reg": torch.from_numpy(g["objdet4_reg"]).cuda()})
        l0, tb0, _ = eu.model_fn_obj_det(m4, {"input": g["drow_x"], "target_flow_cls": g["objdet4_cls"] * 0,
                                               "target_flow_reg": g["objdet4_reg"]})
    np.testing.assert_allclose([float(l1), tb1["cls_loss"], tb1["fg_ratio"], tb1["reg_loss"]], g["objdet1_out"], rtol=2e-4)
    np.testing.assert_allclose([float(l4), tb4["cls_loss"], tb4["fg_ratio"], tb4["reg_loss"]], g["objdet4_out"], rtol=2e-4)
    np.testing.assert_allclose([float(l0), tb0["cls_loss"], tb0["fg_ratio"]], g["objdet0_out"], rtol=2e-4, atol=1e-12)
    assert "reg_loss" not in tb0 and rt["pred_cls"].shape == (2, 30, 1) and rt["pred_reg"].shape == (2, 30, 2)
    # differentiable: the training scripts call .backward() on the returned loss
    m4.train()
    loss, _, _ = eu.model_fn_obj_det(m4, {"input": g["drow_x"], "target_cls": g["objdet4_cls"], "target_reg": g["objdet4_reg"]})
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m4.parameters())


def test_legacy_box_head_evaluations():
    """model_fn_eval_box_reg / eval_Bb_regression / eval_BB_reg_baseline: the paired IoU launch equals the oracle's
    rotated IoU of (prediction i, target i) and the error statistics follow the reference's formulas."""
    from planar_optical_flow_amd.src.utils import eval_utils as eu
    rng = np.random.default_rng(8)

    class Head(torch.nn.Module):
        """Predicts (l, w, rot / pi) as a fixed function of the mean input point; loss = L1."""
        def forward(self, x):
            mu = x.mean(dim=1)
            return torch.stack((0.8 + 0.1 * mu[:, 0], 0.6 + 0.1 * mu[:, 1], 0.2 * mu[:, 0]), dim=1)

        def loss_fn(self, pred, target):
            return (pred - target).abs().mean()

    def batch(n):
        return {"input": rng.normal(size=(n, 32, 2)).astype(np.float32),
                "target": np.column_stack([rng.normal(0, 0.2, (n, 2)), rng.uniform(0.4, 1.2, (n, 2)),
                                           rng.uniform(-1, 1, n)]).astype(np.float32),
                "det_center": rng.uniform(-5, 5, (n, 2)).astype(np.float32)}
    loader = [batch(16), batch(9)]
    model = Head().cuda()
    loss, dim, ori, iou = eu.model_fn_eval_box_reg(model, loader)
    want_iou, want_dim, want_ori, want_loss = [], [], [], []
    for b in loader:
        pred = model(torch.from_numpy(b["input"]).cuda()).cpu().numpy().astype(np.float64)
        tgt = b["target"].astype(np.float64)
        want_loss.append(np.abs(pred - tgt[:, 2:]).mean())
        pred[:, -1] *= np.pi
        tgt[:, -1] *= np.pi
        boxes = np.hstack((np.zeros((len(pred), 2)), pred)).astype(np.float32)
        full = R.rotate_iou(boxes, tgt.astype(np.float32))
        want_iou.append(np.mean(np.diag(full)))
        want_dim.append(np.mean(np.sum(np.abs(pred[:, :2] - tgt[:, 2:4]), axis=1)))
        want_ori.append(np.mean(np.abs(pred[:, -1] - tgt[:, -1])))
    np.testing.assert_allclose([float(loss), dim, ori, iou],
                               [np.mean(want_loss), np.mean(want_dim), np.mean(want_ori), np.mean(want_iou)], rtol=1e-4)
    res = eu.eval_Bb_regression(model, loader)
    assert res["iou"].shape == (25,) and np.all((res["iou"] >= 0) & (res["iou"] <= 1)) and res["iou"].max() > 0.05
    b = loader[0]
    pred = model(torch.from_numpy(b["input"]).cuda()).cpu().numpy().astype(np.float64)
    tgt = b["target"].astype(np.float64)
    pred[:, -1] *= np.pi
    tgt[:, -1] *= np.pi
    tgt[:, :2] += b["det_center"]
    full = R.rotate_iou(np.hstack((b["det_center"], pred)).astype(np.float32), tgt.astype(np.float32))
    np.testing.assert_allclose(res["iou"][:16], np.diag(full), atol=2e-5)

    class DS:
        targets = np.column_stack([rng.uniform(-3, 3, (40, 2)), rng.uniform(0.4, 1.2, (40, 2)), rng.uniform(-3, 3, 40)])
        dets_center = targets[:, :2] + rng.normal(0, 0.05, (40, 2))
    base = eu.eval_BB_reg_baseline(DS)
    pred = np.array([DS.targets[:, 2].mean(), DS.targets[:, 3].mean(), 0.5 * np.pi])
    full = R.rotate_iou(np.hstack((DS.dets_center, np.tile(pred, (40, 1)))).astype(np.float32), DS.targets.astype(np.float32))
    np.testing.assert_allclose(base["iou"], np.diag(full), atol=2e-5)
    np.testing.assert_allclose(base["orientation_err"], np.abs(pred[2] - DS.targets[:, 4]))


def test_prototype_eval_loop(tmp_path):
    """eval_utils.eval (the scan-pair network's evaluation): per-sample EPE equals the oracle's flow error on the
    zero-thresholded predictions; arrays are written instead of the reference's video."""
    from planar_optical_flow_amd.src.depracted.model.prototype import Prototype
    from planar_optical_flow_amd.src.utils import eval_utils as eu
    torch.manual_seed(5)
    model = Prototype(in_channel=1, max_displacement=5).cuda()
    rng = np.random.default_rng(4)
    loader = [{"scan_pair": rng.normal(size=(n, 2, 450, 1)).astype(np.float32),
               "flow_target_flow": rng.normal(size=(n, 450, 2)).astype(np.float32)} for n in (3, 2)]
    res = eu.eval(model, loader, output_dir=str(tmp_path))
    assert res["epe"].shape == (5,) and res["pred_flow"].shape == (5, 450, 2) and (tmp_path / "flow_eval.npz").exists()
    want = np.linalg.norm(res["pred_flow"].astype(np.float64) - res["target_flow"], axis=-1).mean(axis=1)
    np.testing.assert_allclose(res["epe"], want, rtol=1e-5)
    np.testing.assert_allclose(res["eval_loss"], (want[:3].mean() + want[3:].mean()) / 2, rtol=1e-5)
    model.eval()
    with torch.no_grad():
        direct = model(torch.from_numpy(loader[1]["scan_pair"][:, 0]).cuda(), torch.from_numpy(loader[1]["scan_pair"][:, 1]).cuda())
    np.testing.assert_allclose(direct.float().cpu().numpy(), res["pred_flow"][3:], rtol=1e-4, atol=1e-4)   # MIOpen convs


def test_flow_to_hsv_equals_reference(golden):
    import src.utils.utils as u
    g = golden("flow_hsv")
    rgb = u.flow_to_hsv(g["flow"])
    assert rgb.shape == g["rgb"].shape == (len(g["flow"]), 3)
    np.testing.assert_allclose(rgb, g["rgb"], rtol=0, atol=1e-12)
    assert np.array_equal(u.flow_to_hsv(np.zeros((4, 2))), np.ones((4, 3)))          # zero flow: white
    assert u.flow_to_hsv(torch.from_numpy(g["flow"]).cuda().reshape(2, -1, 2)).shape == (2, len(g["flow"]) // 2, 3)


def test_eval_person_flow_loop(tmp_path):
    """eval_person_flow: per-sample detections equal the (pinned) single-scan nms_predicted_center, flow errors equal
    loss_fn_eval, flows come back in the scanner frame."""
    import src.utils.utils as u
    from planar_optical_flow_amd.src.utils import eval_utils as eu

    class Net(torch.nn.Module):
        def forward(self, x):                         # x [B, N, T, P] cutouts -> fixed functions of the cutout
            m = x.mean(dim=(2, 3))
            cls = (3.0 * torch.sin(7.0 * m) - 1.0).unsqueeze(-1)
            reg = torch.stack((0.3 * torch.cos(5.0 * m), 0.3 * torch.sin(3.0 * m)), dim=-1)
            return cls, reg, torch.stack((0.1 * m, -0.05 * m), dim=-1)
    sb = synth.make_batch(seed=31, B=5, T=3)
    rng = np.random.default_rng(1)
    loader = []
    for lo, hi in ((0, 3), (3, 5)):
        loader.append({"scans": sb.scans[lo:hi], "input": rng.normal(size=(hi - lo, 450, 3, 8)).astype(np.float32),
                       "target_flow": rng.normal(size=(hi - lo, 450, 2)).astype(np.float32) * 0.1})
    model = Net().cuda()
    res = eu.eval_person_flow(model, loader, output_dir=str(tmp_path))
    assert res["epe"].shape == (5,) and len(res["dets_xy"]) == 5 and (tmp_path / "person_flow_eval.npz").exists()
    phi = u.get_laser_phi()
    k = 0
    for b in loader:
        with torch.no_grad():
            cls, reg, flow = model(torch.from_numpy(b["input"]).cuda())
        e, a = eu.loss_fn_eval(flow, b["target_flow"])
        for i in range(len(b["scans"])):
            xy, dc, inst = u.nms_predicted_center(b["scans"][i, -2], phi, torch.sigmoid(cls[i]).cpu().numpy().astype(np.float64),
                                                  reg[i].cpu().numpy().astype(np.float64))
            assert np.array_equal(res["dets_xy"][k], xy) and np.array_equal(res["dets_cls"][k], dc)
            assert np.array_equal(res["instance_masks"][k], inst)
            assert res["epe"][k] == e[i].item() and res["aae"][k] == a[i].item()
            np.testing.assert_allclose(res["pred_flow"][k], u.canonical_to_global_flow(flow[i].cpu().numpy(), phi), atol=1e-6)
            k += 1
    assert sum(len(d) for d in res["dets_xy"]) > 0


@pytest.mark.gpu
def test_bench_launcher_stops_the_other_ranks_when_one_dies():
    """A rank that cannot start (here: LOCAL_RANK 1 on a one-GPU box, no shared-GPU hook) must not leave the others
    waiting in the rendezvous: the launcher stops them and fails within seconds, not after a collective timeout."""
    import subprocess, sys, time
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "POF_BENCH_SHARE_GPU")}
    env["POF_BENCH_BACKEND"] = "gloo"
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a one-GPU box")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                        "--repeats", "1", "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr
    assert time.time() - t0 < 120
