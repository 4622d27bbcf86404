"""Seeded random-shape sweeps of the HIP entry points against the oracle: geometries, strides, window
parameters, batch sizes and detection counts that the hand-picked cases do not cover.  Every case is
small (the oracle is per-sample NumPy); the bar is the same as in test_hip_parity.py."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_numpy as R  # noqa: E402
from planar_optical_flow_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu

# POF_FUZZ_SCALE=k multiplies the number of seeds of every sweep (soak runs; the default suite uses 1)
_SCALE = max(1, int(os.environ.get("POF_FUZZ_SCALE", "1")))


@pytest.fixture(scope="module")
def ops():
    from planar_optical_flow_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_fuzz_cutout_bit_exact(ops, seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([33, 64, 90, 180, 255, 450, 720]))
    T = int(rng.integers(1, 7))
    P = int(rng.choice([8, 12, 17, 32, 48, 56, 60]))
    inc = float(rng.choice([0.25, 0.5, 1.0, 2.0]))
    kw = dict(stride=int(rng.choice([1, 1, 2, 3])), centered=bool(rng.integers(0, 2)), fixed=bool(rng.integers(0, 2)),
              window_width=float(rng.choice([0.5, 1.0, 1.66, 3.0])), window_depth=float(rng.choice([0.3, 0.5, 1.0, 0.7])),
              num_cutout_pts=P, padding_val=float(rng.choice([29.99, 0.0, 12.5])), area_mode=bool(rng.integers(0, 2)))
    B = int(rng.integers(1, 4))
    sb = synth.make_batch(seed=2000 + seed, B=B, T=T, N=N, angle_inc=np.radians(inc))
    scans = sb.scans.copy()
    scans[rng.random(scans.shape) < 0.02] = 0.02          # near-field returns: very wide windows
    phi = R.laser_phi(np.radians(inc), N)
    tab = ops.phi_table(np.radians(inc), N)
    got, dbg = ops.cutout(dev(scans), tab, return_debug=True, **kw)
    for b in range(B):
        want, wd = R.cutout(scans[b], phi, atan_mode="cr", return_debug=True, **kw)
        assert np.array_equal(dbg["lo"][b].cpu().numpy(), wd["lo"]), (seed, kw)
        assert np.array_equal(got[b].cpu().numpy(), want), (seed, kw)
    fast = ops.cutout(dev(scans), tab, exact_values=False, **kw)
    assert (fast - got).abs().max().item() <= 2e-5 * max(1.0, 1.0 / kw["window_depth"]) * (30.0 if not kw["centered"] else 1.0)


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_fuzz_scan_preprocess(ops, seed):
    rng = np.random.default_rng(3000 + seed)
    N = int(rng.choice([31, 90, 225, 450, 451, 900]))
    B = int(rng.integers(1, 40))
    inc = float(rng.choice([0.5, 1.0, 0.25]))
    sb = synth.make_batch(seed=4000 + seed, B=B, T=2, N=N, angle_inc=np.radians(inc),
                          max_legs=int(rng.choice([0, 3, 6, 14])), mixed_classes=bool(rng.integers(0, 2)))
    ped = bool(rng.integers(0, 2))
    o, r, c = sb.det_csr(pedestrian_only=ped)
    tab = ops.phi_table(np.radians(inc), N)
    phi = R.laser_phi(np.radians(inc), N)
    canonical = bool(rng.integers(0, 2))
    labels = (1, 1, 1) if ped else (1, 2, 3)
    out = ops.scan_preprocess(dev(sb.scans), tab, dev(sb.odom0), dev(sb.odom1), ops.DetCSR.from_numpy(o, r, c, "cuda"),
                              canonical=canonical, out_dtype=torch.float64, labels=labels,
                              want=("xy", "flow", "closest", "target_cls", "target_reg", "dyn_mask", "exclude_mask"))
    for b in range(B):
        cur = sb.scans[b, -1]
        wc, wa, wp = sb.dets[b]["wc"], sb.dets[b]["wa"], sb.dets[b]["wp"]
        if ped:
            wc, wa = [], []
        xy = np.array(R.polar_to_xy(cur, phi)).T
        flow = R.displacement_from_odometry(xy, sb.odom0[b], sb.odom1[b])
        if canonical:
            flow = R.flow_to_canonical(flow, phi)
        np.testing.assert_allclose(out["flow"][b].cpu().numpy(), flow, rtol=0, atol=1e-12)
        cls, reg = R.regression_target(cur, phi, wc, wa, wp, pedestrian_only=ped)
        assert np.array_equal(out["target_cls"][b].cpu().numpy(), cls), (seed, b)
        np.testing.assert_allclose(out["target_reg"][b].cpu().numpy(), reg, rtol=0, atol=1e-6)
        dyn = R.dynamic_mask(xy, wc, wa, wp)
        assert np.array_equal(out["dyn_mask"][b].cpu().numpy().astype(np.float64), dyn), (seed, b)


@pytest.mark.parametrize("seed", range(14 * _SCALE))
def test_fuzz_scan_preprocess_float32_flat_form(ops, seed):
    """The flat-axis kernel (N >= 256, even) with FLOAT32 outputs -- the instantiation bench.py times: chunk
    boundaries that fall inside samples (N not a multiple of 256), partial last chunks (odd batch sizes), crowded
    samples (more than 8 detections: CSR path), every flow kind, chained launches whose next batch has a different
    size.  Association, labels and masks are bit-exact; target_reg <= 1e-6; flow EPE <= 1e-5 m (float32 arithmetic;
    kind 1 keeps float64); xy within float32 rounding."""
    rng = np.random.default_rng(9000 + seed)
    N = int(rng.choice([256, 258, 450, 512, 900, 3600]))
    B = int(rng.integers(1, 60))
    inc = float(rng.choice([0.5, 0.25, 0.1]))
    sb = synth.make_batch(seed=9100 + seed, B=B, T=2, N=N, angle_inc=np.radians(inc),
                          max_legs=int(rng.choice([0, 3, 6, 14])), mixed_classes=bool(rng.integers(0, 2)))
    o, r, c = sb.det_csr()
    det = ops.DetCSR.from_numpy(o, r, c, "cuda")
    tab = ops.phi_table(np.radians(inc), N)
    phi = R.laser_phi(np.radians(inc), N)
    kind = int(rng.choice([ops.FLOW_DISPLACEMENT, ops.FLOW_DISPLACEMENT, ops.FLOW_TARGET, ops.FLOW_VELOCITY]))
    canonical = bool(rng.integers(0, 2))
    want = ("xy", "flow", "closest", "target_cls", "target_reg", "dyn_mask", "valid_mask", "exclude_mask")
    scans, o0, o1 = dev(sb.scans), dev(sb.odom0), dev(sb.odom1)
    if seed % 2:
        # chained form: params of this batch from a priming launch, params of a (different-size) next batch ride along
        nb = synth.make_batch(seed=9200 + seed, B=int(rng.integers(1, 30)), T=2, N=N, angle_inc=np.radians(inc))
        no, nr, nc = nb.det_csr()
        ndet = ops.DetCSR.from_numpy(no, nr, nc, "cuda")
        ws = torch.empty(ops.scan_preprocess_workspace_bytes(B, det.rphi.shape[0]), dtype=torch.uint8, device="cuda")
        nws = torch.empty(ops.scan_preprocess_workspace_bytes(len(nb.scans), ndet.rphi.shape[0]), dtype=torch.uint8, device="cuda")
        ops.scan_preprocess(scans, tab, o0, o1, det, flow_kind=kind, canonical=canonical, want=want, workspace=ws, phases=1)
        out = ops.scan_preprocess(scans, tab, o0, o1, det, flow_kind=kind, canonical=canonical, want=want, workspace=ws,
                                  next_batch={"odom0": dev(nb.odom0), "odom1": dev(nb.odom1), "dets": ndet, "workspace": nws})
        # the params the launch left for the next batch give the same result as a self-contained call
        a_ = ops.scan_preprocess(dev(nb.scans), tab, dev(nb.odom0), dev(nb.odom1), ndet, flow_kind=kind, canonical=canonical,
                                 want=("flow", "target_cls"), workspace=nws, phases=2)
        b_ = ops.scan_preprocess(dev(nb.scans), tab, dev(nb.odom0), dev(nb.odom1), ndet, flow_kind=kind, canonical=canonical,
                                 want=("flow", "target_cls"))
        assert torch.equal(a_["flow"], b_["flow"]) and torch.equal(a_["target_cls"], b_["target_cls"])
    else:
        out = ops.scan_preprocess(scans, tab, o0, o1, det, flow_kind=kind, canonical=canonical, want=want)
    assert out["flow"].dtype == torch.float32 and out["xy"].dtype == torch.float32
    ref_flow = {ops.FLOW_DISPLACEMENT: R.displacement_from_odometry, ops.FLOW_VELOCITY: R.velocity_from_odometry}
    epes = []
    for b in range(B):
        cur = sb.scans[b, -1]
        d = sb.dets[b]
        xy = np.array(R.polar_to_xy(cur, phi)).T
        if kind == ops.FLOW_TARGET:
            flow = R.flow_target(cur, phi, sb.odom0[b], sb.odom1[b])
        else:
            flow = ref_flow[kind](xy, sb.odom0[b], sb.odom1[b])
        if canonical:
            flow = R.flow_to_canonical(flow, phi)
        epes.append(np.linalg.norm(out["flow"][b].cpu().numpy().astype(np.float64) - flow, axis=-1).mean())
        np.testing.assert_allclose(out["xy"][b].cpu().numpy(), xy, rtol=2e-7, atol=1e-7)
        cls, reg = R.regression_target(cur, phi, d["wc"], d["wa"], d["wp"])
        assert np.array_equal(out["target_cls"][b].cpu().numpy(), cls), (seed, b)
        radii = [0.6] * len(d["wc"]) + [0.4] * len(d["wa"]) + [0.35] * len(d["wp"])
        dets = list(d["wc"]) + list(d["wa"]) + list(d["wp"])
        assert np.array_equal(out["closest"][b].cpu().numpy(), np.asarray(R.closest_detection(cur, phi, dets, radii))), (seed, b)
        np.testing.assert_allclose(out["target_reg"][b].cpu().numpy(), reg, rtol=0, atol=1e-6)
        dyn, val = R.dynamic_mask(xy, d["wc"], d["wa"], d["wp"]), R.valid_point_mask(cur)
        assert np.array_equal(out["dyn_mask"][b].cpu().numpy().astype(np.float64), dyn), (seed, b)
        assert np.array_equal(out["valid_mask"][b].cpu().numpy(), val)
        assert np.array_equal(out["exclude_mask"][b].cpu().numpy().astype(np.float64), dyn * val)
    assert max(epes) < 1e-5, (seed, kind, max(epes))


@pytest.mark.parametrize("seed", range(12 * _SCALE))
def test_fuzz_attention_and_correlation(ops, seed):
    rng = np.random.default_rng(5000 + seed)
    B, N = int(rng.integers(1, 4)), int(rng.choice([3, 17, 40, 64, 100, 257]))
    E, F, w = int(rng.choice([4, 12, 32, 128])), int(rng.choice([4, 16, 56, 260])), int(rng.choice([1, 3, 7, 11, 15]))
    ex = rng.normal(0, 0.4, (B, N, E)).astype(np.float32)
    et = rng.normal(0, 0.4, (B, N, E)).astype(np.float32)
    x = rng.normal(0, 1, (B, N, F)).astype(np.float32)
    t = rng.normal(0, 1, (B, N, F)).astype(np.float32)
    alpha = float(rng.uniform(0.1, 0.9))
    out, band, prob = ops.spatial_attention(dev(ex), dev(et), dev(x), dev(t), alpha, w)
    wo, wb = R.spatial_attention(ex.astype(np.float64), et.astype(np.float64), x.astype(np.float64),
                                 t.astype(np.float64), alpha, w)
    np.testing.assert_allclose(band.cpu().numpy(), wb, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), wo, rtol=1e-4, atol=1e-5)
    C, n = int(rng.choice([1, 5, 32, 70])), int(rng.choice([2, 9, 57, 64, 65, 200]))
    K, md = int(rng.choice([1, 3, 5])), int(rng.integers(0, 8))
    f1 = rng.integers(-3, 4, (B, C, n)).astype(np.float32)
    f2 = rng.integers(-3, 4, (B, C, n)).astype(np.float32)
    got = ops.band_correlation(dev(f1), dev(f2), K, md).cpu().numpy()
    assert np.array_equal(got, R.band_correlation(f1.astype(np.float64), f2.astype(np.float64), K, md).astype(np.float32))


def test_pedestrian_only_batch_keeps_all_classes_in_the_mask(ops):
    """DROWBatchPreprocessor(pedestrian_only=True) like the reference's __getitem__: regression target from
    the persons only (label 1), dynamic mask from wheelchairs, walkers and persons."""
    from planar_optical_flow_amd.preprocess import DROWBatchPreprocessor
    sb = synth.make_batch(seed=77, B=24, T=3, mixed_classes=True)
    pre = DROWBatchPreprocessor(cutout_kwargs=None, pedestrian_only=True)
    dets = pre.make_detections([d["wc"] for d in sb.dets], [d["wa"] for d in sb.dets], [d["wp"] for d in sb.dets])
    batch = pre(dev(sb.scans), dev(sb.odom0), dev(sb.odom1), dets)
    phi = R.laser_phi()
    n_other = 0
    for b in range(24):
        cur, d = sb.scans[b, -1], sb.dets[b]
        n_other += len(d["wc"]) + len(d["wa"])
        cls, reg = R.regression_target(cur, phi, d["wc"], d["wa"], d["wp"], pedestrian_only=True)
        assert np.array_equal(batch["target_cls"][b].cpu().numpy(), cls)
        np.testing.assert_allclose(batch["target_reg"][b].cpu().numpy(), reg, rtol=0, atol=1e-6)
        xy = np.array(R.polar_to_xy(cur, phi)).T
        want = R.dynamic_mask(xy, d["wc"], d["wa"], d["wp"]) * R.valid_point_mask(cur)
        assert np.array_equal(batch["exclude_mask"][b].cpu().numpy().astype(np.float64), want)
    assert n_other > 0


def test_nms_distances_at_the_threshold(ops):
    """Pairs of centres whose distance is min_dist up to a few float32 steps of one range -- inside the band the float32
    screen of the wave kernel cannot decide -- and far / huge coordinates next to them: kept set and instance ids
    equal the oracle's float64 decisions."""
    rng = np.random.default_rng(99)
    N, md = 450, 0.5
    phi = R.laser_phi()
    tab = ops.phi_table()
    B = 64
    scans = np.full((B, N), 25.0, np.float32)
    reg = np.zeros((B, N, 2))
    cls = np.tile(np.linspace(0.4, 0.1, N), (B, 1))          # distinct, low everywhere else
    for b in range(B):
        i = int(rng.integers(5, N - 40))
        j = i + int(rng.integers(1, 30))
        Rr = float(rng.uniform(1.0, 20.0))
        dl = phi[j] - phi[i]
        disc = md * md - (Rr * np.sin(dl)) ** 2
        if disc <= 0:
            continue
        rj = Rr * np.cos(dl) + np.sqrt(disc)                 # |c_i - c_j| = min_dist
        scans[b, i] = np.float32(Rr)
        scans[b, j] = np.nextafter(np.float32(rj), np.float32(np.inf if b % 2 else 0.0))
        for _ in range(b % 5):
            scans[b, j] = np.nextafter(scans[b, j], np.float32(np.inf if b % 2 else 0.0))
        cls[b, i], cls[b, j] = 0.9, 0.8
        if b % 7 == 0:
            scans[b, 3] = np.float32(3.0e6)                  # a huge coordinate widens the screen's band
    xy, dc, num, inst = ops.nms_predicted_center(dev(scans), tab, dev(cls), dev(reg), md)
    for b in range(B):
        wxy, wcls, winst = R.nms_predicted_center(scans[b], phi, cls[b][:, None], reg[b], md)
        m = int(num[b].item())
        assert m == len(wxy), b
        assert np.array_equal(inst[b].cpu().numpy(), winst), b
        np.testing.assert_allclose(xy[b, :m].cpu().numpy(), wxy, rtol=0, atol=1e-9 * max(1.0, float(np.abs(wxy).max())))


@pytest.mark.parametrize("seed", range(10 * _SCALE))
def test_fuzz_nms_and_polar_grid(ops, seed):
    rng = np.random.default_rng(6000 + seed)
    N = int(rng.choice([17, 64, 100, 450, 451, 700]))
    inc = float(rng.choice([0.5, 1.0]))
    B = int(rng.integers(1, 4))
    sb = synth.make_batch(seed=7000 + seed, B=B, T=1, N=N, angle_inc=np.radians(inc))
    scans = sb.scans[:, 0]
    phi = R.laser_phi(np.radians(inc), N)
    tab = ops.phi_table(np.radians(inc), N)
    cls = rng.permutation(B * N).reshape(B, N).astype(np.float64) / (B * N)      # distinct scores
    reg = rng.normal(0, 0.3, (B, N, 2))
    md = float(rng.choice([0.2, 0.5, 1.5]))
    xy, dc, num, inst = ops.nms_predicted_center(dev(scans), tab, dev(cls), dev(reg), md)
    for b in range(B):
        wxy, wcls, winst = R.nms_predicted_center(scans[b], phi, cls[b][:, None], reg[b], md)
        m = int(num[b].item())
        assert m == len(wxy), (seed, b)
        assert np.array_equal(inst[b].cpu().numpy(), winst)
        np.testing.assert_allclose(xy[b, :m].cpu().numpy(), wxy, rtol=0, atol=1e-12)
        assert np.array_equal(dc[b, :m].cpu().numpy(), wcls[:, 0])
    # tied scores (a float32 sigmoid saturates to exactly 1.0 for confident points): the kernel's order is total,
    # equal scores by descending point index
    tied = np.round(cls * 7) / 7
    tied[:, ::3] = 1.0
    xy, dc, num, inst = ops.nms_predicted_center(dev(scans), tab, dev(tied), dev(reg), md)
    for b in range(B):
        wxy, wcls, winst = R.nms_predicted_center(scans[b], phi, tied[b][:, None], reg[b], md, stable_ties=True)
        m = int(num[b].item())
        assert m == len(wxy) and np.array_equal(inst[b].cpu().numpy(), winst), (seed, b)
        np.testing.assert_allclose(xy[b, :m].cpu().numpy(), wxy, rtol=0, atol=1e-12)
        assert np.array_equal(dc[b, :m].cpu().numpy(), wcls[:, 0])
    kw = dict(min_range=float(rng.choice([0.0, 0.5])), max_range=float(rng.choice([20.0, 29.5, 30.0])),
              range_bin_size=float(rng.choice([0.25, 0.5, 1.0])), tsdf_clip=float(rng.choice([0.0, 1.0, 2.5])),
              normalize=bool(rng.integers(0, 2)))
    T_ = int(rng.integers(1, 5))
    sbp = synth.make_batch(seed=8000 + seed, B=B, T=T_, N=N, angle_inc=np.radians(inc))
    got = ops.polar_grid(dev(sbp.scans), **kw).cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], R.polar_grid(sbp.scans[b], **kw)), (seed, kw)


@pytest.mark.parametrize("seed", range(8 * _SCALE))
def test_fuzz_segments_iou_conv(ops, seed):
    rng = np.random.default_rng(9000 + seed)
    # A13: cut indices bit-exact, simple features tight, fits on well conditioned segments
    N = int(rng.choice([120, 450, 451]))
    inc = float(rng.choice([0.5, 1.0]))
    jump = float(rng.choice([0.3, 0.5, 0.8]))
    sb = synth.make_batch(seed=9500 + seed, B=3, T=1, N=N, angle_inc=np.radians(inc), dropout=0.0)
    scans = sb.scans[:, 0]
    phi = R.laser_phi(np.radians(inc), N)
    sid, num, feat = ops.segment_features(dev(scans), ops.phi_table(np.radians(inc), N), jump)
    for b in range(3):
        cuts, want = R.segment_features(scans[b], phi, jump)
        S = len(cuts) + 1
        assert int(num[b].item()) == S
        ids = np.zeros(N, dtype=np.int32)
        ids[cuts] = 1
        assert np.array_equal(sid[b].cpu().numpy(), np.cumsum(ids))
        got = feat[b, :S].cpu().numpy()
        assert np.array_equal(got[:, 0], want[:, 0])
        np.testing.assert_allclose(got[:, [1, 2, 3, 4, 8, 9]], want[:, [1, 2, 3, 4, 8, 9]], rtol=1e-9, atol=1e-12,
                                   equal_nan=True)
    # A16: random boxes, the reference's float32 tolerance
    def boxes(n, s):
        b = np.zeros((n, s), dtype=np.float32)
        b[:, :2] = rng.uniform(-1.5, 1.5, (n, 2))
        if s == 5:
            b[:, 2:4] = rng.uniform(0.2, 2.0, (n, 2))
            b[:, 4] = rng.uniform(-np.pi, np.pi, n)
        else:
            b[:, 2] = rng.uniform(-0.5, 0.5, n)
            b[:, 3:6] = rng.uniform(0.2, 2.0, (n, 3))
            b[:, 6] = rng.uniform(-np.pi, np.pi, n)
        return b
    for s, is3d in ((5, False), (7, True)):
        bx, qx = boxes(int(rng.integers(1, 40)), s), boxes(int(rng.integers(1, 40)), s)
        crit = int(rng.choice([-1, 0, 1]))
        got = ops.rotate_iou(dev(bx), dev(qx), is_3d=is3d, criterion=crit).cpu().numpy()
        np.testing.assert_allclose(got, R.rotate_iou(bx, qx, criterion=crit, is_3d=is3d), rtol=0, atol=2e-5)
    # N2 layer: random channel / length / sequence counts, integer data exact
    S, Ci, Co, L = int(rng.integers(1, 20)), int(rng.integers(1, 70)), int(rng.integers(1, 140)), int(rng.integers(1, 30))
    pool = bool(rng.integers(0, 2)) and L % 2 == 0 and L >= 2
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randint(-3, 4, (S, Ci, L), generator=g).float().cuda()
    w = torch.randint(-2, 3, (Co, Ci, 3), generator=g).float().cuda()
    scale = torch.full((Co,), 0.5).cuda()
    shift = torch.randint(-3, 4, (Co,), generator=g).float().cuda()
    got = ops.conv3_bn_lrelu(x, w.permute(2, 1, 0).contiguous(), scale, shift, pool=pool, negative_slope=0.25)
    y = torch.nn.functional.conv1d(x.double(), w.double(), None, padding=1) * 0.5 + shift.double()[None, :, None]
    y = torch.nn.functional.leaky_relu(y, 0.25)
    want = (torch.max_pool1d(y, 2) if pool else y).float()
    assert torch.equal(got, want), (S, Ci, Co, L, pool)


@pytest.mark.parametrize("seed", range(8 * _SCALE))
def test_fuzz_standalone_geometry_and_store(ops, seed):
    """A3 / A5 / A12 stand-alone entry points and the N1 store kernels on random shapes."""
    rng = np.random.default_rng(11000 + seed)
    B, N = int(rng.integers(1, 6)), int(rng.choice([7, 64, 450, 451]))
    inc = float(rng.choice([0.5, 1.0]))
    phi = R.laser_phi(np.radians(inc), N)
    tab = ops.phi_table(np.radians(inc), N)
    xy = rng.uniform(-20, 20, (B, N, 2))
    o0 = np.column_stack([rng.uniform(-3, 3, (B, 2)), rng.uniform(-3, 3, B)])
    o1 = o0 + np.column_stack([rng.uniform(-0.1, 0.1, (B, 2)), rng.uniform(-0.05, 0.05, B)])
    for kind, fn in ((ops.FLOW_DISPLACEMENT, R.displacement_from_odometry), (ops.FLOW_VELOCITY, R.velocity_from_odometry)):
        for canonical in (False, True):
            got = ops.flow_from_xy(dev(xy), dev(o0), dev(o1), flow_kind=kind, canonical=canonical, tab=tab).cpu().numpy()
            for b in range(B):
                want = fn(xy[b], o0[b], o1[b])
                if canonical:
                    want = R.flow_to_canonical(want, phi)
                np.testing.assert_allclose(got[b], want, rtol=0, atol=1e-11)
    # A5 both ways
    r = rng.uniform(0.5, 20, (B, N)).astype(np.float32)
    dr, dp = rng.uniform(0.5, 20, (B, N)), rng.uniform(-1.9, 1.9, (B, N))
    dx, dy = ops.det_to_canonical(dev(r), tab, dev(dr), dev(dp))
    wx, wy = R.det_to_canonical(r.astype(np.float64), phi[None], dr, dp)
    np.testing.assert_allclose(dx.cpu().numpy(), wx, rtol=0, atol=1e-12)
    np.testing.assert_allclose(dy.cpu().numpy(), wy, rtol=0, atol=1e-12)
    rr, pp = ops.canonical_to_det(dev(r), tab, dx, dy)
    wr, wp = R.canonical_to_det(r.astype(np.float64), phi[None], wx, wy)
    # r = (r + dy) / cos(atan2(dx, r + dy)) is ill-conditioned where the detection is at right angles to the ray:
    # a last-bit difference of atan2 / cos is amplified by 1 / |cos|, and so is the bound (found by the 64x soak)
    cond = 1.0 / np.maximum(np.abs(np.cos(np.arctan2(wx, r.astype(np.float64) + wy))), 1e-9)
    assert np.all(np.abs(rr.cpu().numpy() - wr) <= 1e-12 * (1.0 + np.abs(wr)) * np.maximum(cond, 1.0))
    np.testing.assert_allclose(pp.cpu().numpy(), wp, rtol=0, atol=1e-12)
    # A12
    pred = rng.normal(0, 1, (B, N, 2)).astype(np.float32)
    tgt = rng.normal(0, 1, (B, N, 2)).astype(np.float32)
    mask = (rng.random((B, N)) < 0.7).astype(np.float32)
    e, a, c = ops.flow_errors(dev(pred), dev(tgt))
    we, wa = R.epe_aae_eval(pred.astype(np.float64), tgt.astype(np.float64))
    np.testing.assert_allclose((e / c).cpu().numpy(), we, rtol=1e-5)
    np.testing.assert_allclose((a / c).cpu().numpy() * 180 / np.pi, wa, rtol=1e-4)
    e, a, c = ops.flow_errors(dev(pred), dev(tgt), dev(mask))
    if mask.sum() > 0:
        np.testing.assert_allclose((e.sum() / c.sum()).item(), R.epe_masked(pred.astype(np.float64), tgt.astype(np.float64), mask), rtol=1e-5)
    # N1: window gather and time association on a random store
    S = int(rng.integers(12, 60))
    scans_all = rng.uniform(0.3, 25, (S, N)).astype(np.float32)
    t_s = np.sort(rng.uniform(0, 10, S)).astype(np.float32)
    O = int(rng.integers(5, 90))
    t_o = np.sort(rng.uniform(0, 10, O)).astype(np.float32)
    odoms = rng.uniform(-5, 5, (O, 3)).astype(np.float32)
    ns, dist, stride = int(rng.integers(1, 7)), int(rng.integers(0, 7)), int(rng.integers(1, 4))
    idx = rng.integers(0, S, 9).astype(np.int32)
    win, rc, rp = ops.gather_windows(dev(scans_all), dev(np.zeros(9, np.int32)), dev(idx), ns, dist, stride)
    od0, od1, i0, i1 = ops.associate_odometry(dev(t_s), dev(t_o), dev(odoms), dev(np.zeros(9, np.int32)),
                                              dev(np.full(9, O, np.int32)), rc, rp)
    for k in range(9):
        inds = R.window_indices(int(idx[k]), ns, dist, stride)
        assert np.array_equal(win[k].cpu().numpy(), np.vstack((scans_all[inds], scans_all[idx[k]])))
        w0, w1 = R.associate_odometry(t_o, t_s, int(idx[k]), inds)
        assert (int(i0[k]), int(i1[k])) == (w0, w1)
        assert np.array_equal(od1[k].cpu().numpy(), odoms[w1].astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12 * _SCALE))
def test_fuzz_training_trunk_kernels(ops, seed):
    """N2 training kernels on random shapes (ragged channel counts, odd lengths, sequence counts that leave the
    last chunk / stage / split partly empty): fused BatchNorm(train) tail forward + backward against torch in
    float64, split-K weight gradient and the conv forward / data gradient against float64 conv1d."""
    import torch
    rng = np.random.default_rng(9000 + seed)
    g = torch.Generator(device="cuda").manual_seed(9000 + seed)
    # ---- tail
    L = int(rng.choice([2, 4, 6, 7, 8, 12, 14, 24, 28, 30, 48, 56, 64, 100]))
    C = int(rng.integers(1, 40)) * 4 if L % 4 else int(rng.integers(1, 160))
    if (C * L) % 4:
        C *= 4
    S = int(rng.integers(1, 400))
    pool = bool(rng.integers(0, 2)) and L % 2 == 0
    slope = float(rng.choice([0.1, 0.01, 0.3]))
    y = (torch.randn(S, C, L, device="cuda", generator=g) * float(rng.uniform(0.5, 3)) + float(rng.uniform(-2, 2)))
    gam = torch.rand(C, device="cuda", generator=g) + 0.5
    bet = torch.rand(C, device="cuda", generator=g) - 0.5
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    z, mu, istd = ops.bn_lrelu_pool_forward(y, gam, bet, rm, rv, momentum=0.1, eps=1e-5, negative_slope=slope, pool=pool)
    y64 = y.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm64, rv64 = torch.zeros(C, device="cuda", dtype=torch.float64), torch.ones(C, device="cuda", dtype=torch.float64)
    u = torch.nn.functional.batch_norm(y64, rm64, rv64, g64, b64, True, 0.1, 1e-5)
    z64 = torch.nn.functional.leaky_relu(u, slope)
    if pool:
        z64 = torch.max_pool1d(z64, 2)
    assert torch.allclose(z.double(), z64, rtol=1e-5, atol=3e-5)
    if S * L > 1:
        assert torch.allclose(rv.double(), rv64, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rm.double(), rm64, rtol=1e-5, atol=1e-6)
    dz = torch.randn(z.shape, device="cuda", generator=g)
    z64.backward(dz.double())
    dy, dgam, dbet, dsum = ops.bn_lrelu_pool_backward(y, dz, gam, bet, mu, istd, negative_slope=slope, pool=pool,
                                                      bias_grad=True)
    # near a zero of u or a pooled tie the float32 forward may pick the other branch than float64 does: compare
    # where the float64 pre-activation is clear of both
    scale = max(float(y64.grad.abs().max()), 1e-3)
    bad = (dy.double() - y64.grad).abs() > 1e-4 * scale
    assert float(bad.double().mean()) < 2e-4
    # every element whose branch differs moves a per-channel sum by up to |dz| * |xhat| (gamma) or |dz| (beta)
    flips = float(bad.sum())
    xh_max = float(((y64 - y64.mean(dim=(0, 2), keepdim=True)) / y64.std(dim=(0, 2), keepdim=True)).abs().max())
    dz_max = float(dz.abs().max())
    assert float((dgam.double() - g64.grad).abs().max()) <= 2e-3 * max(float(g64.grad.abs().max()), 1.0) \
        + 2.0 * flips * dz_max * xh_max
    assert float((dbet.double() - b64.grad).abs().max()) <= 2e-3 * max(float(b64.grad.abs().max()), 1.0) \
        + flips * dz_max
    assert float((dsum.double() - dy.double().sum(dim=(0, 2))).abs().max()) <= 1e-4 * max(float(dy.abs().sum(dim=(0, 2)).max()), 1.0)
    # ---- convolution passes
    Lc = int(rng.choice([3, 6, 7, 9, 12, 14, 17, 24, 28, 33, 48, 56, 64]))
    Ci, Co = int(rng.integers(1, 200)), int(rng.integers(1, 200))
    Sc = int(rng.integers(1, 120))
    x = torch.randn(Sc, Ci, Lc, device="cuda", generator=g)
    gy = torch.randn(Sc, Co, Lc, device="cuda", generator=g)
    w = (torch.randn(Co, Ci, 3, device="cuda", generator=g) * 0.2)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    yc64 = torch.nn.functional.conv1d(x64, w64, padding=1)
    yc64.backward(gy.double())
    from planar_optical_flow_amd import torch_ops
    assert ops.conv3_wgrad_supported(Sc, Ci, Co, Lc) == (Lc % 2 == 0 or Lc <= 32)   # long odd rows: the library's
    dw = torch_ops._weight_grad(x, gy, w)
    assert float((dw.double() - w64.grad).abs().max()) <= 3e-5 * max(float(w64.grad.abs().max()), 1.0)
    one, zero = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    yc = ops.conv3_bn_lrelu(x, w.permute(2, 1, 0).contiguous(), one, zero, pool=False, negative_slope=1.0)
    assert float((yc.double() - yc64.detach()).abs().max()) <= 3e-5 * max(float(yc64.detach().abs().max()), 1.0)
    dx = ops.conv3_bn_lrelu(gy, w.flip(2).permute(2, 0, 1).contiguous(), torch.ones(Ci, device="cuda"),
                            torch.zeros(Ci, device="cuda"), pool=False, negative_slope=1.0)
    assert float((dx.double() - x64.grad).abs().max()) <= 3e-5 * max(float(x64.grad.abs().max()), 1.0)
