"""N4 / A13: the boosted decision-stump baseline.  tests/golden/adaboost.npz holds outputs of the reference's own
BoostedFeatureDetector / nms_predicted_center / Dataset.scan_to_segments + compute_feature (tools/gen_golden.py);
the CPU tests pin the oracle restatement to them, the GPU tests pin the HIP path to both."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "adaboost.npz")
_SCALE = max(1, int(os.environ.get("POF_FUZZ_SCALE", "1")))      # soak runs: more random tables

# reference feature column -> column of the plain 16-column table (R.segment_features / pof_segment_features);
# the reference's median deviation (2), kept-list jumps (3, 4) and mean speed (13) are coupled to its data set
# and live in the 15-column reference table (R.compute_feature_reference / pof_segment_features_ex)
REF_TO_OURS = {0: 0, 1: 1, 5: 4, 6: 5, 7: 6, 8: 7, 9: 8, 10: 9, 11: 10, 12: 11}


def _reference_rows(gold, b, dtype=np.float64):
    """R.compute_feature_reference on fixture scan b, next scan / odometry as gen_golden.py set them up."""
    nb = min(b + 1, 2)
    return R.compute_feature_reference(gold["ft%d_scan" % b].astype(dtype).astype(np.float64), R.laser_phi(),
                                       gold["ft%d_wps" % b], gold["ft%d_scan" % nb].astype(dtype).astype(np.float64),
                                       0.1 * b, 0.1 * nb)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


# ---------------------------------------------------------------------------------------------
# oracle vs the reference's outputs (CPU)
# ---------------------------------------------------------------------------------------------
def test_oracle_simple_classifier_matches_reference(gold):
    for q in range(int(gold["sc_cases"])):
        j, th = R.simple_classifier(gold["sc%d_X" % q], gold["sc%d_Y" % q])
        assert (j, th) == (int(gold["sc%d_out" % q][0]), gold["sc%d_out" % q][1]), q


def test_oracle_adaboost_and_vote_match_reference(gold):
    for q in range(2):
        X, Y = gold["ab%d_X" % q], gold["ab%d_Y" % q]
        K, ns, seed = (int(v) for v in gold["ab%d_cfg" % q])
        np.random.seed(seed)
        alpha, para = R.adaboost(X, Y, K, ns)
        assert np.array_equal(alpha, gold["ab%d_alpha" % q]) and np.array_equal(para, gold["ab%d_para" % q])
        labels, result = R.stump_vote(X, alpha, para)
        assert np.array_equal(result, gold["ab%d_result" % q]) and np.array_equal(labels, gold["ab%d_labels" % q])
    assert gold["ab1_alpha"][np.nonzero(gold["ab1_alpha"])[0][-1]] == 1.0 and gold["ab1_alpha"][-1] == 0.0   # early stop


def _nms_inputs(gold):
    segs = np.split(gold["nms_seg_pts"], np.cumsum(gold["nms_seg_len"])[:-1])
    return segs, gold["nms_preds"], gold["nms_scores"]


def test_oracle_nms_predicted_center_matches_reference(gold):
    segs, preds, scores = _nms_inputs(gold)
    order, p, s = R.nms_segment_centers(segs, preds, scores)
    assert np.array_equal(p, gold["nms_out_preds"]) and np.array_equal(s, gold["nms_out_scores"])
    assert np.array_equal(np.array([segs[i][0] for i in order]), gold["nms_out_first_pt"])
    assert 0 < np.count_nonzero(s > 0) < np.count_nonzero(scores > 0)


def _aligned_features(feat, n_col=0):
    """Rows of segment_features for segments of more than two points (the reference's filter)."""
    return feat[feat[:, n_col] > 2]


def test_oracle_segment_features_match_reference(gold):
    phi = R.laser_phi()
    for b in range(3):
        scan = gold["ft%d_scan" % b]
        cuts, feat = R.segment_features(scan, phi)
        assert np.array_equal(cuts, gold["ft%d_cut_ids" % b])
        ref = gold["ft%d_features" % b]
        ours = _aligned_features(feat)
        assert len(ours) == len(ref) >= 4
        for rc, oc in REF_TO_OURS.items():
            scale = max(1.0, np.abs(ref[:, rc]).max())
            tol = 5e-7 if rc in (6, 7) else 1e-9            # line residual / circle criterion: pinv vs normal equations
            assert np.allclose(ours[:, oc], ref[:, rc], rtol=tol, atol=tol * scale), (b, rc)
        # preceding jump: the reference measures it to the previous KEPT segment; same thing when that is the neighbour
        prev_kept = np.concatenate([[True], (feat[:-1, 0] > 2)])[feat[:, 0] > 2]
        prev_kept[0] = feat[0, 0] > 2 and prev_kept[0]
        rows = np.nonzero(prev_kept)[0][1:]
        assert len(rows) and np.allclose(ours[rows, 2], ref[rows, 3], rtol=1e-12)


def test_oracle_compute_feature_all_columns_match_reference(gold):
    """All 14 feature columns and the label of the reference's compute_feature, incl. the three that depend on its
    bookkeeping (median deviation :127-130, succeeding jump :133-138, mean speed :196-203): identical except
    the line residual (sklearn's lstsq there, pinv here)."""
    for b in range(3):
        ref = gold["ft%d_features" % b]
        ours = _reference_rows(gold, b)
        assert ours.shape == ref.shape
        for c in range(15):
            if c == 6:
                assert np.allclose(ours[:, c], ref[:, c], rtol=0, atol=1e-12 * len(ref)), (b, c)
            else:
                assert np.array_equal(ours[:, c], ref[:, c]), (b, c)
    # fewer than four kept segments: the reference raises at segments[min(idx + 1, 3)]; the restatement marks it
    phi = R.laser_phi()
    scan = np.full(450, 5.0)
    scan[100:140] = 2.0
    rows = R.compute_feature_reference(scan, phi, [], scan, 0.0, 0.0)
    assert len(rows) == 3 and np.isnan(rows[-1, 4]) and not np.isnan(rows[:2, 4]).any()


# ---------------------------------------------------------------------------------------------
# HIP path (GPU)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def detector():
    from planar_optical_flow_amd.src.depracted.model.adaboost_person_det import BoostedFeatureDetector
    return BoostedFeatureDetector()


@pytest.mark.gpu
def test_simple_classifier_matches_reference(gold, detector):
    for q in range(int(gold["sc_cases"])):
        j, th = detector.simple_classifier(gold["sc%d_X" % q], gold["sc%d_Y" % q].reshape(-1, 1))
        assert (j, th) == (int(gold["sc%d_out" % q][0]), gold["sc%d_out" % q][1]), q


@pytest.mark.gpu
def test_adaboost_and_eval_match_reference(gold, detector):
    for q in range(2):
        X, Y = gold["ab%d_X" % q], gold["ab%d_Y" % q]
        K, ns, seed = (int(v) for v in gold["ab%d_cfg" % q])
        np.random.seed(seed)
        alpha, para = detector.adaboost(X, Y.reshape(-1, 1), K, ns)
        assert np.array_equal(alpha, gold["ab%d_alpha" % q]) and np.array_equal(para, gold["ab%d_para" % q])
        labels, result = detector.eval(X, alpha, para)
        assert np.array_equal(result, gold["ab%d_result" % q]) and np.array_equal(labels, gold["ab%d_labels" % q])


@pytest.mark.gpu
def test_stump_search_fuzz_against_oracle(detector):
    """Seeded tables incl. integer-valued features (many equal values across classes), duplicated rows, the
    largest sample count and one-dimensional tables: per-dimension counts and thresholds equal the oracle's."""
    import torch
    rng = np.random.default_rng(99)
    for trial in range(14 * _SCALE):
        n = int(rng.choice([2, 3, 17, 64, 200, 257, 1000, 2048]))
        D = int(rng.integers(1, 20))
        X = rng.normal(size=(n, D)) * 3
        if trial % 3 == 0:
            X = np.round(X)                                   # heavy ties
        if trial % 4 == 1:
            X = X[rng.integers(0, max(n // 3, 1), n)]         # duplicated rows
        Y = np.where(rng.normal(size=n) + X[:, 0] > 0, 1.0, -1.0)
        Y[0], Y[-1] = 1.0, -1.0
        ints, thetas = detector._search(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), None, n)
        ok_dims = 0
        for d in range(D):
            th, err = R.stump_thresholds(X[:, d], Y)
            assert ints[2, d] == len(th), (trial, d)
            if len(th) == 0:
                assert ints[0, d] == ints[1, d] == -1
                continue
            ok_dims += 1
            assert ints[0, d] == err.min() and ints[1, d] == err.max(), (trial, d)
            assert thetas[0, d] == th[np.argmin(err)] and thetas[1, d] == th[np.argmax(err)], (trial, d)
        if ok_dims == D:
            assert detector.simple_classifier(X, Y) == R.simple_classifier(X, Y)


@pytest.mark.gpu
def test_adaboost_fuzz_against_oracle(detector):
    rng = np.random.default_rng(7)
    for trial in range(3 * _SCALE):
        N, D = int(rng.integers(150, 900)), int(rng.integers(2, 13))
        X = rng.normal(size=(N, D))
        Y = np.where(X @ rng.normal(size=D) + 0.7 * rng.normal(size=N) > 0, 1.0, -1.0)
        from planar_optical_flow_amd.src.depracted.model.adaboost_person_det import BoostedFeatureDetector
        got = BoostedFeatureDetector(rng=np.random.default_rng(trial)).adaboost(X, Y, 8, 120)
        want = R.adaboost(X, Y, 8, 120, rng=np.random.default_rng(trial))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        lab, res = detector.eval(X, *got)
        wl, wr = R.stump_vote(X, *want)
        assert np.array_equal(res, wr) and np.array_equal(lab, wl)
        assert np.mean(lab == Y) > 0.6


@pytest.mark.gpu
def test_detector_argument_errors(detector):
    X = np.random.default_rng(0).normal(size=(10, 3))
    Y = np.array([1.0, -1.0] * 5)
    with pytest.raises(ValueError):
        detector.simple_classifier(X, np.ones(10))                   # one class only: no threshold candidates
    with pytest.raises(ValueError):
        detector.simple_classifier(X, Y * 2)                         # labels must be +1 / -1
    bad = X.copy()
    bad[3, 1] = np.nan
    with pytest.raises(ValueError):
        detector.simple_classifier(bad, Y)
    with pytest.raises(ValueError):
        detector.simple_classifier(np.zeros((3000, 2)), np.ones(3000))
    with pytest.raises(IndexError):
        detector.eval(X, np.ones(1), np.array([[4.0, 0.0]]))
    lab, res = detector.eval(X, np.zeros(2), np.zeros((2, 2)))       # unused rounds of a stopped run: j = 0, alpha = 0
    assert np.array_equal(res, np.zeros(10)) and np.array_equal(lab, np.zeros(10))


@pytest.mark.gpu
def test_nms_predicted_center_matches_reference(gold):
    from planar_optical_flow_amd.src.depracted.model.adaboost_person_det import nms_predicted_center
    segs, preds, scores = _nms_inputs(gold)
    sg, p, s = nms_predicted_center([[x, 1.0] for x in segs], preds, scores)
    assert np.array_equal(p, gold["nms_out_preds"]) and np.array_equal(s, gold["nms_out_scores"])
    assert np.array_equal(np.array([x[0] for x in sg]), gold["nms_out_first_pt"])
    assert np.array_equal(scores, gold["nms_scores"])                # caller's array untouched


@pytest.mark.gpu
def test_segment_dataset_matches_reference(gold):
    """SegmentDataset (one pof_segment_features_ex launch, one wave per segment) against the reference's own
    Dataset rows: all 14 columns + label."""
    from planar_optical_flow_amd.src.depracted.model.adaboost_person_det import SegmentDataset, BoostedFeatureDetector
    scans = np.stack([gold["ft%d_scan" % b] for b in range(3)])
    ds = SegmentDataset(scans, [gold["ft%d_wps" % b] for b in range(3)], odom_t=[0.0, 0.1, 0.2])
    for b in range(3):
        ref = gold["ft%d_features" % b]
        assert np.array_equal(ds.labels[b], ref[:, 14])
        assert ds.scans_feature[b].shape == (len(ref), 14) and len(ds.segments[b]) == len(ref)
        assert all(len(sg[0]) == n for sg, n in zip(ds.segments[b], ref[:, 0]))
        for c in range(14):
            scale = max(1.0, np.abs(ref[:, c]).max())
            # the device path takes float32 scans (the fixture's are float64): the line / circle fits of nearly
            # straight segments, the curvature of tiny triangles and the range differences of the speed column
            # amplify that input rounding; the tight pins are oracle vs reference (CPU test above) and kernel vs
            # oracle on identical float32 scans (below)
            tol = 5e-4 if c in (6, 7, 8, 11, 12, 13) else 2e-6
            assert np.allclose(ds.scans_feature[b][:, c], ref[:, c], rtol=tol, atol=tol * scale), (b, c)
        want = _reference_rows(gold, b, np.float32)
        assert np.array_equal(ds.labels[b], want[:, 14])
        scale = np.abs(want[:, :14]).max(axis=0)
        assert np.allclose(ds.scans_feature[b], want[:, :14], rtol=1e-5, atol=1e-5 * scale.max()), b
        for c in (0, 1, 2, 3, 4, 5, 9, 10, 13):               # no fit involved: float64 round-off only
            assert np.allclose(ds.scans_feature[b][:, c], want[:, c], rtol=1e-9, atol=1e-12), (b, c)
    assert ds.input.shape == (len(ds), 14) and set(np.unique(ds.target)) == {-1.0, 1.0}
    alpha, para = BoostedFeatureDetector(rng=np.random.default_rng(0)).adaboost(ds.input, ds.target, 6, 64)
    assert para[0, 0] >= 1 and np.isfinite(alpha).all()
    with pytest.raises(IndexError):                           # fewer than four kept segments: as the reference
        flat = np.full((1, 450), 5.0, dtype=np.float32)
        flat[0, 100:140] = 2.0
        SegmentDataset(flat, [[]])


@pytest.mark.gpu
def test_segment_reference_rows_fuzz():
    """pof_segment_features_ex vs the oracle's compute_feature restatement on seeded synthetic scans: long wall
    segments (hundreds of points per wave), legs, ties in the median ranks (repeated ranges), segments at both ends,
    the 3600-point geometry; plain 16-column table from the same launch equals pof_segment_features."""
    import torch
    from planar_optical_flow_amd import ops, synth
    for seed, N, inc in ((1, 450, 0.5), (2, 450, 0.5), (3, 3600, 0.1), (4, 225, 1.0)):
        sb = synth.make_batch(seed=500 + seed, B=5, T=1, N=N, angle_inc=np.radians(inc), dropout=0.0)
        scans = sb.scans[:, 0].copy()
        scans[1, 40:90] = np.float32(3.25)                    # equal ranges: ties for the median selection
        scans[2, :3] = np.float32(1.0)                        # a kept segment at index 0
        phi = R.laser_phi(np.radians(inc), N)
        rng = np.random.default_rng(seed)
        wps = []
        for b in range(5):
            xy = np.array(R.polar_to_xy(scans[b], phi)).T
            wps.append(xy[rng.integers(0, N, 4)] + 0.05)
        offs = np.zeros(6, dtype=np.int32)
        offs[1:] = np.cumsum([len(w) for w in wps])
        nxt = np.roll(scans, -1, axis=0) + np.float32(0.01)
        dt = rng.uniform(0.02, 0.2, 5)
        tab = ops.phi_table(np.radians(inc), N)
        dev = "cuda"
        sid, num, kept, ref, plain = ops.segment_features_reference(
            torch.from_numpy(scans).to(dev), tab, torch.from_numpy(nxt).to(dev), torch.from_numpy(dt).to(dev),
            torch.from_numpy(offs).to(dev), torch.from_numpy(np.concatenate(wps)).to(dev), want_plain=True)
        sid2, num2, plain2 = ops.segment_features(torch.from_numpy(scans).to(dev), tab)
        assert torch.equal(sid, sid2) and torch.equal(num, num2)
        assert torch.equal(torch.nan_to_num(plain, nan=-7.0), torch.nan_to_num(plain2, nan=-7.0))
        for b in range(5):
            want = R.compute_feature_reference(scans[b].astype(np.float64), phi, wps[b], nxt[b].astype(np.float64),
                                               0.0, dt[b])
            K = int(kept[b].item())
            assert K == len(want) > 0
            got = ref[b, :K].cpu().numpy()
            assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 14], want[:, 14])
            assert np.array_equal(np.isnan(got[:, 4]), np.isnan(want[:, 4]))
            for c in (1, 2, 3, 4, 5, 9, 10, 13):
                np.testing.assert_allclose(got[:, c], want[:, c], rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=str((seed, b, c)))
            good = (want[:, 0] >= 8) & (want[:, 8] < 50)      # well conditioned fits (as test_hip_parity A13)
            np.testing.assert_allclose(got[good][:, [6, 7, 8]], want[good][:, [6, 7, 8]], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(got[good][:, [11, 12]], want[good][:, [11, 12]], rtol=1e-6, atol=1e-8)
