"""N4 / A13: the boosted decision-stump baseline.  tests/golden/adaboost.npz holds outputs of the reference's own
BoostedFeatureDetector / nms_predicted_center / Dataset.scan_to_segments + compute_feature (tools/gen_golden.py);
the CPU tests pin the oracle restatement to them, the GPU tests pin the HIP path to both."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "adaboost.npz")
_SCALE = max(1, int(os.environ.get("POF_FUZZ_SCALE", "1")))      # soak runs: more random tables

# reference feature column -> column of segment_features (oracle and pof_segment_features); the reference's
# median deviation (2), succeeding jump (4) and mean speed (13) are defects and not restated
REF_TO_OURS = {0: 0, 1: 1, 5: 4, 6: 5, 7: 6, 8: 7, 9: 8, 10: 9, 11: 10, 12: 11}


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
    from planar_optical_flow_amd.src.depracted.model.adaboost_person_det import SegmentDataset, BoostedFeatureDetector
    scans = np.stack([gold["ft%d_scan" % b] for b in range(3)])
    ds = SegmentDataset(scans, [gold["ft%d_wps" % b] for b in range(3)])
    for b in range(3):
        ref = gold["ft%d_features" % b]
        assert np.array_equal(ds.labels[b], ref[:, 14])
        assert len(ds.scans_feature[b]) == len(ref) == len(ds.segments[b])
        for rc, oc in REF_TO_OURS.items():
            scale = max(1.0, np.abs(ref[:, rc]).max())
            # the device path takes float32 scans (the fixture's are float64): the line / circle fits of nearly
            # straight segments and the curvature of tiny triangles amplify that input rounding; the tight pins are
            # oracle vs reference (CPU test above) and kernel vs oracle on identical float32 scans (below)
            tol = 5e-4 if rc in (6, 7, 8, 11, 12) else 2e-6
            assert np.allclose(ds.scans_feature[b][:, oc], ref[:, rc], rtol=tol, atol=tol * scale), (b, rc)
        # against the oracle on the same float32 scans (tolerances of tests/test_hip_parity.py for the fits)
        _, feat = R.segment_features(scans[b].astype(np.float32).astype(np.float64), R.laser_phi())
        ours = _aligned_features(feat)
        assert np.allclose(ds.scans_feature[b], ours[:, :12], rtol=1e-5, atol=1e-5 * np.abs(ours[:, :12]).max())
    assert ds.input.shape == (len(ds), 12) and set(np.unique(ds.target)) == {-1.0, 1.0}
    alpha, para = BoostedFeatureDetector(rng=np.random.default_rng(0)).adaboost(ds.input, ds.target, 6, 64)
    assert para[0, 0] >= 1 and np.isfinite(alpha).all()
