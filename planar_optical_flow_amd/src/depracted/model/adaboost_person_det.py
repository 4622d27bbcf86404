"""Boosted decision-stump person detector on jump-distance segments (reference:
src/depracted/model/adaboost_person_det.py).

The reference module is a script (argv parsing and an import of a data handle that is not in the repository
at import time); its pieces are rebuilt here as a library:

* ``BoostedFeatureDetector`` (:212-378) -- ``simple_classifier`` runs as ONE launch of ``pof_stump_search``
  over all feature dimensions (sort + scan instead of the reference's interpreted D x T x n comparison
  loops), ``eval`` as one launch of ``pof_stump_vote``; ``adaboost`` keeps the sample weights and the
  weighted resampling on the host in the reference's exact NumPy arithmetic (so a seeded run draws the same
  samples) and sends only the sampled row numbers to the device each round.
* ``SegmentDataset`` (:39-210) -- segments, labels and features of many scans from one launch of
  ``pof_segment_features`` (A13).
* ``nms_predicted_center`` (:11-37).
"""
import numpy as np
import torch

from .... import _lib, ops

_MAX_SAMPLES = 2048


def _dev_f64(x, device):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)) if isinstance(x, np.ndarray) else x
    return t.to(device=device, dtype=torch.float64).contiguous()


def _host_f64(x):
    return x.detach().cpu().numpy().astype(np.float64, copy=False) if torch.is_tensor(x) else np.asarray(x, np.float64)


class BoostedFeatureDetector:
    def __init__(self, device="cuda", rng=None):
        self.device = torch.device(device)
        self._rng = np.random if rng is None else rng      # the reference draws from the global NumPy state

    # ---- device side ---------------------------------------------------------------------
    def _search(self, Xd, Yd, index, n):
        D = Xd.shape[1]
        dev = Xd.device
        ints = torch.empty((3, D), dtype=torch.int32, device=dev)         # min_err, max_err, n_thresh
        thetas = torch.empty((2, D), dtype=torch.float64, device=dev)     # theta_min, theta_max
        with torch.cuda.device(dev):
            _lib.call("pof_stump_search", Xd.data_ptr(), Yd.data_ptr(), Xd.shape[0],
                      index.data_ptr() if index is not None else None, int(n), D,
                      ints[0].data_ptr(), thetas[0].data_ptr(), ints[1].data_ptr(), thetas[1].data_ptr(),
                      ints[2].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return ints.cpu().numpy(), thetas.cpu().numpy()

    @staticmethod
    def _select(ints, thetas, n):
        """The reference's walk over the dimensions (:297-345): keep the first dimension that strictly lowers the
        least error, where a dimension offers min(error)/n and min(1 - error/n); theta is the first threshold
        reaching whichever of the two equals the new least error."""
        least, j, theta = 1, 1, 0
        for jj in range(ints.shape[1]):
            if ints[2, jj] == 0:
                raise ValueError("simple_classifier: dimension %d has no threshold between samples of different class"
                                 % jj)              # the reference fails here too (min() of an empty sequence)
            le1 = ints[0, jj] / np.float64(n)
            le2 = 1 - ints[1, jj] / np.float64(n)
            le0 = min([le1, le2, least])
            if least == le0:
                continue
            least, j = le0, jj + 1
            theta = thetas[0, jj] if le1 == least else thetas[1, jj]
        return j, theta

    def _prepare(self, X, Y):
        Xh, Yh = _host_f64(X), _host_f64(Y).reshape(-1)
        if Xh.ndim != 2 or len(Yh) != len(Xh):
            raise ValueError("X must be [N, D] and Y [N] (or [N, 1])")
        if not np.isfinite(Xh).all():
            raise ValueError("features must be finite")
        if not np.isin(Yh, (-1.0, 1.0)).all():
            raise ValueError("labels must be +1 / -1")
        return Xh, Yh, _dev_f64(X, self.device), _dev_f64(Yh, self.device)

    # ---- reference API -------------------------------------------------------------------
    def simple_classifier(self, X, Y):
        """-> (j, theta): 1-based feature dimension and threshold of the best stump on (X [n, D], Y [n])."""
        Xh, _, Xd, Yd = self._prepare(X, Y)
        n = len(Xh)
        if not 2 <= n <= _MAX_SAMPLES:
            raise ValueError("simple_classifier takes 2..%d samples (the boosting loop draws nSamples per round)" % _MAX_SAMPLES)
        return self._select(*self._search(Xd, Yd, None, n), n)

    def adaboost(self, X, Y, K, nSamples):
        """-> (alphaK [K], para [K, 2] = (j, theta) per round); rounds after an early stop stay zero (:216-281)."""
        Xh, Yh, Xd, Yd = self._prepare(X, Y)
        if not 2 <= nSamples <= _MAX_SAMPLES:
            raise ValueError("nSamples must be in 2..%d" % _MAX_SAMPLES)
        N = len(Xh)
        Yc = Yh.reshape(N, 1)
        j, theta, alpha = np.zeros(K), np.zeros(K), np.zeros(K)
        w = np.ones((N, 1))
        w[Yc == 1.0] = 1 / np.sum(Yc == 1.0) / 2
        w[Yc == -1.0] = 1 / np.sum(Yc == -1.0) / 2
        w = w / np.sum(w)
        for k in range(K):
            index = self._rng.choice(N, nSamples, True, w.ravel())
            idx_dev = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int32)).to(self.device)
            j[k], theta[k] = self._select(*self._search(Xd, Yd, idx_dev, nSamples), nSamples)
            cY = np.where(Xh[:, int(j[k] - 1)] > theta[k], 1.0, -1.0).reshape(N, 1)
            ek = np.sum(w * (Yc != cY))
            if ek < 1.0e-01:
                alpha[k] = 1
                break
            alpha[k] = 0.5 * np.log((1 - ek) / ek)
            w = w * np.exp(-alpha[k] * (Yc * cY))
            w = w / np.cumsum(w)[-1]             # left-to-right total, as the reference's builtin sum(w)
        return alpha, np.stack((j, theta), axis=1)

    def eval(self, X, alphaK, para):
        """-> (classLabels [N], result [N]) as NumPy arrays (:349-378)."""
        para = np.asarray(para, np.float64).reshape(-1, 2)
        Xd = _dev_f64(X, self.device)
        N, D = Xd.shape
        dims = para[:, 0].astype(np.int32)
        if len(dims) and (dims.min() < 0 or dims.max() > D):
            raise IndexError("para[:, 0] must be a 1-based dimension of X")
        dev = self.device
        dim_d = torch.from_numpy(dims).to(dev)
        th_d, al_d = _dev_f64(para[:, 1].copy(), dev), _dev_f64(np.asarray(alphaK, np.float64).reshape(-1), dev)
        if len(al_d) != len(dims):
            raise ValueError("alphaK and para must have one row per round")
        out = torch.empty((2, N), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _lib.call("pof_stump_vote", Xd.data_ptr(), N, D, dim_d.data_ptr(), th_d.data_ptr(), al_d.data_ptr(),
                      len(dims), out[0].data_ptr(), out[1].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        res = out.cpu().numpy()
        return res[1], res[0]


def nms_predicted_center(data, preds, scores, min_dist=1.0):
    """data: list of [segment, label]; preds / scores: arrays.  Visits the segments by descending prediction
    and clears the score of every segment whose centre lies within ``min_dist`` of a kept one (:11-37).
    -> (segments, preds, scores) in visiting order.  Equal predictions keep the reverse input order (the
    reference leaves ties to np.argsort)."""
    preds, scores = np.asarray(preds), np.array(scores, dtype=np.float64)
    order = np.argsort(preds, kind="stable")[::-1]
    preds, scores = preds[order], scores[order]
    segments = [data[i][0] for i in order]
    centers = np.array([np.mean(s, axis=0) for s in segments]).reshape(len(segments), -1)
    diff = centers[:, None, :2] - centers[None, :, :2]
    dist = np.sqrt(np.square(diff[..., 0]) + np.square(diff[..., 1]))
    for i in range(len(segments)):
        if scores[i] <= 0.0:
            continue
        keep = scores[i]
        scores[dist[i] < min_dist] = 0.0
        scores[i] = keep
    return segments, preds, scores


class SegmentDataset:
    """Segments of many scans with their boosting features and labels (reference ``Dataset``, :39-210).

    scans [S, N] ranges; dets: per scan a list / array of annotated xy positions; odom_t [S]: the per-scan
    odometry value the reference keeps as ``data[-1]`` (default zeros).  A segment is positive when its centre lies
    within ``radius_wp`` of an annotation (:84-88); segments of fewer than three points are dropped (:53-55).
    ``input`` is the reference's table: its 14 columns in its order, *with* its data-set coupled definitions
    (Frobenius "median deviation", succeeding jump to kept[min(idx + 1, 3)], mean speed over the idx-th piece of
    the unfiltered split against the next scan of the set, next = min(idx + 1, S - 1)); like the reference it raises
    IndexError for a scan with fewer than four kept segments.  All of it comes from one ``pof_segment_features_ex``
    launch (one wave per segment)."""

    def __init__(self, scans, dets, odom_t=None, angle_inc=np.radians(0.5), radius_wp=0.5, jump_dist=0.5,
                 device="cuda"):
        scans = torch.as_tensor(np.asarray(scans, dtype=np.float32)).to(device)
        S, N = scans.shape
        dev = scans.device
        tab = ops.phi_table(angle_inc, N, device=dev)
        nxt = scans[torch.clamp(torch.arange(S, device=dev) + 1, max=max(S - 1, 0))].contiguous()
        ot = np.zeros(S) if odom_t is None else np.asarray(odom_t, dtype=np.float64).reshape(S)
        dt = torch.from_numpy(ot[np.minimum(np.arange(S) + 1, max(S - 1, 0))] - ot).to(dev)
        wps = [np.asarray(d, dtype=np.float64).reshape(-1, 2) for d in dets]
        offs = np.zeros(S + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(w) for w in wps])
        wxy = np.concatenate(wps + [np.zeros((1, 2))])              # one spare row keeps the pointer valid
        seg_id, num, kept, ref = ops.segment_features_reference(
            scans, tab, nxt, dt, torch.from_numpy(offs).to(dev), torch.from_numpy(wxy).to(dev),
            radius_wp=radius_wp, jump_dist=jump_dist)
        xy = ops.scan_preprocess(scans, tab, want=("xy",), out_dtype=torch.float64)["xy"]
        seg_id_h, kept_h, ref_h, xy_h = seg_id.cpu().numpy(), kept.cpu().numpy(), ref.cpu().numpy(), xy.cpu().numpy()
        self.scans_feature, self.labels, self.segments = [], [], []
        for s in range(S):
            K = int(kept_h[s])
            if K < 4:
                raise IndexError("list index out of range")     # the reference's segments[min(idx + 1, 3)] (:135)
            f = ref_h[s, :K]
            ids, counts = np.unique(seg_id_h[s], return_counts=True)
            segs = [xy_h[s][seg_id_h[s] == q] for q in ids[counts > 2]]
            self.scans_feature.append(f[:, :14])
            self.labels.append(f[:, 14].copy())
            self.segments.append([[seg, l] for seg, l in zip(segs, f[:, 14])])
        self.input = np.vstack(self.scans_feature) if S else np.zeros((0, 14))
        self.target = np.hstack(self.labels) if S else np.zeros(0)

    def __len__(self):
        return len(self.input)
