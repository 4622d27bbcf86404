"""Seeded synthetic DROW-shaped inputs.

Input generation only (host side, NumPy): range scans, odometry pairs and
person detections with the statistics of a SICK S300 sweep.  There is no
network access for the real DROW / JRDB data, so every test and the benchmark
draw from here.

Ranges r[b,t,i] (float32):
    clip(6 + 3 sin(2 phi_i + a_b) + 1.5 sin(7 phi_i + c_b), 0.3, 25)
    + K_b in {0..6} "legs" (discs of radius 0.15 m at (rho, theta))
    + N(0, 0.01^2) noise per (t, i), + 1 % drop-outs set to 29.99.
Odometry (x, y, phi) float64: one random-walk step per sample.
Detections: the leg centres, as the `wp` class; `wc`/`wa` optional.
"""
import numpy as np

PADDING_RANGE = 29.99
LEG_RADIUS = 0.15


def laser_grid(angle_inc=np.radians(0.5), num_pts=450):
    """Host copy of the angle grid, used only to place synthetic legs."""
    fov = (num_pts - 1) * angle_inc
    return np.linspace(-0.5 * fov, 0.5 * fov, num_pts)


class ScanBatch:
    """Container for one synthetic batch.

    scans  (B,T,N) float32   last row of each window is the current scan
    odom0  (B,3) float64     odometry at the template scan
    odom1  (B,3) float64     odometry at the current scan
    dets   list[B] of dict(wc=(k,2), wa=(k,2), wp=(k,2)) float64 (r, phi)
    """

    def __init__(self, scans, odom0, odom1, dets, phi):
        self.scans, self.odom0, self.odom1, self.dets, self.phi = scans, odom0, odom1, dets, phi

    def det_csr(self, pedestrian_only=False):
        """Flatten the ragged detection lists to CSR: offsets (B+1) int32,
        rphi (D,2) float64, cls (D,) uint8 with 0=wc, 1=wa, 2=wp, in the
        reference's concatenation order wc + wa + wp (utils.py:170)."""
        offs = [0]
        rphi, cls = [], []
        for d in self.dets:
            groups = [(2, d["wp"])] if pedestrian_only else [(0, d["wc"]), (1, d["wa"]), (2, d["wp"])]
            for c, arr in groups:
                arr = np.asarray(arr, dtype=np.float64).reshape(-1, 2)
                rphi.append(arr)
                cls.append(np.full(len(arr), c, dtype=np.uint8))
            offs.append(offs[-1] + sum(len(np.asarray(a).reshape(-1, 2)) for _, a in groups))
        rphi = np.concatenate(rphi, axis=0) if rphi else np.zeros((0, 2))
        cls = np.concatenate(cls) if cls else np.zeros((0,), dtype=np.uint8)
        return np.asarray(offs, dtype=np.int32), np.ascontiguousarray(rphi), cls


def make_batch(seed, B, T=2, N=450, angle_inc=np.radians(0.5), max_legs=6, mixed_classes=False,
               dropout=0.01, noise=0.01):
    rng = np.random.default_rng(seed)
    phi = laser_grid(angle_inc, N)
    a = rng.uniform(0, 2 * np.pi, (B, 1))
    c = rng.uniform(0, 2 * np.pi, (B, 1))
    base = np.clip(6 + 3 * np.sin(2 * phi[None] + a) + 1.5 * np.sin(7 * phi[None] + c), 0.3, 25)
    k = rng.integers(0, max_legs + 1, B)
    dets = []
    for b in range(B):
        rho = rng.uniform(1.0, 10.0, k[b])
        th = rng.uniform(phi[0], phi[-1], k[b])
        for r_, t_ in zip(rho, th):
            lat = r_ * np.sin(phi - t_)
            front = np.cos(phi - t_) > 0
            hit = (np.abs(lat) < LEG_RADIUS) & front
            depth = r_ * np.cos(phi[hit] - t_) - np.sqrt(LEG_RADIUS ** 2 - lat[hit] ** 2)
            base[b, hit] = np.minimum(base[b, hit], depth)
        d = np.stack([rho, th], axis=1)
        if mixed_classes and k[b] > 0:
            lab = rng.integers(0, 3, k[b])
            dets.append({"wc": d[lab == 0], "wa": d[lab == 1], "wp": d[lab == 2]})
        else:
            dets.append({"wc": np.zeros((0, 2)), "wa": np.zeros((0, 2)), "wp": d})
    scans = base[:, None, :] + rng.normal(0.0, noise, (B, T, N))
    scans = np.clip(scans, 0.05, 29.0)
    drop = rng.random((B, T, N)) < dropout
    scans[drop] = PADDING_RANGE
    scans = scans.astype(np.float32)
    odom0 = np.concatenate([rng.uniform(-5, 5, (B, 2)), rng.uniform(-np.pi, np.pi, (B, 1))], axis=1)
    step = np.concatenate([rng.uniform(-0.05, 0.05, (B, 2)), rng.uniform(-0.03, 0.03, (B, 1))], axis=1)
    odom1 = odom0 + step
    return ScanBatch(scans, odom0, odom1, dets, phi)
