"""Per-kernel timing on the GPU box (events on torch's current stream, which is the
stream the C-ABI launches on).  Usage: python tools/bench_kernels.py [cutout] [attn] [corr] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import ops, synth

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

which = set(sys.argv[1:]) or {"cutout", "attn", "corr"}
dev = "cuda"
if "cutout" in which:
    tab = ops.phi_table()
    for (B, T, fixed, P) in [(2048, 5, True, 56), (1024, 11, False, 56), (2048, 5, True, 48)]:
        sb = synth.make_batch(seed=3, B=B, T=T)
        scans = torch.from_numpy(sb.scans).to(dev)
        out = torch.empty((B, 450, T, P), dtype=torch.float32, device=dev)
        kw = dict(fixed=fixed, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=P, padding_val=29.99, area_mode=True)
        ms = timeit(lambda: ops.cutout(scans, tab, out=out, **kw))
        byt = B * (T * 450 * 4 + 450 * T * P * 4)
        print("cutout B=%d T=%d fixed=%d P=%d: %.3f ms  %.0f GB/s  %.2f Msamples/s" % (B, T, fixed, P, ms, byt / ms / 1e6, B / ms / 1e3))
        ms = timeit(lambda: ops.cutout(scans, tab, out=out, exact_values=False, **kw))
        print("   float32 value path: %.3f ms  %.0f GB/s" % (ms, byt / ms / 1e6))
    # dense 3600-pt geometry (rows do not fit LDS)
    tab2 = ops.phi_table(np.radians(0.1), 3600)
    sb = synth.make_batch(seed=5, B=64, T=11, N=3600, angle_inc=np.radians(0.1))
    scans = torch.from_numpy(sb.scans).to(dev)
    out = torch.empty((64, 3600, 11, 56), dtype=torch.float32, device=dev)
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=56, padding_val=29.99, area_mode=True)
    ms = timeit(lambda: ops.cutout(scans, tab2, out=out, **kw), iters=5)
    byt = 64 * (11 * 3600 * 4 + 3600 * 11 * 56 * 4)
    print("cutout dense3600 B=64 T=11: %.3f ms  %.0f GB/s" % (ms, byt / ms / 1e6))
if "attn" in which:
    for B in (16, 64, 256):
        N, E, F = 450, 128, 3584
        g = torch.Generator(device=dev).manual_seed(10)
        ex = torch.randn((B, N, E), device=dev, generator=g) * 0.3
        et = torch.randn((B, N, E), device=dev, generator=g) * 0.3
        x = torch.randn((B, N, F), device=dev, generator=g)
        t = torch.randn((B, N, F), device=dev, generator=g)
        out = torch.empty_like(x)
        per = 3 * N * F * 4 + 2 * N * E * 4 + 2 * N * 11 * 4
        ms = timeit(lambda: ops.spatial_attention(ex, et, x, t, 0.5, 11, out=out), iters=10)
        print("attn B=%d: %.3f ms  %.0f GB/s" % (B, ms, per * B / ms / 1e6))
if "corr" in which:
    for (B, C, n) in [(4096, 256, 57), (256, 256, 450)]:
        f1 = torch.randn((B, C, n), device=dev); f2 = torch.randn((B, C, n), device=dev)
        out = torch.empty((B, 11, n), device=dev)
        ms = timeit(lambda: ops.band_correlation(f1, f2, 3, 5, out=out), iters=10)
        byt = B * (2 * C * n * 4 + 11 * n * 4)
        print("corr B=%d C=%d n=%d: %.3f ms  %.0f GB/s  %.1f TFLOP/s" % (B, C, n, ms, byt / ms / 1e6, B * 2 * 11 * n * 3 * C / ms / 1e9))
if "bwd" in which:
    for (B, C, n) in [(4096, 256, 57)]:
        f1 = torch.randn((B, C, n), device=dev); f2 = torch.randn((B, C, n), device=dev)
        g = torch.randn((B, 11, n), device=dev)
        ms = timeit(lambda: ops.band_correlation_backward(f1, f2, g, 3, 5), iters=10)
        byt = B * (4 * C * n * 4 + 11 * n * 4)
        print("corr backward B=%d C=%d n=%d: %.3f ms  %.0f GB/s" % (B, C, n, ms, byt / ms / 1e6))
    for B in (64,):
        N, E, F = 450, 128, 3584
        gen = torch.Generator(device=dev).manual_seed(10)
        ex = torch.randn((B, N, E), device=dev, generator=gen) * 0.3
        et = torch.randn((B, N, E), device=dev, generator=gen) * 0.3
        x = torch.randn((B, N, F), device=dev, generator=gen)
        t = torch.randn((B, N, F), device=dev, generator=gen)
        out, band, prob = ops.spatial_attention(ex, et, x, t, 0.5, 11)
        go = torch.randn_like(out); gb = torch.randn_like(band)
        per = 4 * N * F * 4 + 4 * N * E * 4 + 3 * N * 11 * 4     # read g, tmpl; write dx, dtmpl; emb in/out
        for fused in (False, True):
            ms = timeit(lambda: ops.spatial_attention_backward(ex, et, t, prob, go, gb, 0.5, 11, fused=fused), iters=10)
            print("attn backward B=%d [%s]: %.3f ms  %.0f GB/s" % (B, "fused" if fused else "two-pass", ms,
                                                                    per * B / ms / 1e6))
if "polar" in which:
    B, Tn, N = 2048, 5, 450
    sb = synth.make_batch(seed=3, B=B, T=Tn)
    scans = torch.from_numpy(sb.scans).to(dev)
    out = torch.empty((B, Tn, 31, N), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.polar_grid(scans, out=out))
    byt = B * Tn * N * 4 * 32
    print("polar grid B=%d T=%d: %.3f ms  %.0f GB/s" % (B, Tn, ms, byt / ms / 1e6))
if "f16" in which:
    for (B, C, n) in [(4096, 256, 57), (256, 256, 450)]:
        f1 = torch.randn((B, C, n), device=dev).half(); f2 = torch.randn((B, C, n), device=dev).half()
        out = torch.empty((B, 11, n), device=dev)
        ms = timeit(lambda: ops.band_correlation(f1, f2, 3, 5, out=out), iters=10)
        byt = B * (2 * C * n * 2 + 11 * n * 4)
        print("corr f16 B=%d C=%d n=%d: %.3f ms  %.0f GB/s" % (B, C, n, ms, byt / ms / 1e6))
    for B in (64, 256):
        N, E, F = 450, 128, 3584
        gen = torch.Generator(device=dev).manual_seed(10)
        ex = torch.randn((B, N, E), device=dev, generator=gen) * 0.3
        et = torch.randn((B, N, E), device=dev, generator=gen) * 0.3
        x = torch.randn((B, N, F), device=dev, generator=gen).half()
        t = torch.randn((B, N, F), device=dev, generator=gen).half()
        out = torch.empty_like(x)
        ms = timeit(lambda: ops.spatial_attention(ex, et, x, t, 0.5, 11, out=out), iters=10)
        per = 3 * N * F * 2 + 2 * N * E * 4 + 2 * N * 11 * 4
        print("attn f16 B=%d: %.3f ms  %.0f GB/s" % (B, ms, per * B / ms / 1e6))

if "boost" in which:
    # N4: boosted stumps.  Device rounds vs the NumPy restatement (which is already far faster than the
    # reference's interpreted per-sample loops) on a segment table of realistic size.
    import time
    from oracle import ref_numpy as R
    from planar_optical_flow_amd.src.depracted.model.adaboost_person_det import BoostedFeatureDetector
    rng = np.random.default_rng(0)
    N, D, K, ns = 50000, 12, 60, 200
    X = rng.normal(size=(N, D))
    Y = np.where(X @ rng.normal(size=D) + 1.5 * rng.normal(size=N) > 0, 1.0, -1.0)
    BoostedFeatureDetector(rng=np.random.default_rng(5)).adaboost(X[:1000], Y[:1000], 2, ns)      # warm-up
    det = BoostedFeatureDetector(rng=np.random.default_rng(1))
    t0 = time.time(); a, p = det.adaboost(X, Y, K, ns); torch.cuda.synchronize(); t1 = time.time()
    print("adaboost HIP   N=%d D=%d K=%d nSamples=%d: %.3f s (%.2f ms/round)" % (N, D, K, ns, t1 - t0, (t1 - t0) / K * 1e3))
    t0 = time.time(); a2, p2 = R.adaboost(X, Y, 6, ns, rng=np.random.default_rng(1)); t1 = time.time()
    print("adaboost NumPy restatement, 6 rounds: %.3f s (%.1f ms/round)" % (t1 - t0, (t1 - t0) / 6 * 1e3))
    assert np.array_equal(a[:6], a2) and np.array_equal(p[:6], p2)
    Xd = torch.from_numpy(X).cuda()
    ms = timeit(lambda: det.eval(Xd, a, p))
    print("eval (stump vote) N=%d K=%d: %.3f ms incl. D2H" % (N, K, ms))
    idx = torch.from_numpy(rng.integers(0, N, 2048).astype(np.int32)).cuda()
    Yd = torch.from_numpy(Y).cuda()
    for n in (200, 2048):
        ms = timeit(lambda: det._search(Xd, Yd, idx[:n].contiguous(), n))
        print("stump search n=%d D=%d: %.3f ms incl. D2H of the %d results" % (n, D, ms, 5 * D))
