#!/usr/bin/env python3
"""Throughput of the planar-flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload = BASELINE.json configs[1]: batch 4096 synthetic 450-point
scan pairs, flow-only (no detector head): one step = ONE launch of
pof_scan_preprocess over one batch per rank (scan -> xy -> rigid-motion flow ->
canonical frame, detection association, regression target, exclude mask).
Inputs are resident in HBM before the timed region; a ring of distinct batches
larger than the 256 MiB Infinity Cache is cycled so the stream really comes from
HBM.  Batches shard over ranks with no data-path collective (weak scaling: every
rank processes its own 4096 scans per step).

One JSON line on stdout (rank 0).  Extra objects:
  roofline      dominant kernel of the headline step (scan_preprocess_kernel)
  cpu_baseline  the NumPy oracle (a port of the reference's per-sample path)
                timed on this host's cores, rank 0, N=1 only, bounded sample
  cutout        secondary measurement: the A8 cutout kernel at BASELINE config 3
                shape with its own HBM roofline (not part of `value`)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_PTS = 450
BATCH = 4096


def pmc_traffic(prefixes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json,
    produced by tools/collect_profiles.sh with separate --pmc FETCH_SIZE / WRITE_SIZE runs).
    Units are KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md
    section HBM), hence the factor 2 on the read side.  Returns (bytes, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        tot = 0.0
        for k, v in d.items():
            if k.startswith(prefixes):
                tot += (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
        return (tot if tot > 0 else None), os.path.relpath(files[-1], REPO)
    except Exception:  # noqa: BLE001
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--ring", type=int, default=8, help="distinct resident batches cycled through")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--pipeline", action="store_true",
                    help="graph: params launch of batch i+1 on a second stream (measured: no gain, see DESIGN.md)")
    ap.add_argument("--no-chain", dest="chained", action="store_false",
                    help="two launches per step (params + streaming) instead of the chained single launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-model", action="store_true",
                    help="skip the DR-SPAAM forward extra (PMC passes: keeps the per-kernel averages per shape)")
    ap.add_argument("--no-extra", action="store_true")
    return ap.parse_args()


def cpu_sample_worker(args):
    """Reference execution shape: one __getitem__-style call per sample
    (dataset_dr_spaam.py:384-409), NumPy oracle."""
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import ref_numpy as R
    scans, odom0, odom1, dets = args
    phi = R.laser_phi()
    out = []
    for b in range(len(scans)):
        cur = scans[b]
        phi = R.laser_phi()
        cls, reg = R.regression_target(cur, phi, [], [], dets[b])
        xy = np.array(R.polar_to_xy(cur, phi)).T
        flow = R.flow_to_canonical(R.displacement_from_odometry(xy, odom0[b], odom1[b]), phi)
        mask = R.dynamic_mask(xy, [], [], dets[b]) * R.valid_point_mask(cur)
        out.append((flow, cls, reg, mask))
    # collate (dataset_dr_spaam.py:464-471)
    return [np.array([o[k] for o in out]) for k in range(4)]


def cpu_baseline(sb, budget_s=12.0):
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 16)
    n = 256 * cores
    n = min(n, len(sb.scans))
    chunks = np.array_split(np.arange(n), cores)
    jobs = [(sb.scans[c, -1], sb.odom0[c], sb.odom1[c], [sb.dets[i]["wp"] for i in c]) for c in chunks]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(cpu_sample_worker, jobs[:cores])  # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:
            pool.map(cpu_sample_worker, jobs)
            reps += 1
            if time.perf_counter() - t0 > budget_s or reps >= 20:
                break
        dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "scans/s", "cores": cores, "kind": "port",
            "sample": "%d x %d scans of the headline workload, per-sample NumPy oracle calls + collate in a "
                      "%d-process pool (reference DataLoader shape)" % (reps, n, cores)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    from planar_optical_flow_amd import synth
    B, N = a.batch, N_PTS
    sb = synth.make_batch(seed=2 + 1000 * rank, B=B, T=2, N=N)
    # CPU baseline first: the worker pool is forked before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(sb)

    # test hooks (single-GPU rehearsal of the N > 1 path): POF_BENCH_SHARE_GPU=1 puts every rank
    # on cuda:0, POF_BENCH_BACKEND=gloo replaces RCCL (which needs one device per rank)
    if os.environ.get("POF_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("POF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from planar_optical_flow_amd import ops

    offs, rphi, cls = sb.det_csr()
    tab = ops.phi_table(device=dev)
    ring = []
    for r in range(a.ring):
        sh = (r * 509) % B  # distinct memory and distinct content per ring slot
        scans = torch.from_numpy(np.roll(sb.scans, sh, axis=0)).to(dev)
        o0 = torch.from_numpy(np.roll(sb.odom0, sh, axis=0)).to(dev)
        o1 = torch.from_numpy(np.roll(sb.odom1, sh, axis=0)).to(dev)
        counts = np.roll(np.diff(offs), sh)
        order = np.roll(np.arange(B), sh)
        ro = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        rr = np.concatenate([rphi[offs[i]:offs[i + 1]] for i in order]) if len(rphi) else rphi
        det = ops.DetCSR.from_numpy(ro, rr, np.full(len(rr), 2, np.uint8), dev)
        outs = {
            "flow": torch.empty((B, N, 2), dtype=torch.float32, device=dev),
            "target_cls": torch.empty((B, N), dtype=torch.int64, device=dev),
            "target_reg": torch.empty((B, N, 2), dtype=torch.float32, device=dev),
            "exclude_mask": torch.empty((B, N), dtype=torch.float32, device=dev),
        }
        ws = torch.empty(ops.scan_preprocess_workspace_bytes(B, len(rr)), dtype=torch.uint8, device=dev)
        ring.append((scans, o0, o1, det, outs, ws))
    want = ("flow", "target_cls", "target_reg", "exclude_mask")

    def step(i, phases=3):
        scans, o0, o1, det, outs, ws = ring[i % a.ring]
        if a.chained and phases == 3:
            # one launch: stream batch i (its params were produced by the previous step's launch)
            # and evaluate the params of batch i+1 on extra workgroups of the same grid
            _, n0, n1, ndet, _, nws = ring[(i + 1) % a.ring]
            ops.scan_preprocess(scans, tab, o0, o1, det, want=want, out=outs, workspace=ws,
                                next_batch={"odom0": n0, "odom1": n1, "dets": ndet, "workspace": nws})
        else:
            ops.scan_preprocess(scans, tab, o0, o1, det, want=want, out=outs, workspace=ws, phases=phases)

    if a.chained:
        step(0, phases=1)  # prime the chain: params of ring slot 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- optional hipGraph of one trip round the ring --------------------------
    graph = None
    if not a.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(a.ring):
                step(i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if not a.pipeline:
                for i in range(a.ring):
                    step(i)
            else:
                # two-stream software pipeline over the independent batches of the ring: the
                # tiny per-sample params launch of batch i+1 runs under the streaming launch of
                # batch i (each batch has its own workspace; an event orders params(i) -> main(i))
                main = torch.cuda.current_stream()
                side2 = torch.cuda.Stream()
                side2.wait_stream(main)
                evs = []
                with torch.cuda.stream(side2):
                    for i in range(a.ring):
                        step(i, phases=1)
                        ev = torch.cuda.Event()
                        ev.record(side2)
                        evs.append(ev)
                for i in range(a.ring):
                    main.wait_event(evs[i])
                    step(i, phases=2)
                main.wait_stream(side2)

    # K or W need not be multiples of the ring: the remainder steps get their own captured graph
    # (an eager launch costs tens of microseconds of Python per step, several times the kernel)
    rem_graphs = {}
    if graph is not None and not a.pipeline:
        for rem in sorted({a.steps % a.ring, a.warmup % a.ring} - {0}):
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(rem):
                    step(i)
            rem_graphs[rem] = gr

    def run(k):
        if graph is not None:
            full, rem = divmod(k, a.ring)
            for _ in range(full):
                graph.replay()
            if rem in rem_graphs:
                rem_graphs[rem].replay()
            else:
                for i in range(rem):
                    step(i)
        else:
            for i in range(k):
                step(i)

    run(a.warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(a.steps)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- parity of what was just computed (outside the timed region) ------------
    from oracle import ref_numpy as R
    scans, o0, o1, det, outs, _ = ring[0]
    phi = R.laser_phi()
    flow = outs["flow"][:64].cpu().numpy()
    epe = 0.0
    for b in range(64):
        cur = sb.scans[b, -1]
        xy = np.array(R.polar_to_xy(cur, phi)).T
        ref = R.flow_to_canonical(R.displacement_from_odometry(xy, sb.odom0[b], sb.odom1[b]), phi)
        epe += float(np.linalg.norm(flow[b] - ref, axis=-1).mean()) / 64

    result = None
    if rank == 0:
        # algorithmic bytes per scan: 4N range row in; out 8N flow f32 +
        # 8N target_cls int64 + 8N target_reg + 4N exclude mask  = 32N = 14 400 B
        bytes_per_scan = 4 * N + (8 + 8 + 8 + 4) * N
        launch_ms = dev_ms / a.steps
        achieved = bytes_per_scan * B / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(("scan_preprocess_chain_kernel",) if a.chained
                                           else ("scan_params_kernel", "scan_preprocess_kernel"))
        if B != BATCH:
            traffic = None
        result = {
            "metric": "scans/sec, flow-only preprocess of 450-pt synthetic scan pairs, batch 4096 per GPU "
                      "(+ flow EPE vs reference oracle)",
            "value": world * B * a.steps / dt,
            "unit": "scans/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch %d x 450-pt scan pairs, flow-only "
                                   "(A1-A7 fused: xy, displacement flow, canonical frame, association, "
                                   "regression target, exclude mask), float32 outputs" % B,
                       "global_batch": world * B, "ring_batches": a.ring,
                       "launch": ("eager" if graph is None else
                                  "hipGraph replay" + (", params launch of batch i+1 on a second stream" if a.pipeline else "")
                                  + (", chained (params of batch i+1 ride in the launch of batch i)" if a.chained else "")),
                       "parallelism": "batch-sharded x%d, no collective" % world},
            "epe_vs_oracle_m": epe,
            "roofline": {"bound": "hbm", "kernel": ("scan_preprocess_chain_kernel<float,2,1>" if a.chained
                                    else "scan_params_kernel + scan_preprocess_kernel<float,2,1>"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "bytes_per_launch": bytes_per_scan * B, "launch_ms": launch_ms},
        }
        if not a.no_extra and world == 1:   # per-kernel extras only on the single-GPU line
            result["cutout"] = bench_cutout(ops, synth, tab, dev, variants=not a.no_model)
            result["spatial_attention"] = bench_attention(ops, dev)
            result["band_correlation"] = bench_band_corr(ops, dev)
            if not a.no_model:
                result["dr_spaam_forward"] = bench_dr_spaam(ops, synth, tab, dev)
            # PMC traffic of the same shapes (the profile run executes this very function)
            result["cutout"]["roofline"]["traffic"] = pmc_traffic(("cutout_area_kernel", "cutout_kernel<1, 7, 1,"))[0]
            result["spatial_attention"]["roofline"]["traffic"] = pmc_traffic(("attn_",))[0]
            result["band_correlation"]["roofline"]["traffic"] = pmc_traffic(("band_corr_",))[0]
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_baseline"] = result["value"] / cpu["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _time_kernel(torch, fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_cutout(ops, synth, tab, dev, variants=True):
    """A8 at BASELINE config 3 shape: T=5, 450 pts, P=56, dr_spaam.yaml window."""
    import torch
    B, T, N, P = 2048, 5, N_PTS, 56
    sb = synth.make_batch(seed=3, B=B, T=T, N=N)
    scans = torch.from_numpy(sb.scans).to(dev)
    out = torch.empty((B, N, T, P), dtype=torch.float32, device=dev)
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=P,
              padding_val=29.99, area_mode=True)
    ms = _time_kernel(torch, lambda: ops.cutout(scans, tab, out=out, **kw), 10)
    per_sample = T * N * 4 + N * T * P * 4  # 513 000 B
    ach = per_sample * B / (ms * 1e-3) / 1e9
    # the two opt-in forms (same indices): float32 value arithmetic; float16 output storage (config 5)
    ms_f32 = ms_f16 = None
    if variants:   # skipped in the PMC passes so that the per-kernel counter averages stay per shape
        ms_f32 = _time_kernel(torch, lambda: ops.cutout(scans, tab, out=out, exact_values=False, **kw), 10)
        out16 = torch.empty((B, N, T, P), dtype=torch.float16, device=dev)
        ms_f16 = _time_kernel(torch, lambda: ops.cutout(scans, tab, out=out16, out_dtype=torch.float16, **kw), 10)
    return {"workload": "cutout T=5 N=450 P=56 area_mode, batch %d" % B, "ms_per_call": ms,
            "samples_per_s": B / (ms * 1e-3),
            "variants_ms": {"float32_value_path": ms_f32, "float16_output": ms_f16},
            "roofline": {"bound": "hbm", "kernel": "cutout_area_kernel + cutout_kernel", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}}


def bench_attention(ops, dev):
    """A10 at DR-SPAAM shape: N=450, F=256*14, E=128, w=11."""
    import torch
    B, N, E, F = 256, N_PTS, 128, 3584
    g = torch.Generator(device=dev).manual_seed(10)
    ex = torch.randn((B, N, E), device=dev, generator=g) * 0.3
    et = torch.randn((B, N, E), device=dev, generator=g) * 0.3
    x = torch.randn((B, N, F), device=dev, generator=g)
    t = torch.randn((B, N, F), device=dev, generator=g)
    out = torch.empty_like(x)
    ms = _time_kernel(torch, lambda: ops.spatial_attention(ex, et, x, t, 0.5, 11, out=out), 10)
    per = 3 * N * F * 4 + 2 * N * E * 4 + 2 * N * 11 * 4
    ach = per * B / (ms * 1e-3) / 1e9
    return {"workload": "spatial attention N=450 F=3584 E=128 w=11, batch %d" % B, "ms_per_call": ms,
            "roofline": {"bound": "hbm", "kernel": "attn_band_kernel + attn_merge_kernel<11>", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}}


def bench_dr_spaam(ops, synth, tab, dev):
    """BASELINE configs[2]: 5-scan DROW windows -> cutout -> DR-SPAAM (SpatialDROW, random-init weights of
    the reference architecture, eval mode): trunk on pof_conv3_bn_lrelu (float32 MFMA), gate on the HIP
    attention, embedding GEMM and the two 1x1 heads on hipBLASLt / MIOpen through torch."""
    import torch
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW
    B, T, N, P = 32, 5, N_PTS, 56
    torch.manual_seed(3)
    model = SpatialDROW(num_scans=T, num_pts=P, alpha=0.5, window_size=11, pedestrian_only=True).to(dev).eval()
    model.fuse_for_inference()
    sb = synth.make_batch(seed=3, B=B, T=T, N=N)
    scans = torch.from_numpy(sb.scans).to(dev)
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=P,
              padding_val=29.99, area_mode=True)

    def step():
        with torch.no_grad():
            return model(ops.cutout(scans, tab, **kw))
    ms = _time_kernel(torch, step, 5, warm=2)
    # conv3 layers: 2 * L * 3 * Ci * Co per sequence; embedding GEMMs: 2 per gate step
    seq_a = 2.0 * (56 * 3 * (1 * 64 + 64 * 64 + 64 * 128) + 28 * 3 * (128 * 128 * 2 + 128 * 256))
    seq_c = 2.0 * (14 * 3 * (256 * 256 * 2 + 256 * 512) + 7 * 3 * (512 * 256 + 256 * 128))
    flops = B * (N * T * seq_a + N * seq_c + (T - 1) * 2 * 2.0 * N * 3584 * 128)
    ach = flops / (ms * 1e-3) / 1e12
    return {"workload": "DR-SPAAM forward (cutout + SpatialDROW), %d windows of %d x %d-pt scans" % (B, T, N),
            "ms_per_call": ms, "scans_per_s": B / (ms * 1e-3), "data": "synthetic scans, random-init weights",
            "roofline": {"bound": "mfma", "kernel": "conv3_kernel<4> (float32 MFMA implicit GEMM)", "achieved": ach,
                         "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None}}


def bench_band_corr(ops, dev):
    """A9 at the Prototype shape: C=256 channels, n=57 positions, kernel 3, max displacement 5."""
    import torch
    B, C, n, D = 4096, 256, 57, 11
    g = torch.Generator(device=dev).manual_seed(9)
    f1 = torch.randn((B, C, n), device=dev, generator=g)
    f2 = torch.randn((B, C, n), device=dev, generator=g)
    out = torch.empty((B, D, n), dtype=torch.float32, device=dev)
    ms = _time_kernel(torch, lambda: ops.band_correlation(f1, f2, 3, 5, out=out), 20)
    per = 2 * C * n * 4 + D * n * 4
    ach = per * B / (ms * 1e-3) / 1e9
    return {"workload": "band correlation C=256 n=57 k=3 maxdisp=5, batch %d" % B, "ms_per_call": ms,
            "mfma_tflops": B * 2.0 * 64 * 64 * C / (ms * 1e-3) / 1e12,
            "roofline": {"bound": "hbm", "kernel": "band_corr_small_kernel<3> (float32 MFMA 32x32x2 Gram block)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None}}


if __name__ == "__main__":
    main()
