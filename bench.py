#!/usr/bin/env python3
"""Throughput of the planar-flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload = BASELINE.json configs[1]: batch 4096 synthetic 450-point
scan pairs, flow-only (no detector head): one step = ONE launch of
pof_scan_preprocess over one batch per rank (scan -> xy -> rigid-motion flow ->
canonical frame, detection association, regression target, exclude mask).
Inputs are resident in HBM before the timed region; a ring of distinct batches
larger than the 256 MiB Infinity Cache is cycled so the stream really comes from
HBM.  Batches shard over ranks with no data-path collective.  `value` is the weak
form (every rank processes its own 4096 scans per step); for N > 1 the line also
carries SURVEY 8(e)'s strong split (the 4096-scan batch cut into 4096 / N per rank)
as `strong_scaling`.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (one fresh child process per GPU, before anything here touches HIP) and
relays rank 0's JSON line; under torch.distributed.run the ranks are taken from
the environment as usual.  The K-step region is timed `--repeats` times (default 5),
each bracketed by barrier + synchronize and maximised over ranks; the line reports
the median (and min / max).

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
    ap.add_argument("--ring", type=int, default=16,
                    help="distinct resident batches cycled through (>= 2 x slots: a launch streams `slots` of them "
                         "and evaluates the params of the next `slots`)")
    ap.add_argument("--dets-per-sample", type=int, default=-1,
                    help="ablation: exactly this many detections per sample instead of the synthetic 0..6 legs")
    ap.add_argument("--graph-collectives", action="store_true",
                    help="N > 1: also time the box-head step as ONE hipGraph with the RCCL collectives captured inside "
                         "(validated on a one-rank group in the GPU suite; opt-in so that a first multi-rank run "
                         "cannot stall the headline line)")
    ap.add_argument("--no-host-fed", action="store_true",
                    help="skip the host-fed extra (profile passes: its one-batch launches share the headline kernel's name)")
    ap.add_argument("--no-single", action="store_true",
                    help="skip the one-batch-per-launch and float64 legs (PMC passes: one launch shape per kernel name)")
    ap.add_argument("--no-params", action="store_true",
                    help="ablation: the timed launches do not evaluate the next slots' params (they are constant)")
    ap.add_argument("--launch", choices=("prepared", "graph", "eager"), default="prepared",
                    help="how the launches of the K-step region are issued: marshalled once (ops.PreparedScanLaunch) and issued "
                         "directly, one ctypes call each (default: 0.5 us per step less than a graph replay in a 20-step "
                         "region); captured as one hipGraph; or through the checked per-call marshalling")
    ap.add_argument("--no-graph", action="store_true", help="alias of --launch eager")
    ap.add_argument("--no-arena", dest="arena", action="store_false",
                    help="allocate every ring tensor on its own instead of carving the ring out of one allocation")
    ap.add_argument("--slots", type=int, default=8,
                    help="ring slots (batches) handed to ONE launch of pof_scan_preprocess_multi; a step is still one "
                         "batch: K steps = ceil(K / slots) launches (1: one launch per batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-model", action="store_true",
                    help="skip the DR-SPAAM forward extra (PMC passes: keeps the per-kernel averages per shape)")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training-step rows (box head incl. its hipGraph replay, detector): the rocprofv3 "
                         "--pmc passes leave them out -- counter collection aborts the queue on the captured step")
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of the K-step region (median reported)")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes from this one even for --gpus 1")
    a = ap.parse_args()
    if a.no_graph:
        a.launch = "eager"
    return a


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


def host_cores():
    """(physical cores per lscpu, CPUs this process may use: affinity mask capped by the cgroup CPU quota)."""
    import subprocess
    phys = None
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = dict((l.split(":", 1)[0].strip(), l.split(":", 1)[1].strip()) for l in txt.splitlines() if ":" in l)
        phys = int(kv["Core(s) per socket"]) * int(kv["Socket(s)"])
        model = kv.get("Model name", "?")
    except Exception:  # noqa: BLE001
        model = "?"
    usable = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = min(usable, max(1, int(quota) // int(period)))
    except Exception:  # noqa: BLE001
        pass
    return phys or usable, usable, model


def cpu_baseline(sb, budget_s=20.0):
    """the NumPy oracle in the reference's execution shape (one __getitem__-style call per sample +
    collate), (i) one process, (ii) a pool of n processes, OMP_NUM_THREADS = 1 per worker, warm-up one batch,
    >= 3 repeats, median.  n = the physical cores of the host, capped by what this box lets the process use (the
    1-GPU boxes of this pool run under a cgroup quota of 16 CPUs of the 128-core host; a larger pool would only
    time the throttle)."""
    import multiprocessing as mp
    phys, usable, model = host_cores()
    cores = max(1, min(phys, usable))
    per = 192
    n = min(per * cores, len(sb.scans))
    chunks = np.array_split(np.arange(n), cores)
    jobs = [(sb.scans[c, -1], sb.odom0[c], sb.odom1[c], [sb.dets[i]["wp"] for i in c]) for c in chunks]
    t_all = time.perf_counter()
    # (i) single process, same per-sample calls
    cpu_sample_worker(jobs[0])                              # warm-up
    single = []
    for _ in range(3):
        t0 = time.perf_counter()
        cpu_sample_worker(jobs[0])
        single.append(len(chunks[0]) / (time.perf_counter() - t0))
    # (ii) pool
    rates = []
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(cpu_sample_worker, jobs)                   # warm-up: one batch
        while len(rates) < 3 or (time.perf_counter() - t_all < budget_s and len(rates) < 9):
            t0 = time.perf_counter()
            pool.map(cpu_sample_worker, jobs)
            rates.append(n / (time.perf_counter() - t0))
    return {"value": float(np.median(rates)), "unit": "scans/s", "cores": cores, "kind": "port",
            "single_process_scans_per_s": float(np.median(single)),
            "host": {"model": model, "physical_cores": phys, "usable_cpus": usable},
            "repeats": len(rates), "min": float(min(rates)), "max": float(max(rates)),
            "sample": "%d repeats (median) of %d scans of the headline workload after one warm-up batch: per-sample "
                      "NumPy oracle calls + collate in a %d-process pool, OMP_NUM_THREADS=1 (reference DataLoader "
                      "shape); the single-process figure is 3 repeats of %d scans"
                      % (len(rates), n, cores, len(chunks[0]))}


def launch_ranks(a):
    """--gpus N without a launcher: start N rank processes (fresh interpreters; this parent never touches HIP),
    relay rank 0's stdout, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread; the parent watches all ranks: when one of them dies, the others would
    # sit in the rendezvous or in the next collective until its timeout -- they are stopped (exactly these PIDs) instead
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    rcs = [p.wait() for p in procs]
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if failed or any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        return 1
    return 0


def main():
    a = parse()
    if (a.gpus > 1 or a.spawn) and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, a.gpus)

    from planar_optical_flow_amd import synth
    B, N = a.batch, N_PTS
    sb = synth.make_batch(seed=2 + 1000 * rank, B=B, T=2, N=N)
    if a.dets_per_sample >= 0:
        rs = np.random.default_rng(99 + rank)
        for b in range(B):
            d = np.stack([rs.uniform(1.0, 10.0, a.dets_per_sample),
                          rs.uniform(sb.phi[0], sb.phi[-1], a.dets_per_sample)], axis=1)
            sb.dets[b] = {"wc": np.zeros((0, 2)), "wa": np.zeros((0, 2)), "wp": d}
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
    tab = ops.phi_table(device=dev)
    want = ("flow", "target_cls", "target_reg", "exclude_mask")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(sbm, slots, out_dtype=torch.float32):
        """Ring of resident batches; K steps = ceil(K / slots) launches of pof_scan_preprocess_multi, each streaming
        up to `slots` ring slots and evaluating the params of as many slots further round the ring; the K-step
        region (and the W warm-up steps) captured as one hipGraph on the current stream.  W warm-up steps, then
        `repeats` timed regions of EXACTLY K steps, each bracketed by barrier + synchronize, wall time maximised
        over ranks.  -> dict(wall_s [repeats], dev_ms [repeats], ring)"""
        Bm = len(sbm.scans)
        offs, rphi, _ = sbm.det_csr()
        S = max(1, min(slots, a.ring // 2, ops.SCAN_MAX_SLOTS))
        # the ring's slots are carved out of ONE allocation (what a loader's ring buffer is): a single large mapping
        # gets larger translation fragments than a few hundred separate tensors
        esz = torch.empty((), dtype=out_dtype).element_size()
        ws_bytes = [0] * a.ring
        plan = []
        for r in range(a.ring):
            sh = (r * 509) % Bm  # distinct memory and distinct content per ring slot
            counts = np.roll(np.diff(offs), sh)
            order = np.roll(np.arange(Bm), sh)
            ro = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
            rr = np.concatenate([rphi[offs[i]:offs[i + 1]] for i in order]) if len(rphi) else rphi
            det = ops.DetCSR.from_numpy(ro, rr, np.full(len(rr), 2, np.uint8), dev)
            plan.append((sh, det))
            ws_bytes[r] = ops.scan_preprocess_workspace_bytes(Bm, int(det.rphi.shape[0]))
        Tn = sbm.scans.shape[1]

        def rup(n):
            return (n + 255) // 256 * 256
        per_slot = [rup(Bm * Tn * N * 4) + rup(Bm * N * 2 * esz) + rup(Bm * N * 8) + rup(Bm * N * 2 * 4) + rup(Bm * N * 4)
                    + rup(ws_bytes[r]) for r in range(a.ring)]
        arena = torch.empty(sum(per_slot), dtype=torch.uint8, device=dev) if a.arena else None
        cursor = [0]

        def carve(shape, dtype):
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            if arena is None:
                return torch.empty(shape, dtype=dtype, device=dev)
            t = arena[cursor[0]:cursor[0] + n].view(dtype).view(shape)
            cursor[0] += rup(n)
            return t
        ring = []
        for r in range(a.ring):
            sh, det = plan[r]
            scans = carve(tuple(sbm.scans.shape), torch.float32)
            scans.copy_(torch.from_numpy(np.roll(sbm.scans, sh, axis=0)))
            o0 = torch.from_numpy(np.roll(sbm.odom0, sh, axis=0)).to(dev)
            o1 = torch.from_numpy(np.roll(sbm.odom1, sh, axis=0)).to(dev)
            outs = {
                "flow": carve((Bm, N, 2), out_dtype),
                "target_cls": carve((Bm, N), torch.int64),
                "target_reg": carve((Bm, N, 2), torch.float32),
                "exclude_mask": carve((Bm, N), torch.float32),
            }
            ws = carve((ws_bytes[r],), torch.uint8)
            ring.append({"scans": scans, "odom0": o0, "odom1": o1, "dets": det, "out": outs, "workspace": ws})

        def trip(n):
            """n consecutive steps from ring position 0: launches of up to S slots; a launch that streams slots
            [p, p + m) evaluates the params of slots [p + S, p + S + m) -- every step's params are evaluated once,
            S steps before the step that consumes them."""
            p = 0
            while n > 0:
                m = min(S, n)
                cur = [ring[(p + j) % a.ring] for j in range(m)]
                nxt = [] if a.no_params else [ring[(p + S + j) % a.ring] for j in range(m)]
                ops.scan_preprocess_multi(cur, tab, next_batches=nxt, want=want, out_dtype=out_dtype)
                p += m
                n -= m

        # prime: params of the first S slots (with --no-params: of all of them, once)
        for r0 in range(0, a.ring if a.no_params else S, S):
            ops.scan_preprocess_multi([], tab, next_batches=ring[r0:r0 + S], want=want, out_dtype=out_dtype)

        # ---- hipGraphs: the K-step region (and the W warm-up steps) are captured whole -------------
        kGraphSteps = 512
        graphs = {}

        def graph_for(n):
            if n not in graphs:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    trip(n)
                graphs[n] = g
            return graphs[n]

        graph = None
        if a.launch == "graph":
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                trip(a.ring)                         # warm-up off the capture: lazy initialisations
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for k in (a.warmup, a.steps):
                full, rem = divmod(k, kGraphSteps)
                if full:
                    graph_for(kGraphSteps)
                if rem:
                    graph_for(rem)
            graph = True

        prepared = {}

        def prepared_for(n):
            """trip(n) as marshalled launches (ops.PreparedScanLaunch): what a loader cycling through its ring keeps."""
            if n not in prepared:
                lst, p, left = [], 0, n
                while left > 0:
                    m = min(S, left)
                    cur = [ring[(p + j) % a.ring] for j in range(m)]
                    nxt = [] if a.no_params else [ring[(p + S + j) % a.ring] for j in range(m)]
                    lst.append(ops.scan_preprocess_multi(cur, tab, next_batches=nxt, want=want, out_dtype=out_dtype,
                                                         prepare=True))
                    p += m
                    left -= m
                prepared[n] = lst
            return prepared[n]

        if a.launch == "prepared":
            trip(a.ring)
            for k in (a.warmup, a.steps):
                prepared_for(k)

        def run(k):
            if a.launch == "prepared":
                for launch in prepared[k]:
                    launch()
                return
            if graph is not None:
                full, rem = divmod(k, kGraphSteps)
                for _ in range(full):
                    graphs[kGraphSteps].replay()
                if rem:
                    graphs[rem].replay()
            else:
                trip(k)

        run(a.warmup)
        walls, devs = [], []
        # the wall-clock regions hold the K steps and nothing else; the HIP-event figure comes from regions of its own
        # (two event records inside the bracket cost ~0.15 us per step of a 20-step region)
        for _ in range(max(1, a.repeats)):
            barrier()
            t0 = time.perf_counter()
            run(a.steps)
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            walls.append(dt)
        for _ in range(max(1, min(3, a.repeats))):
            barrier()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            run(a.steps)
            ev1.record()
            barrier()
            devs.append(ev0.elapsed_time(ev1))
        return {"wall_s": walls, "dev_ms": devs, "ring": ring, "graph": graph is not None, "slots": S,
                "launches": -(-a.steps // S)}

    weak = measure(sb, a.slots)
    # the same steps one batch per launch (the per-kernel duration rocprofv3 reports for THAT form is one step)
    single = measure(sb, 1) if (a.slots > 1 and not a.no_single) else None
    # the float64-output instantiation (the reference's float64 operation order for the flow: utils.py:47-48,
    # 639-662), same region shape
    f64 = f64_out = None
    if not a.no_single:
        f64 = measure(sb, a.slots, torch.float64)
        f64_out = f64["ring"][0]["out"]["flow"][:64].cpu().numpy()
        f64 = {k: v for k, v in f64.items() if k != "ring"}
    torch.cuda.empty_cache()
    dt = float(np.median(weak["wall_s"]))
    dev_ms = float(np.median(weak["dev_ms"]))
    ring = weak["ring"]
    graph = weak["graph"]
    strong = None
    if world > 1 and B % world == 0:
        # the contiguous split of ONE 4096-scan batch, 4096 / N scans per rank and step
        per = B // world
        sbg = synth.make_batch(seed=2, B=B, T=2, N=N)
        sl = slice(rank * per, (rank + 1) * per)
        import copy
        sbs = copy.copy(sbg)
        sbs.scans, sbs.odom0, sbs.odom1, sbs.dets = sbg.scans[sl], sbg.odom0[sl], sbg.odom1[sl], sbg.dets[sl]
        st = measure(sbs, a.slots)
        sdt = float(np.median(st["wall_s"]))
        strong = {"value": B * a.steps / sdt, "unit": "scans/s", "ms_per_step": sdt / a.steps * 1e3,
                  "scans_per_rank_per_step": per, "global_batch": B,
                  "parallelism": "strong: one %d-scan batch split contiguously, %d scans per rank, no collective" % (B, per)}
        del st

    # ---- parity of what was just computed (outside the timed region) ------------
    from oracle import ref_numpy as R
    outs = ring[0]["out"]
    phi = R.laser_phi()
    flow = outs["flow"][:64].cpu().numpy()
    epe = epe64 = 0.0
    for b in range(64):
        cur = sb.scans[b, -1]
        xy = np.array(R.polar_to_xy(cur, phi)).T
        ref = R.flow_to_canonical(R.displacement_from_odometry(xy, sb.odom0[b], sb.odom1[b]), phi)
        epe += float(np.linalg.norm(flow[b] - ref, axis=-1).mean()) / 64
        if f64_out is not None:
            epe64 += float(np.linalg.norm(f64_out[b] - ref, axis=-1).mean()) / 64

    # the training rows run on all ranks (they hold collectives).  They are extras: an error in one of them (the
    # same on every rank) is reported in its place instead of costing the line its headline measurement
    def guarded(fn, *args):
        try:
            return fn(*args)
        except Exception as e:  # noqa: BLE001
            import traceback
            sys.stderr.write("bench.py: %s failed on rank %d:\n%s\n" % (fn.__name__, rank, traceback.format_exc()))
            return {"error": "%s: %s" % (type(e).__name__, e)}

    box = None if (a.no_extra or a.no_train) else guarded(bench_box_head, dev, world, rank, backend, barrier,
                                                            a.graph_collectives)
    det_train = None if (a.no_extra or a.no_model or a.no_train) else guarded(bench_detector_train, ops, synth, tab, dev,
                                                                             world, rank, backend, barrier)

    result = None
    if rank == 0:
        # algorithmic bytes per scan: 4N range row in; out 8N flow f32 +
        # 8N target_cls int64 + 8N target_reg + 4N exclude mask  = 32N = 14 400 B
        bytes_per_scan = 4 * N + (8 + 8 + 8 + 4) * N
        bytes_per_scan_f64 = 4 * N + (16 + 8 + 8 + 4) * N      # float64 flow: 40N = 18 000 B
        S = weak["slots"]
        walls = weak["wall_s"]
        ms_step = dt / a.steps * 1e3
        # ONE clock: the roofline figure comes from the same wall interval as ms_per_step (barrier + synchronize on
        # both sides, graph launch and the final synchronize included); the HIP-event interval on the launch
        # stream is carried beside it
        achieved = bytes_per_scan * B / (ms_step * 1e-3) / 1e9
        ev_step = dev_ms / a.steps
        kernel = "scan_flat_kernel<float, headline cfg>"
        traffic, traffic_src = pmc_traffic(("scan_flat_kernel<float",))
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
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 flow arithmetic (float32 outputs) + f64 exact association / mask decisions; the float64-"
                     "output instantiation (the reference's float64 operation order) is `value_f64`",
            "data": "synthetic",
            "timed_repeats": len(walls),
            "ms_per_step_min": min(walls) / a.steps * 1e3,
            "ms_per_step_max": max(walls) / a.steps * 1e3,
            "config": {"workload": "BASELINE configs[1]: batch %d x 450-pt scan pairs, flow-only "
                                   "(A1-A7 fused: xy, displacement flow, canonical frame, association, "
                                   "regression target, exclude mask), float32 outputs" % B,
                       "global_batch": world * B, "ring_batches": a.ring,
                       "launch": {"prepared": "launches marshalled once, issued directly (one ctypes call each)",
                                  "graph": "hipGraph replay of the K-step region",
                                  "eager": "checked per-call marshalling"}[a.launch]
                                 + ", pof_scan_preprocess_multi: %d ring slot(s) per launch (%d launches for %d steps), "
                                   "the params of the slots %d further round the ring ride in the same launch"
                                   % (S, weak["launches"], a.steps, S),
                       "slots_per_launch": S,
                       "parallelism": "weak: batch-sharded x%d, %d scans per rank and step, no data-path collective" % (world, B),
                       "rccl_ranks": (dist.get_world_size() if world > 1 and backend == "nccl" else None),
                       "collective_backend": (backend if world > 1 else None)},
            "epe_vs_oracle_m": epe,
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_per_step": (traffic / S if traffic else None),
                         "traffic_source": traffic_src,
                         "clock": "wall (the interval of ms_per_step)",
                         "bytes_per_step": bytes_per_scan * B,
                         # K steps are ceil(K / S) launches (the last one may stream fewer slots): per-launch
                         # figures are averages over those launches
                         "bytes_per_launch": bytes_per_scan * B * a.steps / weak["launches"],
                         "launch_ms": ms_step * a.steps / weak["launches"],
                         "launches": weak["launches"], "steps_per_launch": S,
                         "events": {"ms_per_step": ev_step, "achieved": bytes_per_scan * B / (ev_step * 1e-3) / 1e9,
                                    "frac": bytes_per_scan * B / (ev_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "ms_per_step_min": min(weak["dev_ms"]) / a.steps,
                                    "ms_per_step_max": max(weak["dev_ms"]) / a.steps,
                                    "note": "HIP events on the launch stream around K-step regions of their own (the same launches; the wall regions hold no event records)"},
                         "note": "one kernel launch covers up to %d steps (ring slots): a per-kernel duration as "
                                 "rocprofv3 lists it is launch_ms = ms_per_step x steps / launches (%d x ms_per_step for "
                                 "full launches); `traffic` is per full launch" % (S, S)},
        }
        if f64 is not None:
            f_dt = float(np.median(f64["wall_s"]))
            f_ms = f_dt / a.steps * 1e3
            f_ach = bytes_per_scan_f64 * B / (f_ms * 1e-3) / 1e9
            result["value_f64"] = world * B * a.steps / f_dt
            result["roofline_f64"] = {"bound": "hbm", "kernel": "scan_flat_kernel<double, headline cfg>",
                                      "dtype": "f64 flow arithmetic in the reference's operation order, float64 flow output",
                                      "ms_per_step": f_ms, "achieved": f_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": f_ach / HBM_PEAK_GBS, "bytes_per_step": bytes_per_scan_f64 * B,
                                      "steps_per_launch": f64["slots"], "clock": "wall",
                                      "events_ms_per_step": float(np.median(f64["dev_ms"])) / a.steps,
                                      "epe_vs_oracle_m": epe64, "traffic": None}
        if single is not None:
            s_dev = float(np.median(single["dev_ms"])) / a.steps
            s_wall = float(np.median(single["wall_s"]))
            s_ach = bytes_per_scan * B / (s_wall / a.steps) / 1e9
            result["single_batch_launches"] = {"value": world * B * a.steps / s_wall, "ms_per_step": s_wall / a.steps * 1e3,
                                               "achieved": s_ach, "frac": s_ach / HBM_PEAK_GBS,
                                               "events_ms_per_step": s_dev,
                                               "note": "the same K steps as K launches of one ring slot each"}
        if strong is not None:
            result["strong_scaling"] = strong
        if box is not None:
            result["box_head_train"] = box
        if det_train is not None:
            result["detector_train"] = det_train
        if not a.no_extra and world == 1:   # per-kernel extras only on the single-GPU line
            if not a.no_host_fed:
                result["host_fed"] = bench_host_fed(ops, sb, tab, dev)
            result["cutout"] = bench_cutout(ops, synth, tab, dev, variants=not a.no_model)
            result["cutout_dense"] = bench_cutout_dense(ops, synth, dev)
            result["spatial_attention"] = bench_attention(ops, dev)
            result["band_correlation"] = bench_band_corr(ops, dev)
            result["small_kernels"] = bench_small_kernels(ops, synth, tab, dev)
            if not a.no_model:
                result["dr_spaam_forward"] = bench_dr_spaam(ops, synth, tab, dev)
                result["prototype_forward"] = guarded(bench_prototype, ops, dev)
            # PMC traffic of the same shapes, read from the committed counter passes (tools/collect_profiles.sh
            # runs this very function under rocprofv3 --pmc): not measured in this run
            for key, pref in (("cutout", ("cutout_area_kernel", "cutout_kernel<3, 7, 1,", "cutout_kernel<1, 7, 1,")),
                              ("spatial_attention", ("attn_",)), ("band_correlation", ("band_corr_",))):
                tr, src = pmc_traffic(pref)
                result[key]["roofline"]["traffic"] = tr
                result[key]["roofline"]["traffic_source"] = src
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_baseline"] = result["value"] / cpu["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_box_head(dev, world, rank, backend, barrier, graph_collectives=False, steps=30, warm=5):
    """BASELINE configs[3]: one optimisation step of the box-regression head (train_box_regression_1.yaml: PointNet
    on 64-point segments, batch 256 PER RANK, Adam with amsgrad), batch-sharded: forward + backward, ONE flat
    gradient all-reduce over RCCL (dist.GradientAllReduce, 3.8 MB) between backward and the optimiser step,
    BatchNorm statistics over the global batch (dist.SyncBatchNorm1d).  Runs on every rank (it contains the
    collectives); rank 0 reports.  Also times the gradient all-reduce on its own."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
    from planar_optical_flow_amd import dist as pdist
    from src.model.get_model import get_model
    from src.pipeline.optim import Optim
    torch.manual_seed(4)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).to(dev)
    if world > 1:
        pdist.broadcast_parameters(model)
        pdist.convert_sync_batchnorm(model)
    model.train()
    optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "epoch1": 100, "lr0": 1e-3, "lr1": 1e-6}})
    reducer = pdist.GradientAllReduce(model)
    g = torch.Generator(device=dev).manual_seed(40 + rank)
    per = 256
    x = torch.randn((per, 64, 3), device=dev, generator=g) * 0.3
    y = torch.randn((per, 3), device=dev, generator=g) * 0.3

    def step():
        optim.zero_grad()
        optim.set_lr(0)
        loss = model.loss_fn(model(x), y)
        loss.backward()
        reducer()
        optim.step()
        return loss

    for _ in range(warm):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = (time.perf_counter() - t0) / steps
    # host synchronisations in one eager step (forward, backward, SyncBatchNorm / gradient collectives, Adam):
    # sync debug mode "error" raises on any blocking call
    host_syncs = 0
    torch.cuda.set_sync_debug_mode("error")
    try:
        step()
    except RuntimeError as e:       # noqa: BLE001
        host_syncs = "at least one (%s)" % str(e).splitlines()[0][:80]
    finally:
        torch.cuda.set_sync_debug_mode("default")
    barrier()
    ar_ms = None
    if world > 1:
        for _ in range(3):
            reducer()
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            reducer()
        barrier()
        ar_ms = (time.perf_counter() - t0) / 20 * 1e3
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    graphed = None
    if world == 1 or (backend == "nccl" and graph_collectives):
        # the same step as ONE hipGraph replay (graph_step.GraphedTrainStep): ~150 small kernels, eager pacing is the
        # host's.  With more ranks the SyncBatchNorm collectives and the gradient all-reduce are nodes of the graph
        # (RCCL is stream-ordered and capturable); a host-side backend (the gloo rehearsal) cannot be captured.
        from planar_optical_flow_amd.graph_step import GraphedTrainStep
        torch.manual_seed(4)
        gmodel = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).to(dev)
        greducer = None
        if world > 1:
            pdist.broadcast_parameters(gmodel)
            pdist.convert_sync_batchnorm(gmodel)
            greducer = pdist.GradientAllReduce(gmodel)
        gmodel.train()
        goptim = Optim(gmodel, {"scheduler_kwargs": {"epoch0": 0, "epoch1": 100, "lr0": 1e-3, "lr1": 1e-6}})
        gstep = GraphedTrainStep(gmodel, goptim.make_capturable(), {"input": x, "target": y}, reducer=greducer)
        batch = {"input": x, "target": y}
        for _ in range(warm):
            goptim.set_lr(0)
            gstep(batch)
        barrier()
        t0 = time.perf_counter()
        for _ in range(10 * steps):
            goptim.set_lr(0)
            gstep(batch)
        barrier()
        gdt = (time.perf_counter() - t0) / (10 * steps)
        if world > 1:
            t = torch.tensor([gdt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gdt = float(t.item())
        lib_ms = None
        if world == 1:
            torch.manual_seed(4)
            lmodel = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.3}).to(dev)
            lmodel.backbone.hip_train = False
            lmodel.train()
            loptim = Optim(lmodel, {"scheduler_kwargs": {"epoch0": 0, "epoch1": 100, "lr0": 1e-3, "lr1": 1e-6}})
            # (MIOpen's default solver choice: what the library path costs at its fastest -- replay-safe only with
            # deterministic solvers, 4.3 ms; profiles/r3_graph_capture_miopen.txt)
            lstep = GraphedTrainStep(lmodel, loptim.make_capturable(), {"input": x, "target": y}, deterministic_library=False)
            for _ in range(warm):
                loptim.set_lr(0)
                lstep(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10 * steps):
                loptim.set_lr(0)
                lstep(batch)
            torch.cuda.synchronize()
            lib_ms = (time.perf_counter() - t0) / (10 * steps) * 1e3
        graphed = {"ms_per_step": gdt * 1e3, "samples_per_s": world * per / gdt,
                   "ms_per_step_library_modules": lib_ms,
                   "note": "zero_grad + forward + backward%s + Adam captured once, replayed per batch (inputs copied "
                           "into static buffers, learning rate a device scalar)"
                           % (" + SyncBatchNorm collectives + gradient all-reduce (RCCL nodes of the graph)" if world > 1 else "")}
    nbytes = (reducer.bucket.numel() - 1) * 4       # the last element is the agreed stop flag
    return {"workload": "BASELINE configs[3]: box-regression head training step, batch 256 per rank x %d rank(s), "
                        "64-point segments, Adam(amsgrad), one flat gradient all-reduce, global-batch BatchNorm" % world,
            "ms_per_step": dt * 1e3, "samples_per_s": world * per / dt, "per_rank_batch": per, "graphed": graphed,
            "host_syncs_per_step": host_syncs,
            "grad_allreduce_ms": ar_ms, "grad_bucket_bytes": nbytes,
            "grad_allreduce_busbw_GBps": (2.0 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9) if ar_ms else None,
            "collective_backend": backend if world > 1 else None}


def bench_detector_train(ops, synth, tab, dev, world, rank, backend, barrier, steps=10, warm=3):
    """BASELINE configs[2] in training -- "the detector's gradient step" of the north star: 8 windows of 5 scans x
    450 points PER RANK -> area cutouts (HIP) -> DR-SPAAM (SpatialDROW, reference architecture, random init) in
    training mode -> classification + regression loss -> backward -> ONE flat gradient all-reduce over RCCL ->
    Adam.  The trunk (Conv1d + BatchNorm(train) + LeakyReLU + max-pool, forward / data / weight gradients) runs on
    the HIP training kernels (torch_ops.TrunkUnitTrain), the gate on the HIP attention forward / backward; the
    embedding GEMM and the 1x1 heads are hipBLASLt through torch.  Runs on every rank; rank 0 reports."""
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    from planar_optical_flow_amd import dist as pdist
    from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW
    B, T, N, P = 8, 5, N_PTS, 56
    torch.manual_seed(5)
    model = SpatialDROW(num_scans=T, num_pts=P, alpha=0.5, window_size=11, pedestrian_only=True).to(dev)
    if world > 1:
        pdist.broadcast_parameters(model)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=True)
    reducer = pdist.GradientAllReduce(model)
    sb = synth.make_batch(seed=50 + rank, B=B, T=T, N=N)
    scans = torch.from_numpy(sb.scans).to(dev)
    g = torch.Generator(device=dev).manual_seed(60 + rank)
    tcls = (torch.rand((B, N, 1), device=dev, generator=g) < 0.1).float()
    treg = torch.randn((B, N, 2), device=dev, generator=g) * 0.3
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=P, padding_val=29.99,
              area_mode=True)

    def step():
        opt.zero_grad(set_to_none=True)
        x = ops.cutout(scans, tab, **kw)
        pred_cls, pred_reg, _ = model(x)
        loss = F.binary_cross_entropy_with_logits(pred_cls, tcls) + F.mse_loss(pred_reg, treg)
        loss.backward()
        reducer()
        opt.step()
        return loss

    for _ in range(warm):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = (time.perf_counter() - t0) / steps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"workload": "BASELINE configs[2], training: DR-SPAAM step (cutout + forward + backward + gradient "
                        "all-reduce + Adam), %d windows x %d scans x %d points per rank x %d rank(s)" % (B, T, N, world),
            "ms_per_step": dt * 1e3, "scans_per_s": world * B / dt, "per_rank_batch": B,
            "grad_bucket_bytes": (reducer.bucket.numel() - 1) * 4,
            "trunk": "HIP: conv3 forward / dgrad (conv3_kernel), wgrad (conv3_wgrad_kernel), BatchNorm(train) + "
                     "LeakyReLU + max-pool forward / backward (bn_* kernels)",
            "collective_backend": backend if world > 1 else None}


def bench_host_fed(ops, sb, tab, dev, steps=200):
    """The headline step when the boundary is handed HOST buffers (never `value`): pinned host memory, H2D copies
    of batch i+1 on a copy stream under the kernel of batch i.  PCIe-inclusive scans/s."""
    import torch
    B, N = sb.scans.shape[0], sb.scans.shape[2]
    offs, rphi, _ = sb.det_csr()
    host = {"scans": torch.from_numpy(sb.scans).pin_memory(), "o0": torch.from_numpy(sb.odom0).pin_memory(),
            "o1": torch.from_numpy(sb.odom1).pin_memory(), "offs": torch.from_numpy(offs.astype(np.int32)).pin_memory(),
            "rphi": torch.from_numpy(np.ascontiguousarray(rphi)).pin_memory(),
            "cls": torch.from_numpy(np.full(len(rphi), 2, np.uint8)).pin_memory()}
    in_bytes = sum(t.numel() * t.element_size() for t in host.values())
    slots = []
    for _ in range(2):
        d = {k: torch.empty_like(v, device=dev) for k, v in host.items()}
        outs = {"flow": torch.empty((B, N, 2), dtype=torch.float32, device=dev),
                "target_cls": torch.empty((B, N), dtype=torch.int64, device=dev),
                "target_reg": torch.empty((B, N, 2), dtype=torch.float32, device=dev),
                "exclude_mask": torch.empty((B, N), dtype=torch.float32, device=dev)}
        ws = torch.empty(ops.scan_preprocess_workspace_bytes(B, len(rphi)), dtype=torch.uint8, device=dev)
        slots.append((d, outs, ws, torch.cuda.Event(), torch.cuda.Event()))
    copy_stream = torch.cuda.Stream()
    want = ("flow", "target_cls", "target_reg", "exclude_mask")

    def upload(i):
        d, _, _, ready, free = slots[i % 2]
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(free)                 # the kernel that last read this slot has finished
            for k, v in host.items():
                d[k].copy_(v, non_blocking=True)
            ready.record(copy_stream)

    def compute(i):
        d, outs, ws, ready, free = slots[i % 2]
        cur = torch.cuda.current_stream()
        cur.wait_event(ready)
        ops.scan_preprocess(d["scans"], tab, d["o0"], d["o1"], ops.DetCSR(d["offs"], d["rphi"], d["cls"]),
                            want=want, out=outs, workspace=ws)
        free.record(cur)

    for sl in slots:
        sl[4].record(torch.cuda.current_stream())
    upload(0)
    for i in range(20):
        upload(i + 1)
        compute(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20, 20 + steps):
        upload(i + 1)
        compute(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"host_fed_scans_per_s": B / dt, "us_per_step": dt * 1e6, "h2d_GBps": in_bytes / dt / 1e9,
            "input_bytes_per_step": in_bytes,
            "note": "pinned host buffers, copies of batch i+1 overlap the kernel of batch i; reported beside "
                    "`value`, which is device-resident"}


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


def bench_cutout_dense(ops, synth, dev):
    """A8 at BASELINE configs[4] shape: N = 3600 points (0.1 deg), T = 11, P = 56, float16 output."""
    import torch
    B, T, N, P = 64, 11, 3600, 56
    tab = ops.phi_table(np.radians(0.1), N, device=dev)
    sb = synth.make_batch(seed=5, B=B, T=T, N=N, angle_inc=np.radians(0.1))
    scans = torch.from_numpy(sb.scans).to(dev)
    kw = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=P,
              padding_val=29.99, area_mode=True)
    out16 = torch.empty((B, N, T, P), dtype=torch.float16, device=dev)
    out32 = torch.empty((B, N, T, P), dtype=torch.float32, device=dev)
    ms16 = _time_kernel(torch, lambda: ops.cutout(scans, tab, out=out16, out_dtype=torch.float16, **kw), 5)
    ms32 = _time_kernel(torch, lambda: ops.cutout(scans, tab, out=out32, **kw), 5)
    per16 = T * N * 4 + N * T * P * 2  # 4 593 600 B
    ach = per16 * B / (ms16 * 1e-3) / 1e9
    return {"workload": "cutout N=3600 T=11 P=56 area_mode float16 out, batch %d (BASELINE configs[4])" % B,
            "ms_per_call": ms16, "ms_per_call_float32_out": ms32, "samples_per_s": B / (ms16 * 1e-3),
            "roofline": {"bound": "hbm", "kernel": "cutout_area_kernel + cutout_kernel (span staging)", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}}


def bench_small_kernels(ops, synth, tab, dev):
    """Timing + algorithmic bytes of the hot-path kernels that have no roofline line of their own: A11 NMS, A12 flow
    errors, A13 segments + least squares, A16 rotated IoU (batched evaluation shape), N1 window gather, N3 segment
    preparation.  Latency-bound helpers: `GBps` is algorithmic bytes / time, for orientation only."""
    import torch
    rows = {}
    g = torch.Generator(device=dev).manual_seed(77)

    def row(name, ms, nbytes, note):
        rows[name] = {"ms_per_call": ms, "algorithmic_bytes": int(nbytes), "GBps": nbytes / (ms * 1e-3) / 1e9,
                      "note": note}

    B, N = 4096, N_PTS
    sb = synth.make_batch(seed=21, B=B, T=1, N=N, dropout=0.0)
    ranges = torch.from_numpy(sb.scans[:, 0].copy()).to(dev)
    # A13: one wave per segment
    ms = _time_kernel(torch, lambda: ops.segment_features(ranges, tab), 10)
    _, num, _ = ops.segment_features(ranges, tab)
    S = int(num.sum().item())
    row("segment_kernel", ms, B * 8 * N + 128 * S,
        "A13: %d scans, %d segments; read 4N + write 4N seg ids per scan + 128 B per segment" % (B, S))
    nxt = torch.roll(ranges, -1, 0).contiguous()
    ms = _time_kernel(torch, lambda: ops.segment_features_reference(ranges, tab, nxt), 10)
    row("segment_kernel_reference_rows", ms, B * 12 * N + 120 * S,
        "A13 in the reference's 15-column form (median selection, mean speed vs the next scan)")
    # A11
    Bn = 1024
    pc = torch.rand((Bn, N), dtype=torch.float64, device=dev, generator=g)
    pr = torch.randn((Bn, N, 2), dtype=torch.float64, device=dev, generator=g) * 0.3
    ms = _time_kernel(torch, lambda: ops.nms_predicted_center(ranges[:Bn].contiguous(), tab, pc, pr), 10)
    row("nms_kernel", ms, Bn * N * (4 + 8 + 16 + 16 + 8 + 4),
        "A11: %d scans x %d points per launch (one workgroup per scan: LDS bitonic sort + greedy suppression)" % (Bn, N))
    # A12
    pf = torch.randn((B, N, 2), device=dev, generator=g)
    tf = torch.randn((B, N, 2), device=dev, generator=g)
    mk = (torch.rand((B, N), device=dev, generator=g) < 0.8).float()
    ms = _time_kernel(torch, lambda: ops.flow_errors(pf, tf, mk), 20)
    row("flow_errors_kernel", ms, B * N * 20, "A12: EPE / AAE sums, %d x %d points, 20 B per point" % (B, N))
    # A16: batched evaluation shape: 256 groups x 1 prediction x <= 32 neighbour boxes
    G, K = 256, 32
    bx = torch.rand((G, 1, 5), device=dev, generator=g) * 2
    qx = torch.rand((G, K, 5), device=dev, generator=g) * 2
    bx[..., 2:4] += 0.3
    qx[..., 2:4] += 0.3
    kv = torch.randint(1, K + 1, (G,), device=dev, generator=g, dtype=torch.int32)
    ms = _time_kernel(torch, lambda: ops.rotate_iou(bx, qx, k_valid=kv), 20)
    row("rotate_iou_kernel", ms, G * (5 + 5 * K + K) * 4,
        "A16: %d groups x 1 box x <= %d query boxes in ONE launch (the reference launches once per sample)" % (G, K))
    # N1: window gather from the device-resident scan store
    Sn, T = 20000, 5
    store = torch.rand((Sn, N), device=dev, generator=g)
    first = torch.zeros(B, dtype=torch.int32, device=dev)
    idx = torch.randint(0, Sn, (B,), device=dev, generator=g, dtype=torch.int32)
    ms = _time_kernel(torch, lambda: ops.gather_windows(store, first, idx, T), 20)
    row("gather_windows_kernel", ms, B * (T + 1) * N * 8,
        "N1: %d windows of %d + 1 rows x %d points, read + write" % (B, T, N))
    # N4: polar TSDF grid (pure write stream)
    Bp, Tp, Rp = 2048, 5, 31
    sbp = synth.make_batch(seed=3, B=Bp, T=Tp, N=N)
    scp = torch.from_numpy(sbp.scans).to(dev)
    outp = torch.empty((Bp, Tp, Rp, N), dtype=torch.float32, device=dev)
    ms = _time_kernel(torch, lambda: ops.polar_grid(scp, out=outp), 20)
    row("polar_grid_flat_kernel", ms, Bp * Tp * N * 4 * (Rp + 1),
        "N4: %d x %d scan rows -> %d range bins each; aligned 16-byte non-temporal stores over the flat output" % (Bp, Tp, Rp))
    # N2 training tail: BatchNorm(train) + LeakyReLU + max-pool of one trunk unit at the reference's training
    # batch (8 scans x 450 cutouts x 5 scans = 18 000 sequences, 128 channels x 48 points), forward and backward
    Sb, Cb, Lb = 18000, 128, 48
    yb = torch.randn((Sb, Cb, Lb), device=dev, generator=g)
    gam = torch.rand(Cb, device=dev, generator=g) + 0.5
    bet = torch.rand(Cb, device=dev, generator=g) - 0.5
    rm, rv = torch.zeros(Cb, device=dev), torch.ones(Cb, device=dev)
    zb, mu, istd = ops.bn_lrelu_pool_forward(yb, gam, bet, rm, rv, pool=True)
    dzb = torch.randn(zb.shape, device=dev, generator=g)
    el = Sb * Cb * Lb
    ms = _time_kernel(torch, lambda: ops.bn_lrelu_pool_forward(yb, gam, bet, rm, rv, pool=True), 10)
    row("bn_lrelu_pool_forward", ms, el * (4 + 4 + 2),
        "N2 training tail, pooled unit: bn_stats (read y) + bn_apply (read y, write z/2): 10 B per element of y "
        "[%d x %d x %d]" % (Sb, Cb, Lb))
    ms = _time_kernel(torch, lambda: ops.bn_lrelu_pool_backward(yb, dzb, gam, bet, mu, istd, pool=True), 10)
    row("bn_lrelu_pool_backward", ms, el * (4 + 2 + 4 + 2 + 4),
        "N2 training tail backward: bn_bwd_reduce (read y, dz/2) + bn_bwd_dgrad (read y, dz/2, write dy): 16 B per "
        "element")
    del yb, zb, dzb
    # N3: all detections of a frame in one launch
    Np, Sd = 4000, 64
    pts = torch.rand((Np, 2), dtype=torch.float64, device=dev, generator=g) * 10
    ctr = pts[torch.randint(0, Np, (Sd,), device=dev, generator=g)].contiguous()
    ori = torch.zeros(Sd, dtype=torch.float64, device=dev)
    ms = _time_kernel(torch, lambda: ops.segment_inputs(pts, ctr, ori), 20)
    row("segment_inputs_kernel", ms, Np * 16 + Sd * (16 + 64 * 3 * 4),
        "N3: %d points, %d detections: radius query + fixed-size resampling in one launch" % (Np, Sd))
    return rows


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


def bench_prototype(ops, dev):
    """N2: the Prototype flow network (three stride-2 encoders on both scans of a pair, banded correlation, two
    decoders, point-wise head) in inference, random-init weights of the reference architecture: every unit on
    pof_conv1d_bn_lrelu after fuse_for_inference(), against the same modules through MIOpen."""
    import torch
    from planar_optical_flow_amd.src.depracted.model.prototype import Prototype
    B, n = 4096, N_PTS
    torch.manual_seed(7)
    model = Prototype(in_channel=1, max_displacement=5).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(11)
    s1 = torch.randn((B, n, 1), device=dev, generator=g)
    s2 = torch.randn((B, n, 1), device=dev, generator=g)

    def step():
        with torch.no_grad():
            return model(s1, s2)
    ms_lib = _time_kernel(torch, step, 3, warm=2)
    model.fuse_for_inference()
    ms = _time_kernel(torch, step, 5, warm=2)
    l0, l1, l2 = (n + 1) // 2, ((n + 1) // 2 + 1) // 2, (((n + 1) // 2 + 1) // 2 + 1) // 2
    flops = B * 2.0 * (2 * (l0 * 3 * 1 * 64 + l1 * 3 * 64 * 128 + l2 * 3 * 128 * 256)
                       + l1 * 3 * 139 * 128 + l0 * 3 * 192 * 128 + n * 129 * 2 + 11 * l2 * 768)
    ach = flops / (ms * 1e-3) / 1e12
    # training: forward + backward of the EPE loss at 256 pairs, units as ConvUnitTrain nodes on the HIP kernels
    # against the same modules through the library
    from planar_optical_flow_amd.src.depracted.model.prototype import flow_loss
    tmodel = Prototype(in_channel=1, max_displacement=5).to(dev).train()
    tb = 256
    tgt = torch.randn((tb, n, 2), device=dev, generator=g) * 0.2

    def train_step():
        tmodel.zero_grad(set_to_none=True)
        loss, _ = flow_loss(tmodel(s1[:tb], s2[:tb]), tgt)
        loss.backward()
        return loss
    train_ms = {}
    for tag, hip in (("hip_units", True), ("library_modules", False)):
        tmodel.hip_train = hip
        train_ms[tag] = _time_kernel(torch, train_step, 5, warm=3)
    return {"workload": "Prototype forward, %d scan pairs x %d points (inference, BatchNorm folded)" % (B, n),
            "ms_per_call": ms, "pairs_per_s": B / (ms * 1e-3), "ms_per_call_library_modules": ms_lib,
            "train_forward_backward": {"pairs": tb, "ms_hip_units": train_ms["hip_units"],
                                       "ms_library_modules": train_ms["library_modules"],
                                       "note": "EPE loss, batch statistics; convolution forward / data / weight gradients "
                                               "and the BatchNorm tail on the HIP kernels vs MIOpen + ATen"},
            "data": "synthetic scans, random-init weights",
            "roofline": {"bound": "mfma", "kernel": "conv1d_kernel<CT, 3, 2 | 3, 1 | 1, 1> (float32 MFMA implicit GEMM)",
                         "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None}}


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
