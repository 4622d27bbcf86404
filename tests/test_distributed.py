"""N > 1 path on CPU: world_size-2 gloo processes (torch.multiprocessing spawn is
not used -- plain subprocesses with the torchrun environment, 127.0.0.1)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np, torch
    sys.path.insert(0, os.environ["POF_REPO"])
    sys.path.insert(0, os.path.join(os.environ["POF_REPO"], "planar_optical_flow_amd"))
    from planar_optical_flow_amd import dist as pd
    from src.model.get_model import get_model
    from src.pipeline.optim import Optim
    from torch.nn.utils import clip_grad_norm_

    rank, world, dev = pd.init_distributed("gloo")
    assert world == 2 and dev.type == "cpu"
    torch.manual_seed(7)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0})
    pd.broadcast_parameters(model)
    model.eval()   # BatchNorm uses running statistics: the batch split must not change the math
    for p in model.parameters():
        p.requires_grad_(True)
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.normal(0, 0.3, (16, 64, 3))).float()
    Y = torch.from_numpy(rng.normal(0, 0.3, (16, 3))).float()
    lo, hi = pd.shard_range(16)
    optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "lr0": 1e-3, "epoch1": 5, "lr1": 1e-4}})
    red = pd.GradientAllReduce(model)
    losses = []
    for step in range(3):
        optim.zero_grad(); optim.set_lr(0)
        pred = model(X[lo:hi])
        # sum-reduced loss over the shard, scaled so that the rank average is the global mean
        loss = model.loss_fn(pred, Y[lo:hi])
        loss.backward()
        red()
        clip_grad_norm_(model.parameters(), 1.0)
        optim.step()
        t = torch.tensor([loss.item()]); torch.distributed.all_reduce(t); losses.append(t.item() / world)
    if rank == 0:
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        print("RESULT " + json.dumps({"losses": losses, "wsum": float(flat.abs().sum()), "shard": [lo, hi]}))
    torch.distributed.destroy_process_group()
''')


WORKER_TRAIN = textwrap.dedent('''
    import os, sys, json
    import numpy as np, torch
    sys.path.insert(0, os.environ["POF_REPO"])
    sys.path.insert(0, os.path.join(os.environ["POF_REPO"], "planar_optical_flow_amd"))
    from planar_optical_flow_amd import dist as pd
    from src.model.get_model import get_model
    from src.pipeline.optim import Optim
    from torch.nn.utils import clip_grad_norm_

    rank, world, dev = pd.init_distributed("gloo")
    torch.manual_seed(7)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0})
    pd.broadcast_parameters(model)
    keys_before = list(model.state_dict().keys())
    pd.convert_sync_batchnorm(model)
    assert list(model.state_dict().keys()) == keys_before          # checkpoint ABI untouched
    model.train()                                                  # batch statistics: the cross-sample coupling
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.normal(0, 0.3, (16, 64, 3))).float()
    Y = torch.from_numpy(rng.normal(0, 0.3, (16, 3))).float()
    lo, hi = pd.shard_range(16)
    optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "lr0": 1e-3, "epoch1": 5, "lr1": 1e-4}})
    red = pd.GradientAllReduce(model)
    losses = []
    for step in range(3):
        optim.zero_grad(); optim.set_lr(0)
        loss = model.loss_fn(model(X[lo:hi]), Y[lo:hi])
        loss.backward()
        red()
        clip_grad_norm_(model.parameters(), 1.0)
        optim.step()
        t = torch.tensor([loss.item()]); torch.distributed.all_reduce(t); losses.append(t.item() / world)
    # the stop flag of the training loop: one rank's signal stops every rank
    assert pd.any_rank(rank == 1) is True and pd.any_rank(False) is False
    if rank == 0:
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        rm = torch.cat([b.detach().reshape(-1).float() for n, b in model.named_buffers() if "running" in n])
        print("RESULT " + json.dumps({"losses": losses, "wsum": float(flat.abs().sum()), "bsum": float(rm.abs().sum())}))
    torch.distributed.destroy_process_group()
''')


def _single_process_train_reference():
    sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
    import torch
    from torch.nn.utils import clip_grad_norm_
    from src.model.get_model import get_model
    from src.pipeline.optim import Optim
    torch.manual_seed(7)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0})
    model.train()
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.normal(0, 0.3, (16, 64, 3))).float()
    Y = torch.from_numpy(rng.normal(0, 0.3, (16, 3))).float()
    optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "lr0": 1e-3, "epoch1": 5, "lr1": 1e-4}})
    losses = []
    for _ in range(3):
        optim.zero_grad()
        optim.set_lr(0)
        pred = model(X)                                            # ONE global batch: BatchNorm over all 16 samples
        loss = 0.5 * (model.loss_fn(pred[:8], Y[:8]) + model.loss_fn(pred[8:], Y[8:]))
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        optim.step()
        losses.append(loss.item())
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    rm = torch.cat([b.detach().reshape(-1).float() for n, b in model.named_buffers() if "running" in n])
    return losses, float(flat.abs().sum()), float(rm.abs().sum())


def _run_two_ranks(script):
    port = 29000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), POF_REPO=REPO, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")][0]
    return json.loads(line[7:])


def test_two_rank_train_mode_sync_batchnorm_matches_global_batch(tmp_path):
    """BatchNorm is the one cross-sample coupling of the box head.  With dist.SyncBatchNorm1d the
    two-rank TRAIN-mode trajectory (8 + 8 samples) equals the single-process trajectory on the 16-sample global
    batch: losses, weights after three Adam steps and the running statistics."""
    script = tmp_path / "worker_train.py"
    script.write_text(WORKER_TRAIN)
    got = _run_two_ranks(script)
    want_losses, want_wsum, want_bsum = _single_process_train_reference()
    np.testing.assert_allclose(got["losses"], want_losses, rtol=1e-5)
    np.testing.assert_allclose(got["wsum"], want_wsum, rtol=1e-6)
    np.testing.assert_allclose(got["bsum"], want_bsum, rtol=1e-5)


def _single_process_reference():
    sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))
    import torch
    from torch.nn.utils import clip_grad_norm_
    from src.model.get_model import get_model
    from src.pipeline.optim import Optim
    torch.manual_seed(7)
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0})
    model.eval()
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.normal(0, 0.3, (16, 64, 3))).float()
    Y = torch.from_numpy(rng.normal(0, 0.3, (16, 3))).float()
    optim = Optim(model, {"scheduler_kwargs": {"epoch0": 0, "lr0": 1e-3, "epoch1": 5, "lr1": 1e-4}})
    losses = []
    for _ in range(3):
        optim.zero_grad()
        optim.set_lr(0)
        # global-batch loss = mean of the two equal-size shard losses
        loss = 0.5 * (model.loss_fn(model(X[:8]), Y[:8]) + model.loss_fn(model(X[8:]), Y[8:]))
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        optim.step()
        losses.append(loss.item())
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    return losses, float(flat.abs().sum())


def test_two_rank_gloo_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), POF_REPO=REPO, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")][0]
    got = json.loads(line[7:])
    assert got["shard"] == [0, 8]
    want_losses, want_wsum = _single_process_reference()
    np.testing.assert_allclose(got["losses"], want_losses, rtol=1e-5)
    np.testing.assert_allclose(got["wsum"], want_wsum, rtol=1e-6)


def test_shard_range_partitions():
    from planar_optical_flow_amd.dist import shard_range
    for n, w in ((4096, 8), (10, 3), (5, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
