"""hipGraph-captured optimisation step (graph_step.GraphedTrainStep) against the eager step of the pipeline."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "planar_optical_flow_amd"))

_SCHED = {"scheduler_kwargs": {"epoch0": 0, "epoch1": 10, "lr0": 1e-3, "lr1": 1e-5}}


def _box_model(seed, dropout=0.0):
    from src.model.get_model import get_model
    torch.manual_seed(seed)
    return get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": dropout})


def _assert_same_training_state(a, b, lr_sum=6e-3):
    """Two correct runs of the same Adam steps do not end bit-identical on this device (atomics in the library's
    backward kernels), and Adam normalises gradients that are zero up to rounding -- the biases in front of a
    BatchNorm -- to +-lr per step.  So: every parameter within the total step budget of the other run's, the
    statistics close, and the loss of both models on a fresh batch (train-mode BatchNorm cancels those biases)
    equal to 0.5 %."""
    for (n, x), y in zip(a.state_dict().items(), b.state_dict().values()):
        if n.endswith("num_batches_tracked"):
            assert torch.equal(x, y), n
        else:
            assert float((x.float() - y.float()).abs().max()) <= 2 * lr_sum + 0.02 * float(x.float().abs().max()), n
    g = torch.Generator(device="cuda").manual_seed(99)
    batch = {"input": torch.randn(64, 64, 3, device="cuda", generator=g) * 0.3,
             "target": torch.randn(64, 3, device="cuda", generator=g) * 0.3}
    with torch.no_grad():
        la, lb = float(a.model_fn(a.train(), batch)[0]), float(b.model_fn(b.train(), batch)[0])
    assert lb == pytest.approx(la, rel=5e-3)


def test_optim_set_lr_fills_a_capturable_learning_rate():
    from src.pipeline.optim import Optim, exp_decay_lr
    model = _box_model(1)
    opt = Optim(model, _SCHED)
    opt.set_lr(3.5)
    assert opt.get_lr() == pytest.approx(exp_decay_lr(3.5, **_SCHED["scheduler_kwargs"]))
    inner = opt.make_capturable()
    lr_t = inner.param_groups[0]["lr"]
    assert torch.is_tensor(lr_t) and all(g["capturable"] and g["lr"] is lr_t for g in inner.param_groups)
    opt.set_lr(7.25)
    want = exp_decay_lr(7.25, **_SCHED["scheduler_kwargs"])
    assert float(lr_t) == pytest.approx(want, rel=1e-6) and opt.get_lr() == pytest.approx(want)


@pytest.mark.gpu
@pytest.mark.parametrize("clip", [0.0, 0.5])
def test_graphed_step_equals_eager_step(clip):
    """Five optimisation steps of the box head on five different batches: graph replay against the eager sequence
    zero_grad / model_fn / backward / clip / Adam(amsgrad) with the pipeline's per-batch learning-rate schedule."""
    from planar_optical_flow_amd.graph_step import GraphedTrainStep
    from src.pipeline.optim import Optim
    eager, graphed = _box_model(11).cuda().train(), _box_model(11).cuda().train()
    graphed.load_state_dict(eager.state_dict())
    eo, go = Optim(eager, _SCHED), Optim(graphed, _SCHED)
    g = torch.Generator(device="cuda").manual_seed(2)
    batches = [{"input": torch.randn(64, 64, 3, device="cuda", generator=g) * 0.3,
                "target": torch.randn(64, 3, device="cuda", generator=g) * 0.3} for _ in range(5)]
    step = GraphedTrainStep(graphed, go.make_capturable(), batches[0], grad_norm_clip=clip)
    # construction warms up with real steps and must put everything back
    for a, b in zip(eager.state_dict().values(), graphed.state_dict().values()):
        assert torch.equal(a, b)
    losses = []
    for i, batch in enumerate(batches):
        eo.zero_grad()
        eo.set_lr(i * 0.7)
        loss = eager.model_fn(eager, batch)[0]
        loss.backward()
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(eager.parameters(), clip)
        eo.step()
        go.set_lr(i * 0.7)
        gl = step(batch)
        losses.append((float(loss.detach()), float(gl.detach())))
    for a, b in losses:
        assert b == pytest.approx(a, rel=5e-3, abs=1e-6)
    _assert_same_training_state(eager, graphed)
    # parameters the loss does not reach are skipped by both optimisers
    unused = [n for n, p in graphed.named_parameters() if p.grad is None]
    assert unused == [n for n, p in eager.named_parameters() if p.grad is None]


@pytest.mark.gpu
def test_graphed_step_with_dropout_trains():
    """Dropout draws fresh masks on every replay (the generator is advanced by the graph): the loss on a fixed
    batch keeps changing from step to step and goes down."""
    from planar_optical_flow_amd.graph_step import GraphedTrainStep
    from src.pipeline.optim import Optim
    model = _box_model(5, dropout=0.3).cuda().train()
    opt = Optim(model, _SCHED)
    g = torch.Generator(device="cuda").manual_seed(3)
    batch = {"input": torch.randn(128, 64, 3, device="cuda", generator=g) * 0.3,
             "target": torch.randn(128, 3, device="cuda", generator=g) * 0.3}
    step = GraphedTrainStep(model, opt.make_capturable(), batch)
    opt.set_lr(0)
    losses = [float(step(batch)) for _ in range(60)]
    assert len(set(losses)) > 50
    assert np.mean(losses[-10:]) < 0.7 * np.mean(losses[:10])


@pytest.mark.gpu
def test_trainer_graph_step_matches_eager_trainer(tmp_path):
    """The pipeline Trainer with cfg graph_step: same parameters after an epoch as the eager Trainer (dropout 0)."""
    from src.pipeline.optim import Optim
    from src.pipeline.trainer import Trainer

    class _Log:
        def __getattr__(self, name):
            return lambda *a, **k: None

    g = torch.Generator(device="cuda").manual_seed(8)
    loader = [{"input": torch.randn(32, 64, 3, device="cuda", generator=g) * 0.3,
               "target": torch.randn(32, 3, device="cuda", generator=g) * 0.3} for _ in range(4)]
    loader.append({"input": loader[0]["input"][:7].clone(), "target": loader[0]["target"][:7].clone()})  # short last batch
    out = []
    for graph in (False, True):
        model = _box_model(21).cuda()
        cfg = {"grad_norm_clip": 0.0, "ckpt_interval": 100, "eval_interval": 100, "epoch": 2, "graph_step": graph}
        tr = Trainer(_Log(), Optim(model, _SCHED), cfg)
        assert tr.train(model, loader) == 0
        out.append(model)
    _assert_same_training_state(*out)


@pytest.mark.gpu
def test_graphed_step_refuses_freed_gradient_buffers():
    """The graph owns raw addresses: after zero_grad(set_to_none=True) a replay would write into freed memory."""
    from planar_optical_flow_amd.graph_step import GraphedTrainStep
    from src.pipeline.optim import Optim
    model = _box_model(7).cuda().train()
    opt = Optim(model, _SCHED)
    batch = {"input": torch.randn(16, 64, 3, device="cuda"), "target": torch.randn(16, 3, device="cuda")}
    step = GraphedTrainStep(model, opt.make_capturable(), batch)
    step(batch)
    opt.zero_grad(set_to_none=False)
    step(batch)                                   # buffers kept: fine
    opt.zero_grad()                               # default: gradients set to None
    with pytest.raises(RuntimeError, match="moved since the capture"):
        step(batch)


@pytest.mark.gpu
def test_capture_uses_deterministic_library_solvers_and_restores_the_flag():
    """The warm-up and the capture run with deterministic library solvers (an MIOpen convolution-backward solver picked
    by its default, timing-based choice replayed wrong gradients in about one process in ten:
    profiles/r3_graph_capture_miopen.txt); the caller's setting is put back, and the switch can be turned off."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planar_optical_flow_amd"))
    from src.model.get_model import get_model
    from planar_optical_flow_amd.graph_step import GraphedTrainStep, make_capturable
    seen = []
    model = get_model({"type": "box_reg", "input_dim": 3, "target_dim": 3, "dropout": 0.0}).cuda()
    model.backbone.hip_train = False                     # library convolutions in the step
    hook = model.backbone.conv1[0].register_forward_hook(lambda *a: seen.append(torch.backends.cudnn.deterministic))
    batch = {"input": torch.randn(8, 64, 3, device="cuda"), "target": torch.randn(8, 3, device="cuda")}
    for flag in (False, True):
        torch.backends.cudnn.deterministic = flag
        for det_lib in (True, False):
            seen.clear()
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            make_capturable(opt)
            GraphedTrainStep(model, opt, batch, warmup=1, deterministic_library=det_lib)
            assert seen and all(v == (det_lib or flag) for v in seen), (flag, det_lib, seen)
            assert torch.backends.cudnn.deterministic == flag
    torch.backends.cudnn.deterministic = False
    hook.remove()

