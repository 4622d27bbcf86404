"""A whole optimisation step as ONE hipGraph replay.

The box-regression head of BASELINE configs[3] (PointNet on 64-point segments, batch 256) is a step of ~150 small
kernels -- forward, backward, gradient clipping, Adam -- whose eager form is paced by the host (1.86 ms per step for
1.46 ms of kernel time).  ``GraphedTrainStep`` captures

    zero_grad -> loss = model.model_fn(model, batch) -> backward [-> clip_grad_norm_] -> optimiser step

once, on static input buffers, and replays it per batch: the host cost of a step becomes two small copies and
one graph launch (measured: 1.86 -> 1.64 ms; a replay still spends ~10 us per kernel node, so what is left is the
number of kernels, not the host).  The optimiser has to be capturable (``torch.optim.Adam(..., capturable=True)``: its step
counters live on the device); the learning rate is a device scalar that ``set_lr`` fills, so the reference's
per-batch schedule (``Optim.set_lr(epoch + ratio)``, src/pipeline/trainer.py) keeps working under replay.

Multi-rank: the gradient all-reduce sits between backward and clipping (``dist.GradientAllReduce``; every
``p.grad`` is a view into its flat bucket, so the collective is the only thing added to the step).  On RCCL
(backend "nccl") collectives are stream-ordered and capturable: the all-reduce -- and the SyncBatchNorm
collectives of the forward / backward -- are nodes of the SAME graph, one replay per step on every rank.  On a
backend whose collectives run on the host (gloo: the CPU rehearsals) the step is captured as two graphs --
[zero_grad, forward, backward] and [clip, step] -- with the collective eager between them; a forward that itself
holds collectives (SyncBatchNorm) is refused there.
"""
import torch

from . import dist as pdist


def make_capturable(optimizer, lr=None):
    """Switch a freshly built torch optimiser (no state yet) to its capturable form with the learning rate as a
    device scalar.  Returns the lr tensor of the first group (all groups share one tensor)."""
    if any(len(s) for s in optimizer.state.values()):
        raise RuntimeError("make_capturable: the optimiser already holds state; build it, then call this before the "
                           "first step")
    dev = None
    for group in optimizer.param_groups:
        for p in group["params"]:
            dev = p.device
            break
        if dev is not None:
            break
    lr_t = torch.tensor(float(lr if lr is not None else optimizer.param_groups[0]["lr"]), dtype=torch.float32,
                        device=dev)
    for group in optimizer.param_groups:
        group["capturable"] = True
        group["lr"] = lr_t
    return lr_t


class GraphedTrainStep:
    """``step(batch) -> loss`` (a device scalar, valid until the next call) through hipGraph replay.

    model        an nn.Module on the HIP device with ``model.model_fn(model, batch) -> (loss, tb_dict, rtn)`` or any
                 callable ``loss_fn(model, batch) -> loss`` passed as ``loss_fn``
    optimizer    a capturable torch optimiser (see ``make_capturable``)
    example      one batch (dict of tensors / arrays with the shapes every later batch will have); entries named in
                 ``keys`` are copied into static device buffers
    """

    def __init__(self, model, optimizer, example, keys=("input", "target"), loss_fn=None, grad_norm_clip=0.0,
                 reducer=None, warmup=3, restore=True, keep_graph=False, deterministic_library=True):
        self._keep_graph = bool(keep_graph)      # diagnostics: the captured hipGraph_t stays reachable (raw_cuda_graph)
        self._deterministic_library = bool(deterministic_library)
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainStep needs a capturable optimiser (graph_step.make_capturable)")
        import torch.distributed as tdist
        # collectives inside the capture: RCCL only (stream-ordered); host-side backends keep the two-graph form
        self._fused_collective = (reducer is not None and tdist.is_available() and tdist.is_initialized()
                                  and tdist.get_backend() == "nccl")
        if reducer is not None and not self._fused_collective \
                and any(isinstance(m, pdist.SyncBatchNorm1d) for m in model.modules()):
            raise ValueError("a forward pass with collectives (SyncBatchNorm) can only be captured on RCCL "
                             "(backend nccl); run it eagerly")
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.clip = float(grad_norm_clip)
        self.dev = next(model.parameters()).device
        self.keys = tuple(keys)
        self._loss_fn = loss_fn if loss_fn is not None else (lambda m, b: m.model_fn(m, b)[0])
        self.static = {k: self._to_dev(example[k]).clone() for k in self.keys}
        self._extra = {k: v for k, v in example.items() if k not in self.keys}
        self.loss = None
        self._graphs = []
        # the warm-up steps before the capture are real optimisation steps on `example`; with `restore` the
        # parameters, buffers and optimiser state are put back afterwards (in place: the graphs hold their addresses)
        saved = [t.detach().clone() for t in list(model.parameters()) + list(model.buffers())] if restore else None
        self._capture(warmup)
        self._tracked = [p for p in model.parameters() if p.requires_grad]
        self._addresses = [(p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in self._tracked]
        if restore:
            with torch.no_grad():
                for t, v in zip(list(model.parameters()) + list(model.buffers()), saved):
                    t.copy_(v)
                for state in optimizer.state.values():
                    for v in state.values():
                        if torch.is_tensor(v):
                            v.zero_()

    def _to_dev(self, v):
        t = v if torch.is_tensor(v) else torch.as_tensor(v)
        return t.to(self.dev, non_blocking=True).float().contiguous()

    def _batch(self):
        b = dict(self._extra)
        b.update(self.static)
        return b

    def _fwd_bwd(self, fresh_grads=False):
        # fresh_grads (single-rank capture): the gradients are dropped, so backward WRITES each one (a tensor of the
        # graph's private pool, the same address on every replay) instead of zero-filling and accumulating into a
        # buffer -- two kernel nodes fewer per parameter.  With a reducer the gradients are views of its flat bucket
        # and have to stay where they are.
        self.optimizer.zero_grad(set_to_none=fresh_grads)
        loss = self._loss_fn(self.model, self._batch())
        loss.backward()
        return loss

    def _update(self):
        if self.clip > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.optimizer.step()

    def _capture(self, warmup):
        # Library convolutions inside a captured step must use DETERMINISTIC solvers: with MIOpen's default choice the
        # captured module path of the box head (SyncBatchNorm ranks keep the torch modules) replayed wrong convolution
        # gradients in about one process in ten -- a fixed 2 % deviation from the eager twin, gone in 12 of 12 runs with
        # the flag set, unaffected by the collectives (profiles/r3_graph_capture_miopen.txt).  The solver is chosen at a
        # shape's first call, so the warm-up steps run under the flag too.  MIOpen's deterministic solvers are slow on
        # these shapes (the box head's module path: 1.17 -> 4.3 ms per replay); ``deterministic_library=False`` keeps the
        # default choice for measurements that accept the risk.
        prev = torch.backends.cudnn.deterministic
        torch.backends.cudnn.deterministic = self._deterministic_library or prev
        try:
            self._capture_impl(warmup)
        finally:
            torch.backends.cudnn.deterministic = prev

    def _capture_impl(self, warmup):
        model = self.model
        model.train()
        # warm-up on a side stream: lazy initialisations (library handles, optimiser state, gradient buffers) must not
        # happen inside the capture
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            # the first backward creates the gradient buffers the graph will write into; parameters the loss does
            # not reach keep grad = None and are skipped by the optimiser, exactly as in an eager step
            for _ in range(max(1, warmup)):
                self._fwd_bwd()
                if self.reducer is not None:
                    self.reducer()
                self._update()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if self.reducer is None:
            g = torch.cuda.CUDAGraph(keep_graph=self._keep_graph)
            with torch.cuda.graph(g):
                self.loss = self._fwd_bwd(fresh_grads=True)
                self._update()
            self._graphs = [g]
        elif self._fused_collective:
            # the collectives are nodes of the graph.  thread_local: the process group's watchdog thread polls
            # events while this thread captures
            self.reducer.attach()                      # gradients = views of the bucket BEFORE the capture
            g = torch.cuda.CUDAGraph(keep_graph=self._keep_graph)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.loss = self._fwd_bwd()
                self.reducer.reduce_()
                self._update()
            self._graphs = [g]
        else:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self.loss = self._fwd_bwd()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update()
            self._graphs = [g1, g2]

    def _check_addresses(self):
        """The graph holds raw addresses of the parameters and of their gradient buffers: a caller that frees them
        (``zero_grad(set_to_none=True)``, ``p.grad = None``, ``model.to(...)``, ``load_state_dict(assign=True)``)
        would have the replay write into memory that is no longer theirs.  Refuse instead."""
        for p, (pa, ga) in zip(self._tracked, self._addresses):
            if p.data_ptr() != pa or (p.grad.data_ptr() if p.grad is not None else 0) != ga:
                raise RuntimeError("GraphedTrainStep: a parameter or gradient buffer moved since the capture (gradients "
                                   "set to None?) -- keep them allocated (zero_grad(set_to_none=False)) or build a new "
                                   "GraphedTrainStep")

    def step(self, batch):
        self._check_addresses()
        for k in self.keys:
            src = batch[k]
            self.static[k].copy_(src if torch.is_tensor(src) else torch.as_tensor(src), non_blocking=True)
        if self.reducer is None:
            self._graphs[0].replay()
        elif self._fused_collective:
            self._graphs[0].replay()
            self.reducer._publish_flag()               # the agreed stop flag of this step, read one step later
        else:
            self._graphs[0].replay()
            self.reducer()
            self._graphs[1].replay()
        return self.loss

    __call__ = step
