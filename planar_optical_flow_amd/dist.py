"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) on the xGMI links of one node; gloo on CPU for tests.

The hot path shards over the batch axis with no data-path
collective.  The only exchange is the detector / box head's gradient step: one
flat float32 bucket all-reduced once per optimisation step (3.8 MB for
BoundingBoxRegressor, 7.9 MB for SpatialDROW -- latency bound on xGMI, so a
single bucket and no overlap machinery), hooked between ``loss.backward()`` and
gradient clipping exactly where the single-GPU reference clips
(src/pipeline/trainer.py:115-118).
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def backend():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return rank(), world_size(), device


def shard_range(n, r=None, w=None):
    """Contiguous batch-axis split: rank r of w owns [lo, hi) of n samples."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per, rem = divmod(n, w)
    lo = r * per + min(r, rem)
    return lo, lo + per + (1 if r < rem else 0)


class GradientAllReduce:
    """Averages the gradients of `model` over all ranks through ONE flat bucket -- without packing.

    Every ``p.grad`` IS a view into the bucket (set here, kept by ``zero_grad(set_to_none=False)``; autograd
    accumulates into an existing gradient in place), so ``__call__`` is the collective alone: one all-reduce, no
    per-parameter copies, no host synchronisation -- and capturable in a hipGraph together with the step around it
    (RCCL collectives are stream-ordered; ``graph_step.GraphedTrainStep``).  A gradient that was dropped or
    replaced since (``zero_grad(set_to_none=True)``, ``p.grad = None``) is copied back into its slice and
    re-pointed on the next call, so correctness never depends on the caller's zero_grad flavour.  Parameters the
    loss does not reach (the duplicate conv1..conv4 of the box head) hold zeros, so every rank reduces the same
    layout.

    The bucket's LAST element is the agreed stop flag of the training loop (``set_stop`` / ``stop_requested``): it
    rides in the same all-reduce, and is read back one step later through a pinned host word and an event, i.e.
    without a blocking device-to-host copy in the step."""

    def __init__(self, model, always=False):
        self.always = bool(always)       # run the collective in a one-rank group too (RCCL smoke / capture tests)
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.bucket = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.views.append(self.bucket[off:off + k].view(p.shape))
            off += k
        self.flag = self.bucket[n:n + 1]
        self._host_flag = None
        self._flag_event = None
        self._pending = False
        self.attach()

    def attach(self):
        """Point every parameter's gradient at its slice of the bucket (keeping what it holds)."""
        with torch.no_grad():
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                p.grad = v
        return self

    def set_stop(self, stop):
        """This rank's stop request for the coming all-reduce (no synchronisation: a fill)."""
        self.flag.fill_(1.0 if stop else 0.0)

    def stop_requested(self):
        """True once ANY rank had asked to stop at the previous reduced step (the same answer on every rank)."""
        if not self._pending:
            return False
        if self._flag_event is not None:
            self._flag_event.synchronize()          # recorded a whole step ago: normally already complete
        return bool(self._host_flag[0] > 0)

    def _publish_flag(self):
        if self.bucket.is_cuda:
            if self._host_flag is None:
                self._host_flag = torch.zeros(1, dtype=torch.float32).pin_memory()
                self._flag_event = torch.cuda.Event()
            self._host_flag.copy_(self.flag, non_blocking=True)
            self._flag_event.record()
        else:
            self._host_flag = self.flag.clone()
        self._pending = True

    def reduce_(self):
        """The collective alone (what a captured step holds): bucket <- mean over ranks."""
        if dist.get_backend() == "nccl":
            dist.all_reduce(self.bucket, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM)
            self.bucket.mul_(1.0 / world_size())

    def __call__(self, force=False):
        """force=True also runs the (then trivial) collective in a one-rank group: the world-size-1 RCCL smoke
        test drives the device path that way."""
        if not (is_distributed() or ((force or self.always) and dist.is_available() and dist.is_initialized())):
            return
        if any(p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(self.params, self.views)):
            self.attach()
        self.reduce_()
        if not (self.bucket.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._publish_flag()


def any_rank(flag, device=None):
    """True on every rank as soon as `flag` is true on one: the stop flag of the training loop (a SIGTERM may
    reach only some ranks; without agreement the others would block in the next gradient all-reduce).  A blocking
    form for callers without a gradient bucket; the training loop uses the bucket's flag element instead."""
    if not is_distributed():
        return bool(flag)
    use_dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=use_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


class _SyncBatchNormFn(torch.autograd.Function):
    """BatchNorm over the GLOBAL batch: per-channel sum / sum of squares (2C + 1 numbers, float64) are all-reduced
    in the forward, per-channel sum(dy) / sum(dy * xhat) (2C numbers) in the backward.  Works on any backend
    (gloo on CPU for the tests, RCCL on the GPUs)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        C = x.shape[1]
        dims = [d for d in range(x.dim()) if d != 1]
        xd = x.double()
        stat = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
        stat[:C] = xd.sum(dims)
        stat[C:2 * C] = (xd * xd).sum(dims)
        stat[2 * C:].fill_(float(x.numel() // C))      # a fill kernel: no host-to-device copy (capturable)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
        n = stat[2 * C]
        mean = stat[:C] / n
        var = torch.clamp(stat[C:2 * C] / n - mean * mean, min=0.0)            # biased, as BatchNorm normalises
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(momentum * mean.to(running_mean.dtype))
                running_var.mul_(1 - momentum).add_(momentum * (var * n / torch.clamp(n - 1, min=1.0)).to(running_var.dtype))
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = ((xd - mean.view(shape)) * invstd.view(shape)).to(x.dtype)
        ctx.save_for_backward(xhat, weight, invstd.to(x.dtype), n)     # n stays on the device: no host sync
        return xhat * weight.view(shape) + bias.view(shape)

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, invstd, n = ctx.saved_tensors
        C = xhat.shape[1]
        dims = [d for d in range(xhat.dim()) if d != 1]
        shape = [1, C] + [1] * (xhat.dim() - 2)
        red = torch.empty(2 * C, dtype=torch.float64, device=dy.device)
        red[:C] = dy.double().sum(dims)
        red[C:] = (dy.double() * xhat.double()).sum(dims)
        d_bias, d_weight = red[:C].to(dy.dtype).clone(), red[C:].to(dy.dtype).clone()   # local: the gradient
        dist.all_reduce(red, op=dist.ReduceOp.SUM)                                       # bucket averages them
        s1, s2 = (red[:C] / n).to(dy.dtype), (red[C:] / n).to(dy.dtype)
        dx = (weight * invstd).view(shape) * (dy - s1.view(shape) - xhat * s2.view(shape))
        return dx, d_weight, d_bias, None, None, None, None


class SyncBatchNorm1d(torch.nn.BatchNorm1d):
    """nn.BatchNorm1d whose training-mode statistics span all ranks (same parameters, buffers and state-dict
    keys).  Eval mode and un-initialised process groups fall through to the stock module."""

    def forward(self, x):
        if not (self.training and dist.is_available() and dist.is_initialized()) or not self.track_running_stats:
            return super().forward(x)
        if self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        momentum = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
        return _SyncBatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                      momentum)


def convert_sync_batchnorm(module):
    """Swap every nn.BatchNorm1d of `module` for SyncBatchNorm1d IN PLACE (the class of the existing objects is
    changed, so parameters, buffers, optimizer references and checkpoint keys stay what they were)."""
    for m in module.modules():
        if type(m) is torch.nn.BatchNorm1d:
            m.__class__ = SyncBatchNorm1d
    return module


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not is_distributed():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)
