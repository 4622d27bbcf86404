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
    """Averages the gradients of `model` over all ranks through ONE flat bucket.

    The bucket is allocated once; ``__call__`` packs the .grad tensors, issues a
    single all-reduce(sum), scales by 1/world and unpacks.  Parameters without a
    gradient (frozen / unused, e.g. the duplicate conv1..conv4 of the box head)
    contribute zeros so every rank reduces the same layout."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.bucket = torch.zeros(n, dtype=torch.float32, device=dev)

    def __call__(self):
        if not is_distributed():
            return
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.bucket[off:off + n].zero_()
            else:
                self.bucket[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM)
        self.bucket.mul_(1.0 / world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.bucket[off:off + n].reshape(p.shape).clone()
            else:
                p.grad.copy_(self.bucket[off:off + n].reshape(p.shape))
            off += n


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not is_distributed():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)
