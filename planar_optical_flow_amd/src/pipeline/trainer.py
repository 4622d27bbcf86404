"""Train / eval loop (reference src/pipeline/trainer.py) with the multi-GPU
gradient exchange added: between ``loss.backward()`` and gradient clipping the
gradients are averaged over all ranks through one flat RCCL all-reduce
(planar_optical_flow_amd.dist.GradientAllReduce), so clipping and Adam see the
global-batch gradient exactly as on one GPU.  Under torch.distributed the BatchNorm
layers take their training statistics over the global batch (dist.SyncBatchNorm1d;
cfg key ``sync_bn``, default on), so the N-rank trajectory equals the single-process
one on the same global batch, and the stop flag of the signal handler is agreed over
all ranks once per step (a signal that reaches one rank stops all of them at the same
step; rank 0 then writes the sigterm checkpoint)."""
import signal

import torch
from torch.nn.utils import clip_grad_norm_

from planar_optical_flow_amd import dist as pdist


class Trainer:
    def __init__(self, logger, optimizer, cfg):
        self._logger, self._optim = logger, optimizer
        self._epoch, self._step = 0, 0
        self._grad_norm_clip = cfg["grad_norm_clip"]
        self._ckpt_interval = cfg["ckpt_interval"]
        self._eval_interval = cfg["eval_interval"]
        self._max_epoch = cfg["epoch"]
        self._stop = False
        self._reducer = None
        self._sync_bn = bool(cfg.get("sync_bn", True))
        # graph_step: capture zero_grad / forward / backward / clip / Adam of a fixed-shape batch as one hipGraph and
        # replay it per batch (single process, model on the GPU); the last, shorter batch of an epoch runs eagerly
        self._graph_step = bool(cfg.get("graph_step", False))
        self._graphed, self._graphed_shapes = None, None
        self._dist_ready = False
        try:
            signal.signal(signal.SIGINT, self._on_signal)
            signal.signal(signal.SIGTERM, self._on_signal)
        except ValueError:  # not the main thread (e.g. under a test runner)
            pass

    def _on_signal(self, signum, frame):
        self._stop = True
        self._logger.log_info("Received signal %s." % signum)

    # ---- evaluation ---------------------------------------------------------------
    def evaluate(self, model, eval_loader, tb_prefix):
        model.eval()
        keys = ("iou", "loss_z", "loss_dim", "loss_ori")
        acc = dict.fromkeys(keys, 0.0)
        total = 0.0
        for batch in eval_loader:
            if self._stop:
                return 1
            with torch.no_grad():
                loss, _, rtn = model.model_eval_fn(model, batch)
            total += loss.item()
            for k in keys:
                acc[k] += rtn[k]
        n = max(len(eval_loader), 1)
        stats = {"eval_loss": total / n, "avg_iou": acc["iou"] / n, "avg_loss_z": acc["loss_z"] / n,
                 "avg_loss_dim": acc["loss_dim"] / n, "avg_loss_ori": acc["loss_ori"] / n}
        for k, v in stats.items():
            self._logger.add_scalar("%s_%s" % (tb_prefix, k), v, self._step)
            self._logger.log_info("%s: %s" % (k, v))
        return 0

    # ---- training -------------------------------------------------------------------
    def _stop_agreed(self, model):
        """The local flag, or -- under torch.distributed -- the OR over all ranks (and then set locally too).  With
        a gradient bucket the flags ride in the step's own all-reduce (its last element): this call reads the answer
        of the PREVIOUS step (a pinned host word, no blocking copy) and files this rank's request for the next, so
        every rank leaves the loop at the same step, one step after the first signal."""
        if pdist.is_distributed():
            self._prepare_distributed(model)
            if self._reducer is not None:
                if self._reducer.stop_requested():
                    self._stop = True
                    return True
                self._reducer.set_stop(self._stop)
                return False
            dev = next(model.parameters()).device
            self._stop = pdist.any_rank(self._stop, dev)
        return self._stop

    def _prepare_distributed(self, model):
        if self._dist_ready or not pdist.is_distributed():
            return
        if self._sync_bn:
            pdist.convert_sync_batchnorm(model)
        self._reducer = pdist.GradientAllReduce(model)
        self._dist_ready = True

    def train(self, model, train_loader, eval_loader=None):
        for self._epoch in range(0, self._max_epoch):
            if self._stop_agreed(model):
                self._logger.save_sigterm_ckpt(model, self._optim, self._epoch, self._step)
                return 1
            self._train_epoch(model, train_loader)
            if not self._stop:
                if self._epoch % self._ckpt_interval == 0 or self._epoch == self._max_epoch:
                    self._logger.save_ckpt("ckpt_e%d.pth" % self._epoch, model, self._optim, self._epoch,
                                           self._step)
                if eval_loader is not None and (self._epoch % self._eval_interval == 0
                                                or self._epoch == self._max_epoch):
                    self.evaluate(model, eval_loader, tb_prefix="VAL")
            self._logger.flush()
        return 0

    def _graph_shapes(self, batch):
        return tuple((k, tuple(batch[k].shape)) for k in ("input", "target"))

    def _train_batch_graphed(self, model, batch, ratio):
        from planar_optical_flow_amd.graph_step import GraphedTrainStep
        if self._graphed is None:
            if self._step != 0:
                raise RuntimeError("graph_step has to be on from the first step (the optimiser state must be capturable)")
            optim = self._optim.make_capturable()
            self._graphed = GraphedTrainStep(model, optim, batch, grad_norm_clip=self._grad_norm_clip,
                                             reducer=self._reducer)
            self._graphed_shapes = self._graph_shapes(batch)
        self._optim.set_lr(self._epoch + ratio)
        loss = self._graphed.step(batch).item()
        self._logger.add_scalar("TRAIN_lr", self._optim.get_lr(), self._step)
        self._logger.add_scalar("TRAIN_loss", loss, self._step)
        self._logger.add_scalar("TRAIN_epoch", self._epoch + ratio, self._step)
        return loss

    def _train_batch(self, model, batch, ratio):
        self._prepare_distributed(model)
        model.train()
        # with a reducer the step is captured only where the collective can be a node of the graph (RCCL)
        graphable = self._reducer is None or pdist.backend() == "nccl"
        if self._graph_step and graphable and next(model.parameters()).is_cuda \
                and (self._graphed is None or self._graph_shapes(batch) == self._graphed_shapes):
            return self._train_batch_graphed(model, batch, ratio)
        # once a step is captured its graph owns the addresses of the gradient buffers: keep them allocated
        # (and so does a gradient bucket: the gradients are views into it)
        self._optim.zero_grad(set_to_none=self._graphed is None and self._reducer is None)
        self._optim.set_lr(self._epoch + ratio)
        loss, tb_dict, _ = model.model_fn(model, batch)
        loss.backward()
        if self._reducer is not None:
            self._reducer()
        if self._grad_norm_clip > 0:
            clip_grad_norm_(model.parameters(), self._grad_norm_clip)
        self._optim.step()
        self._logger.add_scalar("TRAIN_lr", self._optim.get_lr(), self._step)
        self._logger.add_scalar("TRAIN_loss", loss.item(), self._step)
        self._logger.add_scalar("TRAIN_epoch", self._epoch + ratio, self._step)
        for k, v in tb_dict.items():
            self._logger.add_scalar("TRAIN_%s" % k, v, self._step)
        return loss.item()

    def _train_epoch(self, model, train_loader):
        total, n = 0.0, max(len(train_loader), 1)
        for ib, batch in enumerate(train_loader):
            if self._stop_agreed(model):
                return
            total += self._train_batch(model, batch, ratio=ib / n)
            self._step += 1
        self._logger.log_info("Current epoch: %d, training loss: %s" % (self._epoch, total / n))
