"""Epoch loop, learning-rate schedule and checkpoints of the DR-SPAAM / Prototype training scripts
(reference src/utils/train_utils.py; bin/train_dr_spaam.py:103-118 shows the call).

``Trainer(model, model_fn, optimizer, ckpt_dir, lr_scheduler, model_fn_eval, grad_norm_clip,
tb_logger, logger)`` consumes any iterable of batches -- in particular the device loaders of
``src.utils.dataset_dr_spaam.create_dataloader`` -- and ``model_fn(model, batch)`` from
``src.utils.eval_utils``.  Under torch.distributed (one process per GPU) the gradients are averaged
through one flat all-reduce before clipping, and only rank 0 writes checkpoints and scalars.
"""
import os

import torch
from torch.nn.utils import clip_grad_norm_

from planar_optical_flow_amd import dist as pdist
from planar_optical_flow_amd.src.pipeline.logger import SummaryWriter, _JsonlWriter


# ---- checkpoints ({epoch, it, model_state, optimizer_state}, '<name>.pth') -------------------
def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    net = getattr(model, "module", model)            # unwrap DataParallel-style containers
    return {"epoch": epoch, "it": it,
            "model_state": None if net is None else net.state_dict(),
            "optimizer_state": None if optimizer is None else optimizer.state_dict()}


def save_checkpoint(state=None, filename="checkpoint"):
    torch.save(state, "{}.pth".format(filename))


def load_checkpoint(model=None, optimizer=None, filename="checkpoint", logger=None):
    """-> (iteration, epoch) stored in the file; raises FileNotFoundError when it does not exist."""
    if not os.path.isfile(filename):
        print("Could not find %s" % filename)
        raise FileNotFoundError(filename)
    ckpt = torch.load(filename, map_location="cpu", weights_only=False)
    if model is not None and ckpt.get("model_state") is not None:
        model.load_state_dict(ckpt["model_state"])
    if optimizer is not None and ckpt.get("optimizer_state") is not None:
        optimizer.load_state_dict(ckpt["optimizer_state"])
    return ckpt.get("it", 0.0), ckpt.get("epoch", -1)


def create_tb_logger(root_dir, tb_log_dir_name="tb_log"):
    path = os.path.join(root_dir, tb_log_dir_name)
    os.makedirs(path, exist_ok=True)
    return SummaryWriter(log_dir=path) if SummaryWriter else _JsonlWriter(path)


# ---- learning rate -----------------------------------------------------------------------
def lr_scheduler():
    """The constant the scripts hand to the optimiser before the schedule takes over."""
    return 0.01


class LucasScheduler:
    """v0 until epoch e0, geometric decay to v1 at e1, v1 until eNone (fractional epochs allowed)."""

    def __init__(self, optimizer, e0, v0, e1, v1, eNone=float("inf")):
        self._optim = optimizer
        self._knots = (e0, v0, e1, v1, eNone)

    def step(self, epoch):
        e0, v0, e1, v1, e_none = self._knots
        if epoch < e0:
            lr = v0
        elif epoch < e1:
            lr = v0 * (v1 / v0) ** ((epoch - e0) / (e1 - e0))
        elif epoch < e_none:
            lr = v1
        else:
            return
        for group in self._optim.param_groups:
            group["lr"] = lr

    def get_lr(self):
        return self._optim.param_groups[0]["lr"]


# ---- epoch loop ----------------------------------------------------------------------------
class Trainer:
    def __init__(self, model, model_fn, optimizer, ckpt_dir, lr_scheduler, model_fn_eval=None, grad_norm_clip=1.0,
                 tb_logger=None, logger=None):
        self.model, self.model_fn, self.model_fn_eval = model, model_fn, model_fn_eval
        self.optimizer, self.lr_scheduler = optimizer, lr_scheduler
        self.ckpt_dir, self.grad_norm_clip = ckpt_dir, grad_norm_clip
        self.tb_logger, self.logger = tb_logger, logger
        self._epoch = self._it = 0
        self._sync = pdist.GradientAllReduce(model) if pdist.is_distributed() else None
        self._master = pdist.rank() == 0

    def _scalar(self, key, val, step):
        if self.tb_logger is not None and self._master:
            self.tb_logger.add_scalar(key, val, step)

    def _train_it(self, batch):
        self.model.train()
        self.optimizer.zero_grad()
        out = self.model_fn(self.model, batch)
        loss = out[0] if isinstance(out, (tuple, list)) else out     # model_fn_dr_spaam also returns two norms
        loss.backward()
        if self._sync is not None:
            self._sync()
        if self.grad_norm_clip > 0:
            clip_grad_norm_(self.model.parameters(), self.grad_norm_clip)
        self.optimizer.step()
        return loss.item()

    def train(self, num_epochs, train_loader, eval_loader=None, eval_frequency=1, ckpt_save_interval=5,
              lr_scheduler_each_iter=True, starting_epoch=0, starting_iteration=0):
        self._it = starting_iteration
        n_it = len(train_loader)
        for self._epoch in range(starting_epoch, num_epochs):
            if not lr_scheduler_each_iter:
                self.lr_scheduler.step(self._epoch)
            running = 0.0
            for cur_it, batch in enumerate(train_loader):
                if lr_scheduler_each_iter:
                    self.lr_scheduler.step(self._epoch + cur_it / n_it)
                self._scalar("Learning_rate", self.lr_scheduler.get_lr(), self._it)
                loss = self._train_it(batch)
                running += loss
                self._scalar("Train_loss", loss, self._it)
                self._it += 1
            done = self._epoch + 1
            if self._master:
                print("Current Epoch: %d  [Learning rate: %s]" % (done, self.lr_scheduler.get_lr()))
                print("Epoch loss: ", running / max(n_it, 1))
            self._scalar("Epoch_loss", running / max(n_it, 1), self._epoch)
            if done % ckpt_save_interval == 0 and self._master:
                name = os.path.join(self.ckpt_dir, "ckpt_e{}".format(done))
                print("Saving checkpoint to {}".format(name))
                save_checkpoint(checkpoint_state(self.model, self.optimizer, done, self._it), filename=name)
            if eval_loader is not None and self.model_fn_eval is not None and done % eval_frequency == 0:
                self.model.eval()
                with torch.no_grad():
                    metrics = self.model_fn_eval(self.model, eval_loader)
                metrics = metrics if isinstance(metrics, (tuple, list)) else (metrics,)
                for k, v in enumerate(metrics):
                    self._scalar("val_metric_%d" % k, float(v), self._epoch)
                if self.logger is not None:
                    self.logger.info("Validation after epoch %d: %s" % (done, ", ".join("%.6g" % float(v) for v in metrics)))
            if self.tb_logger is not None and self._master:
                self.tb_logger.flush()
