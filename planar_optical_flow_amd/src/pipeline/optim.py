"""Optimiser wrapper of the training pipeline (reference src/pipeline/optim.py): Adam with amsgrad,
learning rate set once per epoch from an exponential-decay schedule given in the config as
``scheduler_kwargs = {epoch0, lr0, epoch1, lr1}``."""
import torch


def exp_decay_lr(epoch, epoch0, lr0, epoch1, lr1):
    """lr0 up to epoch0, lr1 from epoch1 on, geometric interpolation in between."""
    if epoch < epoch0:
        return lr0
    if epoch > epoch1:
        return lr1
    frac = (epoch - epoch0) / (epoch1 - epoch0)
    return lr0 * (lr1 / lr0) ** frac


class Optim:
    # calls forwarded unchanged to the wrapped torch optimiser
    _FORWARDED = ("zero_grad", "step", "state_dict", "load_state_dict")

    def __init__(self, model, cfg):
        self._schedule = dict(cfg["scheduler_kwargs"])
        params = list(model.parameters())
        # on the device: the fused multi-tensor Adam -- ONE launch for all 72 parameter tensors of the box head
        # instead of ~220 element-wise launches per step (round 2's profile: 16 016 DivFunctor launches in 73 steps);
        # same update rule (amsgrad), capturable in a hipGraph
        fused = len(params) > 0 and all(p.is_cuda and p.is_floating_point() for p in params)
        self._optim = torch.optim.Adam(params, amsgrad=True, fused=True) if fused \
            else torch.optim.Adam(params, amsgrad=True)

    def __getattr__(self, name):
        if name in Optim._FORWARDED:
            return getattr(self.__dict__["_optim"], name)
        raise AttributeError(name)

    def set_lr(self, epoch):
        lr = exp_decay_lr(epoch, **self._schedule)
        for group in self._optim.param_groups:
            if torch.is_tensor(group["lr"]):      # capturable form: a device scalar the captured step reads
                group["lr"].fill_(lr)
            else:
                group["lr"] = lr
        self._lr = lr

    def get_lr(self):
        lr = self._optim.param_groups[0]["lr"]
        return getattr(self, "_lr", float(lr)) if torch.is_tensor(lr) else lr

    def make_capturable(self):
        """Device-resident step counters and learning rate, so that ``step()`` can be captured in a hipGraph
        (planar_optical_flow_amd.graph_step).  Before the first step only."""
        from planar_optical_flow_amd.graph_step import make_capturable
        make_capturable(self._optim)
        return self._optim


class _ExpDecayScheduler:
    """Callable form of ``exp_decay_lr`` under the reference's name and constructor arguments."""

    def __init__(self, epoch0, lr0, epoch1, lr1):
        self._kw = dict(epoch0=epoch0, lr0=lr0, epoch1=epoch1, lr1=lr1)

    def __call__(self, epoch):
        return exp_decay_lr(epoch, **self._kw)
