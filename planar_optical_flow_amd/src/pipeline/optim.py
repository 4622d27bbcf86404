"""Adam(amsgrad) + exponential-decay learning rate (reference src/pipeline/optim.py)."""
from torch import optim


class _ExpDecayScheduler:
    """lr0 until epoch0, geometric interpolation to lr1 at epoch1, lr1 afterwards."""

    def __init__(self, epoch0, lr0, epoch1, lr1):
        self.e0, self.e1, self.lr0, self.lr1 = epoch0, epoch1, lr0, lr1

    def __call__(self, epoch):
        if epoch < self.e0:
            return self.lr0
        if epoch > self.e1:
            return self.lr1
        return self.lr0 * (self.lr1 / self.lr0) ** ((epoch - self.e0) / (self.e1 - self.e0))


class Optim:
    def __init__(self, model, cfg):
        self._optim = optim.Adam(model.parameters(), amsgrad=True)
        self._lr_scheduler = _ExpDecayScheduler(**cfg["scheduler_kwargs"])

    def zero_grad(self):
        self._optim.zero_grad()

    def step(self):
        self._optim.step()

    def state_dict(self):
        return self._optim.state_dict()

    def load_state_dict(self, state_dict):
        self._optim.load_state_dict(state_dict)

    def set_lr(self, epoch):
        lr = self._lr_scheduler(epoch)
        for group in self._optim.param_groups:
            group["lr"] = lr

    def get_lr(self):
        return self._optim.param_groups[0]["lr"]
