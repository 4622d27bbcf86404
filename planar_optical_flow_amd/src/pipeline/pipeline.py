"""Facade over logger + optimiser + trainer with the reference's constructor and method names
(``Pipeline(model, cfg)``, reference src/pipeline/pipeline.py:6-36): the config sections
``Logger`` / ``Optim`` / ``Trainer`` go to the three components, every phase is bracketed by
debug-log lines, checkpoint helpers delegate to the logger."""
from . import logger as _logger
from . import optim as _optim
from . import trainer as _trainer


class Pipeline:
    _PHASES = {"train": "Training", "evaluate": "Evaluation"}

    def __init__(self, model, cfg):
        log = _logger.Logger(cfg["Logger"])
        opt = _optim.Optim(model, cfg["Optim"])
        self.logger, self.optim = log, opt
        self.trainer = _trainer.Trainer(log, opt, cfg["Trainer"])
        self._say("Pipeline starts.")

    # ---- logging helpers ---------------------------------------------------------------
    def _say(self, text):
        self.logger.log_debug(text)

    def _phase(self, name, *args):
        """Run ``trainer.<name>(*args)`` between 'starts' / 'ends (status ...)' debug lines; the status
        is the trainer's: 0 when the phase completed, 1 when it was interrupted (sigterm checkpoint)."""
        label = self._PHASES[name]
        self._say(label + " starts.")
        status = getattr(self.trainer, name)(*args)
        self._say("{} ends (status {}).".format(label, status))
        return status

    # ---- reference API -----------------------------------------------------------------
    def train(self, model, train_loader, eval_loader=None):
        return self._phase("train", model, train_loader, eval_loader)

    def evaluate(self, model, eval_loader, tb_prefix):
        return self._phase("evaluate", model, eval_loader, tb_prefix)

    def close(self):
        self._say("Pipeline closes.")
        self.logger.close()

    def load_ckpt(self, model, ckpt):
        return self.logger.load_ckpt(ckpt, model, self.optim)

    def load_sigterm_ckpt(self, model):
        return self.logger.load_sigterm_ckpt(model, self.optim)

    def sigterm_ckpt_exists(self):
        return self.logger.sigterm_ckpt_exists()
