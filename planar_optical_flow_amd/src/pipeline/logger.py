"""Run directory, python logging, scalar log and checkpoints (reference
src/pipeline/logger.py).  tensorboardX is optional: when it is not installed
scalars are appended to ``scalars.jsonl`` in the run directory instead.  Under
torch.distributed only rank 0 writes files."""
import json
import logging
import os
import pickle
import time
from shutil import copyfile

import numpy as np
import torch

from planar_optical_flow_amd import dist as pdist

try:  # pragma: no cover - not installed in the build image
    from tensorboardX import SummaryWriter
except Exception:  # noqa: BLE001
    SummaryWriter = None


class _JsonlWriter:
    def __init__(self, log_dir):
        self._fp = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, key, val, step):
        self._fp.write(json.dumps({"key": key, "value": float(val), "step": int(step)}) + "\n")

    def add_image(self, key, im, step):
        pass

    def flush(self):
        self._fp.flush()

    def close(self):
        self._fp.close()


class Logger:
    def __init__(self, cfg):
        self._master = pdist.rank() == 0
        root = os.path.abspath(os.path.expanduser(cfg["log_dir"]))
        stamp = time.strftime("%Y%m%d_%H%M%S", time.gmtime())
        self._dir = os.path.join(root, "%s_%s" % (stamp, cfg["tag"]))
        self._sub = {k: os.path.join(self._dir, k) for k in ("backup", "output", "image", "ckpt", "tb")}
        self._sigterm_ckpt = os.path.join(root, "sigterm_ckpt_%s.pth" % cfg["tag"])
        self._log = logging.getLogger("%s.%d" % (__name__, id(self)))
        self._log.setLevel(logging.DEBUG)
        self._log.propagate = False
        fmt = logging.Formatter("%(asctime)s  %(levelname)5s  %(message)s")
        self._tb = None
        if self._master:
            for d in [self._dir] + list(self._sub.values()):
                os.makedirs(d, exist_ok=True)
            fh = logging.FileHandler(os.path.join(self._dir, cfg.get("log_fname", "log.txt")))
            fh.setFormatter(fmt)
            self._log.addHandler(fh)
            sh = logging.StreamHandler()
            sh.setFormatter(fmt)
            self._log.addHandler(sh)
            for f in cfg.get("backup_list", []):
                copyfile(os.path.abspath(f), os.path.join(self._sub["backup"], os.path.basename(f)))
            self._tb = SummaryWriter(log_dir=self._sub["tb"]) if SummaryWriter else _JsonlWriter(self._sub["tb"])
        self.log_debug("Log directory: %s" % self._dir)
        self.log_info("HIP_VISIBLE_DEVICES=%s" % os.environ.get("HIP_VISIBLE_DEVICES",
                                                                os.environ.get("CUDA_VISIBLE_DEVICES", "ALL")))

    # ---- python log -----------------------------------------------------------
    def log_warning(self, s):
        self._log.warning(s)

    def log_info(self, s):
        self._log.info(s)

    def log_debug(self, s):
        self._log.debug(s)

    # ---- scalars / images -------------------------------------------------------
    def add_scalar(self, key, val, step):
        if self._tb is not None:
            self._tb.add_scalar(key, float(val), step)

    def add_im(self, key, im, step):
        if self._tb is not None:
            self._tb.add_image(key, im, step)

    def add_fig(self, key, fig, step, close_fig=False):
        """A matplotlib figure as a [3, H, W] float image in the event log (reference logger.py:107-117)."""
        if self._tb is not None:
            fig.canvas.draw()
            rgba = np.asarray(fig.canvas.buffer_rgba(), dtype=np.uint8)
            self.add_im(key, rgba[..., :3].transpose(2, 0, 1).astype(np.float32) / 255.0, step)
        if close_fig:
            import matplotlib.pyplot as plt
            plt.close(fig)

    def save_fig(self, fig, fname, close_fig=False):
        """The figure as a file under the run's image directory (reference :148-152)."""
        if self._master:
            fig.savefig(os.path.join(self._sub["image"], fname))
        if close_fig:
            import matplotlib.pyplot as plt
            plt.close(fig)

    def flush(self):
        if self._tb is not None:
            self._tb.flush()

    def close(self):
        if self._tb is not None:
            self._tb.close()
        for h in self._log.handlers[:]:
            h.close()
            self._log.removeHandler(h)

    # ---- files --------------------------------------------------------------------
    def save_dict(self, fname, dict_):
        if not self._master:
            return
        scalars = {k: str(v) for k, v in dict_.items() if not isinstance(v, (np.ndarray, tuple, list, dict))}
        with open(os.path.join(self._sub["output"], fname + ".json"), "w") as fp:
            json.dump(scalars, fp, sort_keys=True, indent=4)
        with open(os.path.join(self._sub["output"], fname + ".pkl"), "wb") as fp:
            pickle.dump(dict_, fp, protocol=pickle.HIGHEST_PROTOCOL)
        self.log_info("Dictionary saved to %s.{json,pkl}" % os.path.join(self._sub["output"], fname))

    # ---- checkpoints: {"epoch","step","model_state","optimizer_state"} ----------------
    def save_ckpt(self, fname, model, optimizer, epoch, step):
        if not self._master:
            return
        if not os.path.dirname(fname):
            fname = os.path.join(self._sub["ckpt"], fname)
        core = getattr(model, "module", model)  # un-prefixed keys, as the reference writes them
        torch.save({"epoch": epoch, "step": step,
                    "model_state": core.state_dict() if model is not None else None,
                    "optimizer_state": optimizer.state_dict() if optimizer is not None else None}, fname)
        self.log_info("Checkpoint saved to %s." % fname)

    def load_ckpt(self, fname, model, optimizer=None):
        ckpt = torch.load(fname, map_location=next(model.parameters()).device, weights_only=False)
        model.load_state_dict(ckpt["model_state"])
        if optimizer is not None and ckpt.get("optimizer_state") is not None:
            optimizer.load_state_dict(ckpt["optimizer_state"])
        epoch, step = ckpt.get("epoch", 0), ckpt.get("step", 0)
        self.log_info("Load checkpoint %s: epoch %s, step %s." % (fname, epoch, step))
        return epoch, step

    def save_sigterm_ckpt(self, model, optimizer, epoch, step):
        self.save_ckpt(self._sigterm_ckpt, model, optimizer, epoch, step)

    def load_sigterm_ckpt(self, model, optimizer):
        return self.load_ckpt(self._sigterm_ckpt, model, optimizer)

    def sigterm_ckpt_exists(self):
        return os.path.isfile(self._sigterm_ckpt)
