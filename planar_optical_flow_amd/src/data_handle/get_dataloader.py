"""``get_dataloader(split, batch_size, num_workers, shuffle, dataset_cfg)`` of the reference
(src/data_handle/get_dataloader.py) over the device-resident box-regression data set.

The reference wraps ``JRDBBoxRegressionDataset`` in a ``torch.utils.data.DataLoader`` with worker
processes; here a batch is one kernel launch, so the loader is a plain iterable that draws index
batches (shuffled per epoch when asked) and calls ``dataset.get_batch`` -- ``num_workers`` is
accepted and ignored.  The frames come from ``JRDBHandle(split, dataset_cfg)`` reading the JRDB tree under
``dataset_cfg["data_dir"]``, or -- optional -- from ``dataset_cfg["frames"]`` (already parsed dicts with
``segments``, ``boxes``, ``dets_center``).
"""
import numpy as np

from .jrdb_dataset import JRDBBoxRegressionDataset


class DeviceBatchLoader:
    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, seed=0):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self._rng = np.random.default_rng(seed)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        n = len(self.dataset)
        order = self._rng.permutation(n) if self.shuffle else np.arange(n)
        for k in range(len(self)):
            yield self.dataset.get_batch(order[k * self.batch_size:(k + 1) * self.batch_size].tolist())


def get_dataloader(split, batch_size, num_workers, shuffle, dataset_cfg):
    if "JRDB" not in dataset_cfg["data_dir"]:
        raise RuntimeError("Unknown dataset {}.".format(dataset_cfg.get("name", dataset_cfg["data_dir"])))
    ds = JRDBBoxRegressionDataset(split, dataset_cfg, dataset_cfg.get("frames"))
    return DeviceBatchLoader(ds, batch_size, shuffle=shuffle)
