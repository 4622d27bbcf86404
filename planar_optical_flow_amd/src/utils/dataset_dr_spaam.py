"""``DROWDataset2`` / ``create_dataloader`` of the reference (src/utils/dataset_dr_spaam.py:12-68,
:256-471) on the device-resident scan store.

Same constructor arguments; the split directory is parsed once (``drow_io``), the sequences live in
HBM and a batch is a handful of launches (window gather, odometry association, fused preprocess,
then the network input of the configured ``network_type``):

    cutout / cutout_gating / cutout_spatial   input = scans_to_cutout          [B, N, T+1, P]
    fc1d                                      input = scans[:, :, None]        [B, T+1, 1, N]
    fc1d_fea                                  input = cutout as (T+1, P, N)    [B, T+1, P, N]
    fc2d                                      input = polar TSDF grid          [B, T+1, 1, R, N]

``use_data_augumentation`` is the reference's per-sample left-right flip (src/utils/utils.py:129-144):
with probability 1/2 the window is mirrored and ``target_reg[:, 0]`` negated -- applied, as there,
after the targets are computed and before the network input is built.  The loaders are plain
iterables over index batches (``num_workers`` is accepted and ignored).
"""
import numpy as np
import torch

from ... import ops
from ...scan_store import DROWDeviceDataset
from ...src.data_handle.get_dataloader import DeviceBatchLoader
from ... import drow_io

_CUTOUT_TYPES = ("cutout", "cutout_gating", "cutout_spatial")


class DROWDataset2(DROWDeviceDataset):
    def __init__(self, data_path, split="train", num_scans=5, network_type="cutout", train_with_val=False,
                 cutout_kwargs=None, polar_grid_kwargs=None, use_data_augumentation=False, pedestrian_only=False,
                 scan_stride=1, pt_stride=1, max_scan_dist=6, device="cuda", seed=0, sequences=None,
                 drop_static=True):
        if pt_stride != 1:
            raise NotImplementedError("pt_stride != 1 (marked for removal in the reference) is not supported")
        if network_type == "fc2d_fea":
            raise NotImplementedError
        if network_type in _CUTOUT_TYPES and cutout_kwargs is not None and "area_mode" not in cutout_kwargs:
            raise NotImplementedError("the legacy cv2 cutout (no area_mode key) is not rebuilt")
        self._network_type, self._cutout_kwargs, self._polar_grid_kwargs = network_type, cutout_kwargs, polar_grid_kwargs
        self._use_data_augmentation = use_data_augumentation
        self.max_scan_dist = max_scan_dist
        self._gen = torch.Generator(device="cpu").manual_seed(seed)
        seqs = sequences if sequences is not None else drow_io.load_sequences(data_path, split)
        # the store computes targets only; the network input is built here after the augmentation
        super().__init__(seqs, num_scans=num_scans, cutout_kwargs=None, pedestrian_only=pedestrian_only,
                         scan_stride=scan_stride, device=device, drop_static=drop_static)

    def get_batch(self, indices):
        batch = super().get_batch(indices)
        scans = batch["scans"]
        if self._use_data_augmentation:
            flip = (torch.rand(len(indices), generator=self._gen) < 0.5).to(scans.device)
            scans = torch.where(flip[:, None, None], scans.flip(-1), scans).contiguous()
            reg = batch["target_reg"].clone()
            reg[..., 0] = torch.where(flip[:, None], -reg[..., 0], reg[..., 0])
            batch["scans"], batch["target_reg"] = scans, reg
        nt = self._network_type
        if nt in _CUTOUT_TYPES:
            batch["input"] = ops.cutout(scans, self.pre.tab, stride=1, **self._cutout_kwargs)
        elif nt == "fc1d":
            batch["input"] = scans.unsqueeze(2)
        elif nt == "fc1d_fea":
            batch["input"] = ops.cutout(scans, self.pre.tab, stride=1, **self._cutout_kwargs).permute(0, 2, 3, 1)
        elif nt == "fc2d":
            batch["input"] = ops.polar_grid(scans, **(self._polar_grid_kwargs or {})).unsqueeze(2)
        return batch

    def __getitem__(self, idx):
        """One sample as the reference's dict of NumPy arrays (use get_batch / the loaders for training)."""
        b = self.get_batch([idx])
        return {k: (v[0].cpu().numpy() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1 else
                    (v.cpu().numpy() if torch.is_tensor(v) else v[0])) for k, v in b.items()}

    def collate_batch(self, batch):
        tensor_keys = ("scans", "target_cls", "target_reg", "input", "target_flow", "exclude_mask", "odom")
        return {k: (np.array([s[k] for s in batch]) if k in tensor_keys else [s[k] for s in batch]) for k in batch[0]}


def create_dataloader(data_path, num_scans, batch_size, num_workers, network_type="cutout", train_with_val=False,
                      use_data_augumentation=False, cutout_kwargs=None, polar_grid_kwargs=None,
                      pedestrian_only=False):
    """-> (train_loader, eval_loader or None), both shuffled like the reference's."""
    common = dict(num_scans=num_scans, network_type=network_type, cutout_kwargs=cutout_kwargs,
                  polar_grid_kwargs=polar_grid_kwargs, pedestrian_only=pedestrian_only)
    train_set = DROWDataset2(data_path=data_path, split="train", train_with_val=train_with_val,
                             use_data_augumentation=use_data_augumentation, **common)
    train_loader = DeviceBatchLoader(train_set, batch_size, shuffle=True)
    if not train_with_val:
        return train_loader, None
    eval_set = DROWDataset2(data_path=data_path, split="val", **common)
    return train_loader, DeviceBatchLoader(eval_set, batch_size, shuffle=True)


def create_test_dataloader(data_path, num_scans, network_type="cutout", cutout_kwargs=None, polar_grid_kwargs=None,
                           pedestrian_only=False, split="test", scan_stride=1, pt_stride=1):
    test_set = DROWDataset2(data_path=data_path, split=split, num_scans=num_scans, network_type=network_type,
                            cutout_kwargs=cutout_kwargs, polar_grid_kwargs=polar_grid_kwargs,
                            pedestrian_only=pedestrian_only, scan_stride=scan_stride, pt_stride=pt_stride)
    return DeviceBatchLoader(test_set, 1, shuffle=False)
