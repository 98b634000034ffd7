"""Scene crops on the GPU: the "small" scene images of a dataset stay resident in HBM and the
(b,4,33,33) crop tensor of a batch is produced by one kernel launch (csrc/crop.hip) instead of one PIL crop per
pedestrian plus two full-image resizes per scene on the host (trajectories_scene.py:314-356 of the reference).
Valid when the scene image is not transformed per item, i.e. without data augmentation (validation / test always,
training when `--augment` is off); results are bit-identical to the host path."""
import numpy as np
import torch
from torch.utils.data import Dataset

from mggan.data_utils.trajectories_scene import seq_collate_scene


class DeviceCropDataset(Dataset):
    """View of a TrajectoryDatasetEval whose items carry crop centres instead of crops."""

    def __init__(self, ds, device):
        if ds.data_augmentation and ds.phase == "train":
            raise ValueError("device crops need un-augmented scene images (the rotation / flips of the training "
                             "augmentation transform the image per item)")
        self.ds, self.device = ds, torch.device(device)
        parts, self.scene_rec, off = [], {}, 0
        for scene, rec in ds.images.items():
            arr = np.asarray(rec["small_image"])
            assert arr.ndim == 3 and arr.shape[2] == 3 and arr.dtype == np.uint8, (scene, arr.shape, arr.dtype)
            self.scene_rec[scene] = (off, arr.shape[0], arr.shape[1])
            parts.append(arr.reshape(-1))
            off += arr.size
        self.atlas = torch.from_numpy(np.concatenate(parts)).to(self.device)

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, index):
        ds = self.ds
        start, end = ds.seq_start_end[index]
        scene = ds.scene_list[index]
        xy = torch.from_numpy(ds.transformed_xy(index, 0, 0, ds.images[scene]["scaled_image"].size)).float()
        dxdy = xy[:, 1:] - xy[:, :-1]
        obs = xy[:, :ds.obs_len]
        centers = np.stack([ds.crop_center(obs[i, -1].numpy()) for i in range(end - start)]).astype(np.int32)
        off, h, w = self.scene_rec[scene]
        meta = (np.full(end - start, off, np.int64), np.tile(np.array([h, w], np.int32), (end - start, 1)), centers)
        return [obs, xy[:, ds.obs_len:], dxdy[:, :ds.obs_len - 1], dxdy[:, ds.obs_len - 1:], (end - start) * [scene], meta,
                torch.empty(1)]

    def collate(self, data):
        from mggan.hip.lib import lib

        metas = [d[5] for d in data]
        batch = seq_collate_scene([d[:5] + [torch.empty(0)] + d[6:] for d in data])
        off = torch.from_numpy(np.concatenate([m[0] for m in metas])).to(self.device)
        hw = torch.from_numpy(np.concatenate([m[1] for m in metas])).to(self.device)
        ctr = torch.from_numpy(np.concatenate([m[2] for m in metas])).to(self.device)
        n, m = off.numel(), self.ds.margin_in
        out = torch.empty(n, 4, 2 * m + 1, 2 * m + 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            lib.mggan_crop_patches(self.atlas.data_ptr(), off.data_ptr(), hw.data_ptr(), ctr.data_ptr(), n, m,
                                   out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        out.record_stream(torch.cuda.current_stream())
        batch["features"] = out
        self._keep = (off, hw, ctr)  # alive until the next batch (the launch is asynchronous)
        return batch
