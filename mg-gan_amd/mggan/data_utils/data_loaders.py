"""get_dataloader (surface of /root/reference/mggan/data_utils/data_loaders.py:10-100): the on-disk datasets
(ETH/UCY, GOFP, SDD: tab-separated text + one jpg per scene, mggan/data_utils/trajectories_scene.py) and the
built-in 'synthetic' dataset, both with the reference's collate schema."""
import torch

from mggan.data_utils import synthetic


class SyntheticScenes(torch.utils.data.Dataset):
    """Each item is one batch of `scenes_per_batch` scenes (already collated)."""

    def __init__(self, num_batches, scenes_per_batch, peds_per_scene=None, seed=0, cache_device=None):
        self.num_batches, self.spb, self.pps, self.seed = num_batches, scenes_per_batch, peds_per_scene, seed
        self.dataset_name = "synthetic"
        # cache_device: a produced batch stays resident in HBM (288 GB: an epoch of 64x20-pedestrian batches is 22 MB
        # each); later epochs hand over device tensors, with the all-valid verdict of the NaN scan taken once on the host
        self.cache_device, self._cache = cache_device, {}

    def __len__(self):
        return self.num_batches

    def __getitem__(self, i):
        hit = self._cache.get(i)
        if hit is not None:
            return hit
        sizes = synthetic.scene_sizes(self.spb, self.pps, seed=self.seed + i)
        batch = synthetic.make_batch(sizes, seed=self.seed + 1000 + i)
        if self.cache_device is not None:
            mask = ~batch["gt_xy"].isnan().any(2).any(0)
            batch = {k: (v.to(self.cache_device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            batch["loss_mask"] = None if bool(mask.all()) else mask.to(self.cache_device)
            self._cache[i] = batch
        return batch


def get_dataloader(dataset, phase, augment=False, batch_size=8, workers=0, shuffle=False, synthetic_scenes=64,
                   synthetic_peds=0, crop_device=None, cache_device=None):
    """crop_device: a HIP device -> the image crops of on-disk datasets are cut on the GPU from scene images resident in HBM
    (mggan/data_utils/device_crops.py; augmented training items included: the flip / rotation / Lanczos resize of the
    reference's Pillow path is computed per crop, bit-identically); `features` then arrives on that device."""
    assert phase in ("train", "val", "test")
    if dataset != "synthetic":
        from mggan.data_utils.trajectories_scene import TrajectoryDatasetEval, seq_collate_scene

        if phase in ("val", "test") and augment:
            print("No augmentation during validation or testing.")
            augment = False
        # metres per pixel of the crop source per dataset family (data_loaders.py:60-95)
        if dataset == "stanford":
            name, small = "stanford", 0.7
        elif dataset.lower() in ("eth", "hotel", "zara1", "zara2", "univ", "gofp"):
            name, small = dataset.lower(), 0.5
        else:
            raise NotImplementedError("dataset '{}' (on disk: eth, hotel, univ, zara1, zara2, gofp, stanford; or "
                                      "'synthetic')".format(dataset))
        ds = TrajectoryDatasetEval(dataset_name=name, phase=phase, margin_in=16, margin_out=16, load_occupancy=False,
                                   scaling_small=small, data_augmentation=int(augment))
        if crop_device is not None:
            from mggan.data_utils.device_crops import DeviceCropDataset

            from mggan.data_utils.device_crops import DeviceCropLoader

            return DeviceCropLoader(DeviceCropDataset(ds, crop_device), batch_size, shuffle, workers)
        return torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=workers,
                                           collate_fn=seq_collate_scene, drop_last=False)
    n_batches = max(1, synthetic_scenes // max(batch_size, 1))
    ds = SyntheticScenes(n_batches if phase == "train" else max(1, n_batches // 4), batch_size,
                         synthetic_peds if synthetic_peds > 0 else None, seed={"train": 0, "val": 7, "test": 13}[phase],
                         cache_device=cache_device)
    return torch.utils.data.DataLoader(ds, batch_size=None, shuffle=shuffle,
                                       num_workers=0 if cache_device is not None else workers)
