"""get_dataloader (surface of /root/reference/mggan/data_utils/data_loaders.py:10).  The on-disk
datasets (ETH/UCY/SDD text + jpg) are host-side I/O outside this build's scope; the
'synthetic' dataset yields batches with the reference's collate schema."""
import torch

from mggan.data_utils import synthetic


class SyntheticScenes(torch.utils.data.Dataset):
    """Each item is one batch of `scenes_per_batch` scenes (already collated)."""

    def __init__(self, num_batches, scenes_per_batch, peds_per_scene=None, seed=0):
        self.num_batches, self.spb, self.pps, self.seed = num_batches, scenes_per_batch, peds_per_scene, seed
        self.dataset_name = "synthetic"

    def __len__(self):
        return self.num_batches

    def __getitem__(self, i):
        sizes = synthetic.scene_sizes(self.spb, self.pps, seed=self.seed + i)
        return synthetic.make_batch(sizes, seed=self.seed + 1000 + i)


def get_dataloader(dataset, phase, augment=False, batch_size=8, workers=0, shuffle=False, synthetic_scenes=64,
                   synthetic_peds=0):
    if dataset != "synthetic":
        raise NotImplementedError(
            "dataset '{}' needs the reference's on-disk loaders (data/datasets/<name>/...), which are out of scope "
            "for the MI355X hot-path build; use --dataset synthetic".format(dataset))
    n_batches = max(1, synthetic_scenes // max(batch_size, 1))
    ds = SyntheticScenes(n_batches if phase == "train" else max(1, n_batches // 4), batch_size,
                         synthetic_peds if synthetic_peds > 0 else None, seed={"train": 0, "val": 7, "test": 13}[phase])
    return torch.utils.data.DataLoader(ds, batch_size=None, shuffle=shuffle, num_workers=workers)
