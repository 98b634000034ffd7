"""Scene crops on the GPU: the scene images of a dataset stay resident in HBM and the (b,4,33,33) crop tensor
of a batch is produced by one kernel launch (csrc/crop.hip) instead of one PIL crop per pedestrian plus two full-image
resizes per scene on the host (trajectories_scene.py:314-356 of the reference).
  * un-augmented items (validation / test always, training when `--augment` is off): windows of the resident "small" images
    (mggan_crop_patches);
  * training items with `--augment 1` (the reference's default): the image is flipped, rotated (expand, nearest) and
    Lanczos-resized per ITEM by Pillow in the reference (trajectories_scene.py:276-333) -- here the un-augmented scaled image
    stays resident and mggan_crop_patches_aug computes every pedestrian's window directly from it, with the integers Pillow
    would use (mggan/data_utils/aug_geometry.py); the augmentation draws come from numpy's global generator in the
    reference's order (:276-278).
Both are bit-identical to the host path (tests/test_loader.py, tests/test_aug_crops.py)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from mggan.data_utils import aug_geometry as AG
from mggan.data_utils.trajectories_scene import seq_collate_scene

KS_MAX, SPAN_MAX = 128, 800  # csrc/crop.hip: AUG_KS_MAX, AUG_SPAN_MAX


class DeviceCropDataset(Dataset):
    """View of a TrajectoryDatasetEval whose items carry crop centres instead of crops."""

    def __init__(self, ds, device):
        self.ds, self.device = ds, torch.device(device)
        self.aug = bool(ds.data_augmentation and ds.phase == "train")
        parts, self.scene_rec, off = [], {}, 0
        for scene, rec in ds.images.items():
            arr = np.asarray(rec["scaled_image" if self.aug else "small_image"])
            if arr.ndim == 3 and arr.shape[2] == 4:
                raise ValueError("device crops need RGB scene images ({} has an alpha channel)".format(scene))
            assert arr.ndim == 3 and arr.shape[2] == 3 and arr.dtype == np.uint8, (scene, arr.shape, arr.dtype)
            self.scene_rec[scene] = (off, arr.shape[0], arr.shape[1])
            parts.append(arr.reshape(-1))
            off += arr.size
        self.atlas = torch.from_numpy(np.concatenate(parts)).to(self.device)
        # the Lanczos tables of the resize passes, one entry per (source size, output size), appended as they are met; the
        # device copy is refreshed when the pool has grown
        self._pool, self._pool_idx, self._pool_dev, self._pool_len = [], {}, None, 0
        self.f_small = ds.img_scaling / ds.scaling_small
        if self.aug:
            self._prewarm()

    def _prewarm(self, limit=1500):
        """The rotated canvas of a (w, h) image is between min(w, h) and the diagonal wide: with few distinct scene sizes
        (ETH / UCY, GOFP) every Lanczos table training can ask for is built once, here (~0.5 ms each); otherwise as met."""
        sizes = set()
        for _, h, w in self.scene_rec.values():
            lo, hi = min(w, h), int(np.ceil(np.hypot(w, h))) + 1
            sizes.update(range(lo, hi + 1))
        if len(sizes) > limit:
            return
        for n in sorted(sizes):
            out = int(round(n * self.f_small))
            if out >= 1:
                try:
                    self._tables(n, out)
                except NotImplementedError:
                    return

    def _tables(self, in_size, out_size):
        """-> (offset of the coefficient rows, offset of the (first index, taps) pairs, taps per row) in the table pool."""
        hit = self._pool_idx.get((in_size, out_size))
        if hit is None:
            bounds, kk, ks = AG.resample_coeffs(in_size, out_size)
            if ks > KS_MAX:
                raise NotImplementedError("downscale {} -> {}: {} taps per pixel (the device kernel holds {})".format(
                    in_size, out_size, ks, KS_MAX))
            k_off = self._pool_len
            b_off = k_off + kk.size
            self._pool += [kk.reshape(-1), bounds.reshape(-1)]
            self._pool_len = b_off + bounds.size
            hit = self._pool_idx[(in_size, out_size)] = (k_off, b_off, ks)
        return hit

    def _aug_item(self, scene, alpha, flip):
        """The 26 int32 words of csrc/crop.hip's AugItem for one item."""
        off, h, w = self.scene_rec[scene]
        m, (nw, nh) = AG.rotate_matrix(w, h, alpha)
        rot = 0 if (nw, nh) == (w, h) and m[0] == 1.0 and m[1] == 0.0 else 1
        a = AG.affine_fixed(m, nw, nh) if rot else (65536, 0, 0, 0, 65536, 0)
        sw, sh = AG.small_size(nw, nh, self.f_small)
        kh, bh, ksh = self._tables(nw, sw)
        kv, bv, ksv = self._tables(nh, sh)
        if 32 * (nw / max(sw, 1)) + ksh > SPAN_MAX:
            raise NotImplementedError("downscale {} -> {} needs staged rows beyond {} pixels".format(nw, sw, SPAN_MAX))
        rec = np.zeros(26, np.int32)
        rec[0:2] = np.array([off], np.int64).view(np.int32)
        rec[2:16] = (w, h, flip, rot, nw, nh, sw, sh) + tuple(a)
        rec[16:18] = (ksh, ksv)
        rec[18:26] = np.array([kh, bh, kv, bv], np.int64).view(np.int32)
        return rec

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, index):
        ds = self.ds
        start, end = ds.seq_start_end[index]
        scene = ds.scene_list[index]
        alpha, flip = ds.augmentation()  # (two draws from numpy's global generator per training item, else (0, 0))
        xy = torch.from_numpy(ds.transformed_xy(index, alpha, flip, ds.images[scene]["scaled_image"].size)).float()
        dxdy = xy[:, 1:] - xy[:, :-1]
        obs = xy[:, :ds.obs_len]
        centers = ds.crop_center(obs[:, -1].numpy()).astype(np.int32)  # (all pedestrians of the item at once: elementwise)
        if self.aug:
            meta = ("aug", self._aug_item(scene, alpha, flip), centers)
        else:
            off, h, w = self.scene_rec[scene]
            meta = (np.full(end - start, off, np.int64), np.tile(np.array([h, w], np.int32), (end - start, 1)), centers)
        return [obs, xy[:, ds.obs_len:], dxdy[:, :ds.obs_len - 1], dxdy[:, ds.obs_len - 1:], (end - start) * [scene], meta,
                torch.empty(1)]

    def collate(self, data):
        from mggan.hip.lib import lib

        metas = [d[5] for d in data]
        batch = seq_collate_scene([d[:5] + [torch.empty(0)] + d[6:] for d in data])
        if self.aug:
            return self._collate_aug(batch, metas)
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

    def _collate_aug(self, batch, metas):
        from mggan.hip.lib import lib

        if self._pool_dev is None or self._pool_dev.numel() != self._pool_len:
            self._pool_dev = torch.from_numpy(np.concatenate(self._pool)).to(self.device)
        items = torch.from_numpy(np.stack([m[1] for m in metas])).to(self.device)
        ped_item = torch.from_numpy(np.concatenate([np.full(len(m[2]), i, np.int32) for i, m in enumerate(metas)])).to(self.device)
        ctr = torch.from_numpy(np.concatenate([m[2] for m in metas])).to(self.device)
        n, m = ped_item.numel(), self.ds.margin_in
        out = torch.empty(n, 4, 2 * m + 1, 2 * m + 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            lib.mggan_crop_patches_aug(self.atlas.data_ptr(), items.data_ptr(), self._pool_dev.data_ptr(), ped_item.data_ptr(),
                                       ctr.data_ptr(), n, m, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        out.record_stream(torch.cuda.current_stream())
        batch["features"] = out
        self._keep = (items, ped_item, ctr, self._pool_dev)  # alive until the next batch (the launch is asynchronous)
        return batch
