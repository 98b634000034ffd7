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
        self.atlas = torch.from_numpy(np.concatenate(parts + [np.zeros(8, np.uint8)])).to(self.device)  # (+ spare bytes: a
        #                                                                     pixel is fetched as one dword, csrc/crop.hip)
        # the Lanczos tables of the resize passes, one entry per (source size, output size), appended as they are met; the
        # device copy is refreshed when the pool has grown
        self._pool, self._pool_idx, self._pool_dev, self._pool_len = [], {}, None, 0
        self._stream = None
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

    def _aug_item(self, scene, alpha, flip, resolve=True):
        """The 26 int32 words of csrc/crop.hip's AugItem for one item."""
        off, h, w = self.scene_rec[scene]
        m, (nw, nh) = AG.rotate_matrix(w, h, alpha)
        rot = 0 if (nw, nh) == (w, h) and m[0] == 1.0 and m[1] == 0.0 else 1
        a = AG.affine_fixed(m, nw, nh) if rot else (65536, 0, 0, 0, 65536, 0)
        sw, sh = AG.small_size(nw, nh, self.f_small)
        rec = np.zeros(26, np.int32)
        rec[0:2] = np.array([off], np.int64).view(np.int32)
        rec[2:16] = (w, h, flip, rot, nw, nh, sw, sh) + tuple(a)
        return self._resolve_tables(rec) if resolve else rec

    def _resolve_tables(self, rec):
        """Fill in where the item's Lanczos tables sit in THIS process's pool (loader workers leave it to the main process:
        their pools are copies made at fork time)."""
        nw, nh, sw, sh = (int(v) for v in rec[6:10])
        kh, bh, ksh = self._tables(nw, sw)
        kv, bv, ksv = self._tables(nh, sh)
        if 32 * (nw / max(sw, 1)) + ksh > SPAN_MAX:
            raise NotImplementedError("downscale {} -> {} needs staged rows beyond {} pixels".format(nw, sw, SPAN_MAX))
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
            meta = ("aug", self._aug_item(scene, alpha, flip, resolve=False), centers)
        else:
            off, h, w = self.scene_rec[scene]
            meta = (np.full(end - start, off, np.int64), np.tile(np.array([h, w], np.int32), (end - start, 1)), centers)
        return [obs, xy[:, ds.obs_len:], dxdy[:, :ds.obs_len - 1], dxdy[:, ds.obs_len - 1:], (end - start) * [scene], meta,
                torch.empty(1)]

    def collate_host(self, data):
        """The host half of a batch (runs in loader workers too): the reference's collate of the trajectories, plus what the
        device half needs -- no HIP call."""
        metas = [d[5] for d in data]
        batch = seq_collate_scene([d[:5] + [torch.empty(0)] + d[6:] for d in data])
        # the validity scan of the training loop (abstract_train.py:127-132 of the reference: pedestrians without ground truth)
        # happens here, on the host copy: the batch reaches the trainer on the device, where the scan would be a sync
        valid = ~torch.isnan(batch["gt_xy"]).any(2).any(0)
        batch["loss_mask"] = None if bool(valid.all()) else valid
        batch["_local_mask"] = True  # (sharded training: the ranks still have to agree on the masked / unmasked path)
        if self.aug:
            batch["_crop_meta"] = ("aug", np.stack([m[1] for m in metas]),
                                   np.concatenate([np.full(len(m[2]), i, np.int32) for i, m in enumerate(metas)]),
                                   np.concatenate([m[2] for m in metas]))
        else:
            batch["_crop_meta"] = ("plain", np.concatenate([m[0] for m in metas]), np.concatenate([m[1] for m in metas]),
                                   np.concatenate([m[2] for m in metas]))
        return batch

    def finish(self, batch):
        """The device half, in the process that owns the GPU: one launch cuts every crop of the batch."""
        from mggan.hip.lib import lib

        meta = batch.pop("_crop_meta")
        m = self.ds.margin_in
        # through pinned memory: a copy from pageable memory makes the host wait for the stream -- i.e. for the training
        # iterations queued ahead of it
        up = lambda t: t.pin_memory().to(self.device, non_blocking=True)
        to = lambda a: up(torch.from_numpy(np.ascontiguousarray(a)))
        # on a stream of the loader's own: the crops of batch i + 1 are cut while the training iteration of batch i (queued on
        # the caller's stream just before) still runs -- at 1,280 pedestrians that iteration is a chain of latency-bound
        # launches that leaves most of the chip idle; the caller's stream waits for the loader's below
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device)
        caller = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            for k, v in list(batch.items()):
                if torch.is_tensor(v) and not v.is_cuda:
                    batch[k] = up(v)
            st = torch.cuda.current_stream().cuda_stream
            if meta[0] == "aug":
                recs = np.stack([self._resolve_tables(r.copy()) for r in meta[1]])
                if self._pool_dev is None or self._pool_dev.numel() != self._pool_len:
                    self._pool_dev = torch.from_numpy(np.concatenate(self._pool)).to(self.device)
                items, ped_item, ctr = to(recs), to(meta[2]), to(meta[3])
                n = ped_item.numel()
                out = torch.empty(n, 4, 2 * m + 1, 2 * m + 1, dtype=torch.float32, device=self.device)
                lib.mggan_crop_patches_aug(self.atlas.data_ptr(), items.data_ptr(), self._pool_dev.data_ptr(), ped_item.data_ptr(),
                                           ctr.data_ptr(), n, m, int(recs[:, 16].max()), out.data_ptr(), st)
                self._keep = (items, ped_item, ctr, self._pool_dev)  # alive until the next batch (the launch is asynchronous)
            else:
                off, hw, ctr = to(meta[1]), to(meta[2]), to(meta[3])
                n = off.numel()
                out = torch.empty(n, 4, 2 * m + 1, 2 * m + 1, dtype=torch.float32, device=self.device)
                lib.mggan_crop_patches(self.atlas.data_ptr(), off.data_ptr(), hw.data_ptr(), ctr.data_ptr(), n, m, out.data_ptr(), st)
                self._keep = (off, hw, ctr)
        caller.wait_stream(self._stream)
        for v in list(batch.values()) + [out]:
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(caller)
        batch["features"] = out
        return batch

    def collate(self, data):
        return self.finish(self.collate_host(data))


def _seed_numpy_in_worker(_):
    """torch seeds its own and Python's generator per loader worker, not numpy's -- and the augmentation draws come from
    numpy's global generator (trajectories_scene.py:276-278): without this every worker would repeat the parent's draws."""
    np.random.seed(torch.initial_seed() % (1 << 32))


class DeviceCropLoader:
    """DataLoader whose workers do the host half of every batch (trajectory transforms, crop centres, the augmentation's
    geometry) and whose consumer -- the process that owns the GPU -- does the device half (DeviceCropDataset.finish).
    workers == 0: both halves in this process, the reference's draw order from numpy's global generator."""

    def __init__(self, dds, batch_size, shuffle, workers):
        self.dataset, self.dds = dds, dds
        kw = dict(num_workers=workers, worker_init_fn=_seed_numpy_in_worker, persistent_workers=True, prefetch_factor=4) if workers else {}
        self.loader = torch.utils.data.DataLoader(dds, batch_size=batch_size, shuffle=shuffle, collate_fn=dds.collate_host,
                                                  drop_last=False, **kw)
        self.batch_size = batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            yield self.dds.finish(batch)
