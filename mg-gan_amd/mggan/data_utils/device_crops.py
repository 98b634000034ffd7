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

import os

KS_MAX, SPAN_MAX = 128, 800  # csrc/crop.hip: AUG_KS_MAX, AUG_SPAN_MAX
TILE_MODE = os.environ.get("MGGAN_CROP_TILES", "1") != "0"  # whole resized images first when that is less work (A/B knob)
TILE_BANDS_MAX = int(os.environ.get("MGGAN_CROP_BANDS", "4"))  # row bands per tile at most (1: whole tiles)
TILE_WGS = 900  # ... aimed at about this many workgroups per launch (256 CUs x 3-4 resident)
LOADER_STREAM_PRIORITY = int(os.environ.get("MGGAN_LOADER_STREAM_PRIORITY", "0"))
PREFETCH_THREAD = os.environ.get("MGGAN_LOADER_THREAD", "0") == "1"  # one batch ahead on a thread of the loader's own: measured
#   SLOWER (2.34 vs 2.17-2.26 ms per 1,280-pedestrian iteration of train(); the loader alone 565 k vs 725 k pedestrians/s): both
#   threads are Python and share the GIL; off


class _PoolMiss(Exception):
    pass


class _PreBatched:
    """What DeviceCropDataset.__getitems__ returns: a batch that needs no collation any more."""

    def __init__(self, batch):
        self.batch = batch


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
        self._rings, self._pending_events = {}, []
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

    def _staging(self, numel, dtype):
        """A pinned host buffer of `numel` elements from a ring of eight per dtype: the async copy out of a slot has long
        finished when the ring comes round (its event is waited for anyway)."""
        ring = self._rings.setdefault(dtype, {"bufs": [None] * 8, "events": [None] * 8, "i": 0})
        i = ring["i"] = (ring["i"] + 1) % 8
        buf = ring["bufs"][i]
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()
        if buf is None or buf.numel() < numel:
            # (generously: a slot that met a one-element tensor first and the trajectories later would be pinned twice -- and
            #  pinning costs milliseconds)
            buf = ring["bufs"][i] = torch.empty(max(2 * numel, 1 << 18), dtype=dtype).pin_memory()
        ev = ring["events"][i] = torch.cuda.Event()
        self._pending_events.append(ev)
        return buf[:numel]

    def _tables(self, in_size, out_size):
        """-> (offset of the coefficient rows, offset of the (first index, taps) pairs, taps per row) in the table pool."""
        hit = self._pool_idx.get((in_size, out_size))
        if hit is None:
            if getattr(self, "_frozen", False):  # (a slot worker: its pool is a copy made at fork time, the device copy is the
                raise _PoolMiss()                #  parent's -- a table the parent does not have yet is left to the parent)
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

    def __getitems__(self, indices):
        """A whole batch in one pass (torch's DataLoader hands over the batch's indices when a dataset defines this): the
        per-item work of __getitem__ with its numpy / torch calls made once per BATCH -- the host half of a 32-scene batch
        costs ~1 ms instead of 2.2 (55 us of Python per item).  Same arithmetic, element by element, and the augmentation
        draws in item order from numpy's global generator: the batches are bit-identical to the per-item path
        (tests/test_loader.py compares both with the reference's fixtures)."""
        ds = self.ds
        if ds.format not in ("pixel", "meter"):
            raise AssertionError(" Not valid format '{}': 'meters' or 'pixel'".format(ds.format))
        I = len(indices)
        draws = [ds.augmentation() for _ in indices]  # (rotation angle, flip code) per item, in order
        se = [ds.seq_start_end[i] for i in indices]
        n = np.array([e - s for s, e in se], np.int64)
        scenes = [ds.scene_list[i] for i in indices]
        wh = np.array([ds.images[sc]["scaled_image"].size for sc in scenes], np.float64)  # (w, h) per item
        to_orig = np.array([1 / ds.images[sc]["scale_factor"] for sc in scenes], np.float64) if ds.format == "pixel" \
            else np.full(I, float(ds.img_scaling))
        alpha = np.array([d[0] for d in draws], np.float64)
        flip = np.array([d[1] for d in draws], np.int64)
        c = np.array([np.cos(a) for a, _ in draws], np.float64)  # (scalar calls, like rotate(): the vector loop of np.cos
        sn = np.array([np.sin(a) for a, _ in draws], np.float64)  #  is not promised to round like its scalar one)
        center = wh / 2.0
        # the expanded canvas starts at the minimum of the rotated corners -- which trajectories_scene.rotate writes into an
        # INTEGER array (the corners are ints): truncation toward zero, reproduced here
        corners = np.stack([np.zeros_like(wh), np.stack([np.zeros(I), wh[:, 1]], 1), wh, np.stack([wh[:, 0], np.zeros(I)], 1)], 1)
        d = corners - center[:, None, :]
        cr = np.stack([d[..., 0] * c[:, None] + d[..., 1] * sn[:, None] + center[:, None, 0],
                       -d[..., 0] * sn[:, None] + d[..., 1] * c[:, None] + center[:, None, 1]], -1)
        offset = np.trunc(cr).min(axis=1)  # (I, 2)
        xy = np.concatenate([ds.trajectory[s:e] for s, e in se]).copy()  # (N, 20, 2) f64
        rep = lambda a: np.repeat(a, n, axis=0)
        f, w_o, h_o = rep(flip), rep(wh[:, 0] * to_orig), rep(wh[:, 1] * to_orig)
        m1, m2 = f == 1, f == 2
        xy[m1, :, 0] = w_o[m1, None] - xy[m1, :, 0]
        xy[m2, :, 1] = h_o[m2, None] - xy[m2, :, 1]
        cen = rep(center * to_orig[:, None])[:, None, :]  # rotation centre in trajectory units
        cc, ss = rep(c)[:, None], rep(sn)[:, None]
        dd = xy - cen
        out = np.empty_like(xy)
        out[..., 0] = dd[..., 0] * cc + dd[..., 1] * ss + cen[..., 0]
        out[..., 1] = -dd[..., 0] * ss + dd[..., 1] * cc + cen[..., 1]
        out -= rep(offset * to_orig[:, None])[:, None, :]
        # (everything below in numpy, float32 like the per-item path's torch ops -- IEEE elementwise, the same bits: a torch
        #  op on a 50k-element CPU tensor wakes the whole OpenMP pool, milliseconds on a 256-core host)
        xy32 = out.astype(np.float32)
        dxdy = xy32[:, 1:] - xy32[:, :-1]
        centers = ds.crop_center(xy32[:, ds.obs_len - 1]).astype(np.int32)
        bounds = np.concatenate([[0], np.cumsum(n)]).tolist()
        gt_np = xy32[:, ds.obs_len:]
        # the four time-major tensors as slices of ONE array (T_in + T_gt + (T_in - 1) + T_gt, N, 2): one upload in finish()
        parts = (xy32[:, :ds.obs_len], gt_np, dxdy[:, :ds.obs_len - 1], dxdy[:, ds.obs_len - 1:])
        base = torch.from_numpy(np.concatenate([a.transpose(1, 0, 2) for a in parts], 0))
        cuts = np.cumsum([0] + [a.shape[1] for a in parts])
        sl = lambda i: base[int(cuts[i]):int(cuts[i + 1])]
        batch = {"in_xy": sl(0), "gt_xy": sl(1), "in_dxdy": sl(2), "gt_dxdy": sl(3), "_traj_base": base,
                 "size": torch.LongTensor([xy32.shape[0]]),
                 "scene_img": tuple(sc for sc, k in zip(scenes, n) for _ in range(int(k))), "features": torch.empty(1),
                 "occupancy": tuple(torch.empty(1) for _ in indices),
                 "seq_start_end": [[a, b] for a, b in zip(bounds, bounds[1:])]}
        valid = ~np.isnan(gt_np).any(axis=(1, 2))
        batch["loss_mask"] = None if bool(valid.all()) else torch.from_numpy(valid)
        batch["_local_mask"] = True
        ped_item = np.repeat(np.arange(I, dtype=np.int32), n)
        if self.aug:
            recs = np.stack([self._aug_item(sc, a, fl, resolve=False) for sc, (a, fl) in zip(scenes, draws)])
            batch["_crop_meta"] = ("aug", recs, ped_item, centers)
        else:
            rec = np.array([self.scene_rec[sc] for sc in scenes], np.int64)  # (off, h, w)
            batch["_crop_meta"] = ("plain", rec[ped_item, 0], rec[ped_item, 1:].astype(np.int32), centers)
        return [_PreBatched(batch)]

    def collate_host(self, data):
        """The host half of a batch (runs in loader workers too): the reference's collate of the trajectories, plus what the
        device half needs -- no HIP call."""
        if len(data) == 1 and isinstance(data[0], _PreBatched):
            return data[0].batch  # (__getitems__ built the batch already)
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

    def plan(self, meta):
        """The host arithmetic of the device half (numpy only; slot workers run it too): which launches the batch needs and
        the int32 arrays they read -> {"kind", "arrays": [...], scalars}."""
        m = self.ds.margin_in
        if meta[0] != "aug":
            return {"kind": "plain", "arrays": [meta[1], meta[2], meta[3]], "n": int(meta[1].shape[0])}
        recs = np.stack([self._resolve_tables(r.copy()) for r in meta[1]])
        n = int(meta[2].shape[0])
        # the pedestrians of an item share its canvas, and their windows overlap: when the 33 x 33 tiles of the items'
        # whole resized images are fewer than the windows, the images are computed once (tile by tile, the same
        # kernel) and the crops are cut from them like un-augmented ones
        sw, sh = recs[:, 8].astype(np.int64), recs[:, 9].astype(np.int64)
        tx, ty = (sw + 32) // 33, (sh + 32) // 33
        if m == 16 and int((tx * ty).sum()) < n and TILE_MODE:
            small_off = np.concatenate([[0], np.cumsum(sw * sh * 3)[:-1]]).astype(np.int64)
            t_item = np.repeat(np.arange(len(recs), dtype=np.int32), tx * ty)
            t_ctr = np.concatenate([np.stack([16 + 33 * (np.arange(a * b) % a), 16 + 33 * (np.arange(a * b) // a)], 1)
                                    for a, b in zip(tx, ty)]).astype(np.int32)
            pi = meta[2]
            # few tiles (a strong downscale: each is a long serial job): split them into row bands until the launch has a few
            # workgroups per CU -- a band recomputes the source rows its taps share with its neighbours (+30 % work at three
            # bands and scale 10), the launch is 2x shorter
            nb = int(min(TILE_BANDS_MAX, max(1, TILE_WGS // max(len(t_item), 1))))
            rows = 33 // nb
            lo = t_ctr[:, 1] - 16
            t_rows = np.stack([np.stack([lo + k * rows, (lo + (k + 1) * rows) if k < nb - 1 else lo + 33], 1) for k in range(nb)], 1)
            t_item, t_ctr, t_rows = np.repeat(t_item, nb), np.repeat(t_ctr, nb, axis=0), t_rows.reshape(-1, 2).astype(np.int32)
            return {"kind": "tiles", "n": n, "tiles": int(t_item.shape[0]), "max_taps": int(recs[:, 16].max()),
                    "small_bytes": int((sw * sh * 3).sum()) + 8,
                    "arrays": [recs, t_item, t_ctr, t_rows, small_off, small_off[pi], np.stack([sh[pi], sw[pi]], 1).astype(np.int32),
                               meta[3]]}
        return {"kind": "windows", "n": n, "max_taps": int(recs[:, 16].max()), "arrays": [recs, meta[2], meta[3]]}

    def finish(self, batch, join=True):
        """The device half, in the process that owns the GPU: one launch cuts every crop of the batch.  join=False (a
        prefetching thread): the consumer orders its stream behind the loader's itself (join_stream)."""
        from mggan.hip.lib import lib

        plan = batch.pop("_crop_plan", None)
        meta = batch.pop("_crop_meta", None)
        if plan is None:
            plan = self.plan(meta)
        m = self.ds.margin_in
        # through pinned memory (a copy from pageable memory makes the host wait for the stream -- i.e. for the training
        # iterations queued ahead of it), and as few copies as possible: every int32 array of the batch rides in ONE upload
        # (pack), the four trajectory tensors in another
        def pack(arrays):
            """[int32 / int64 arrays] -> device views of one uploaded buffer (8-byte aligned pieces)."""
            flat = [np.ascontiguousarray(a).reshape(-1).view(np.int32) for a in arrays]
            offs = np.cumsum([0] + [(f.size + 1) // 2 * 2 for f in flat])
            stage = self._staging(int(offs[-1]), torch.int32)
            buf = stage.numpy()
            for f, o in zip(flat, offs):
                buf[o:o + f.size] = f
            dev = stage.to(self.device, non_blocking=True)
            return [dev[int(o):int(o) + f.size] for f, o in zip(flat, offs)], dev

        def up(t):
            """A host tensor through this loader's pinned staging ring (filled with numpy: torch's own copy of a 100k-element
            CPU tensor wakes its whole thread pool -- milliseconds on a 256-core host)."""
            stage = self._staging(t.numel(), t.dtype)
            np.copyto(stage.numpy().reshape(tuple(t.shape)), t.numpy())
            return stage.view(t.shape).to(self.device, non_blocking=True)
        # on a stream of the loader's own: the crops of batch i + 1 are cut while the training iteration of batch i (queued on
        # the caller's stream just before) still runs -- at 1,280 pedestrians that iteration is a chain of latency-bound
        # launches that leaves most of the chip idle; the caller's stream waits for the loader's below
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device, priority=LOADER_STREAM_PRIORITY)
        caller = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            base = batch.pop("_traj_base", None)
            if base is not None:  # (__getitems__ laid the four trajectory tensors out in one array: one copy)
                both, t0 = up(base), 0
                for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
                    batch[k] = both[t0:t0 + batch[k].shape[0]]
                    t0 += batch[k].shape[0]
            for k, v in list(batch.items()):
                if torch.is_tensor(v) and not v.is_cuda:
                    batch[k] = up(v)
            st = torch.cuda.current_stream().cuda_stream
            n = plan["n"]
            out = torch.empty(n, 4, 2 * m + 1, 2 * m + 1, dtype=torch.float32, device=self.device)
            if plan["kind"] != "plain" and (self._pool_dev is None or self._pool_dev.numel() != self._pool_len):
                self._pool_dev = torch.from_numpy(np.concatenate(self._pool)).to(self.device)
            if plan["kind"] == "tiles":
                small = torch.empty(plan["small_bytes"], dtype=torch.uint8, device=self.device)
                (items, d_ti, d_tc, d_tr, d_so, off, hw, ctr), dev = pack(plan["arrays"])
                lib.mggan_aug_small_images(self.atlas.data_ptr(), items.data_ptr(), self._pool_dev.data_ptr(), d_ti.data_ptr(),
                                           d_tc.data_ptr(), d_tr.data_ptr(), plan["tiles"], plan["max_taps"], d_so.data_ptr(),
                                           small.data_ptr(), st)
                lib.mggan_crop_patches(small.data_ptr(), off.data_ptr(), hw.data_ptr(), ctr.data_ptr(), n, m, out.data_ptr(), st)
                self._keep = (dev, small, self._pool_dev)
            elif plan["kind"] == "windows":
                (items, ped_item, ctr), dev = pack(plan["arrays"])
                lib.mggan_crop_patches_aug(self.atlas.data_ptr(), items.data_ptr(), self._pool_dev.data_ptr(),
                                           ped_item.data_ptr(), ctr.data_ptr(), n, m, plan["max_taps"], out.data_ptr(), st)
                self._keep = (dev, self._pool_dev)  # alive until the next batch (the launch is asynchronous)
            else:
                (off, hw, ctr), dev = pack(plan["arrays"])
                lib.mggan_crop_patches(self.atlas.data_ptr(), off.data_ptr(), hw.data_ptr(), ctr.data_ptr(), n, m, out.data_ptr(), st)
                self._keep = (dev,)
        batch["features"] = out
        for ev in self._pending_events:  # (the staging slots used by this batch are free once the loader stream gets here)
            ev.record(self._stream)
        self._pending_events = []
        if join:
            self.join_stream(batch, caller)
        else:
            ev = torch.cuda.Event()
            ev.record(self._stream)
            batch["_ready"] = ev
        return batch

    # -- slot workers (--workers N): the host half in other processes, handed over through shared memory -------------------
    def to_slot(self, batch, slot):
        """(worker) Write the arrays of a host batch into the shared slot (a uint8 numpy view); -> the picklable skeleton
        (array descriptors instead of arrays) the parent rebuilds the batch from, or None when the slot is too small."""
        meta = batch.pop("_crop_meta")
        try:
            plan = self.plan(meta)
        except _PoolMiss:
            plan = None
        pos = [0]

        def put(a):
            a = np.ascontiguousarray(a)
            o = (pos[0] + 7) // 8 * 8
            if o + a.nbytes > slot.size:
                raise MemoryError
            slot[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
            pos[0] = o + a.nbytes
            return ("__arr__", o, a.shape, a.dtype.str)

        skel = {}
        try:
            for k, v in batch.items():
                if k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
                    skel[k] = ("__rows__", int(v.shape[0]))  # (slices of _traj_base, re-cut by the parent)
                elif torch.is_tensor(v):
                    skel[k] = ("__tensor__",) + put(v.numpy())[1:]
                elif k == "occupancy":
                    skel[k] = ("__empties__", len(v))
                else:
                    skel[k] = v
            if plan is not None:
                skel["_crop_plan"] = dict(plan, arrays=[put(a) for a in plan["arrays"]])
            else:
                skel["_crop_meta"] = tuple(put(a) if isinstance(a, np.ndarray) else a for a in meta)
        except MemoryError:
            return None
        return skel

    @staticmethod
    def from_slot(skel, slot):
        """(parent) The batch of a skeleton; its arrays are views of the slot (finish copies them out at once)."""
        def arr(d):
            _, o, shape, dt = d
            dt = np.dtype(dt)
            n = int(np.prod(shape)) * dt.itemsize
            return slot[o:o + n].view(dt).reshape(shape)

        is_arr = lambda v: isinstance(v, tuple) and len(v) == 4 and v[0] in ("__arr__", "__tensor__")
        batch = {}
        for k, v in skel.items():
            if is_arr(v):
                batch[k] = torch.from_numpy(arr(v)) if v[0] == "__tensor__" else arr(v)
            elif isinstance(v, tuple) and v and v[0] == "__empties__":
                batch[k] = tuple(torch.empty(1) for _ in range(v[1]))
            else:
                batch[k] = v
        base, t0 = batch["_traj_base"], 0
        for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
            rows = skel[k][1]
            batch[k] = base[t0:t0 + rows]
            t0 += rows
        if "_crop_plan" in batch:
            batch["_crop_plan"] = dict(batch["_crop_plan"], arrays=[arr(d) for d in batch["_crop_plan"]["arrays"]])
        else:
            batch["_crop_meta"] = tuple(arr(a) if is_arr(a) else a for a in batch["_crop_meta"])
        return batch

    def join_stream(self, batch, caller=None):
        """The consuming stream waits for the loader stream's work on this batch."""
        caller = caller or torch.cuda.current_stream(self.device)
        ev = batch.pop("_ready", None)
        if ev is not None:
            caller.wait_event(ev)
        else:
            caller.wait_stream(self._stream)
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(caller)
        return batch

    def collate(self, data):
        return self.finish(self.collate_host(data))


SLOT_BYTES = int(os.environ.get("MGGAN_LOADER_SLOT_MB", "8")) << 20
SLOTS_PER_WORKER = 2
NEXT_EPOCH_AHEAD = os.environ.get("MGGAN_LOADER_NEXT_EPOCH", "1") != "0"


def _slot_worker(dds, conn, slots, seed):
    """A loader worker process: the host half of the batches it is sent (indices -> trajectories, crop centres, the
    augmentation's geometry, the launch plan), written into a shared-memory slot; only a small skeleton travels through the
    pipe.  It never touches the GPU."""
    try:
        np.random.seed(seed)
        torch.set_num_threads(1)
        dds._frozen = True
        views = [t.numpy() for t in slots]
        while True:
            try:
                msg = conn.recv()
            except (EOFError, KeyboardInterrupt):
                break
            if msg is None:
                break
            slot_id, indices = msg
            try:
                out = dds.to_slot(dds.__getitems__(indices)[0].batch, views[slot_id])
            except BaseException as exc:  # noqa: BLE001  (re-raised in the parent)
                out = exc
            conn.send((slot_id, out))
    finally:
        os._exit(0)  # (no interpreter teardown: the fork inherited handles of the parent's device tensors and must not free them)


class DeviceCropLoader:
    """Loader of device-cropped batches.  workers == 0 (the reference's default): both halves of a batch in this process,
    the augmentation draws in the reference's order from numpy's global generator.  workers > 0 (--workers N): the host half
    (~1.6 ms of Python / numpy per 1,280-pedestrian batch) runs in N forked processes that write the batch into shared-
    memory slots; this process -- the one that owns the GPU -- copies a slot into pinned memory, uploads it and launches the
    crop kernel (~0.3 ms).  (torch's DataLoader workers pickle every tensor through a pipe: measured SLOWER than no workers.)
    Batches arrive in sampler order; each worker draws its augmentations from a numpy generator seeded from the parent's."""

    def __init__(self, dds, batch_size, shuffle, workers):
        self.dataset, self.dds = dds, dds
        self.loader = torch.utils.data.DataLoader(dds, batch_size=batch_size, shuffle=shuffle, collate_fn=dds.collate_host,
                                                  drop_last=False)
        self.batch_size = batch_size
        self.n_workers = int(workers)
        self._procs, self._ahead, self._rr = None, None, 0

    def _start_workers(self):
        import atexit
        import multiprocessing as mp

        ctx = mp.get_context("fork")
        self._procs = []
        for _ in range(self.n_workers):
            slots = [torch.empty(SLOT_BYTES, dtype=torch.uint8).share_memory_() for _ in range(SLOTS_PER_WORKER)]
            parent, child = ctx.Pipe()
            pr = ctx.Process(target=_slot_worker, args=(self.dds, child, slots, int(np.random.randint(0, 2 ** 31 - 1))),
                             daemon=True, name="mggan-loader-worker")
            pr.start()
            child.close()
            self._procs.append({"proc": pr, "conn": parent, "slots": slots, "views": [t.numpy() for t in slots], "free":
                                list(range(SLOTS_PER_WORKER))})
        atexit.register(self.close)

    def close(self):
        for w in self._procs or []:
            try:
                w["conn"].send(None)
                w["conn"].close()
            except (OSError, BrokenPipeError):
                pass
        for w in self._procs or []:
            w["proc"].join(timeout=2)
            if w["proc"].is_alive():
                w["proc"].terminate()
        self._procs, self._ahead = None, None

    def _iter_workers(self):
        from collections import deque

        if self._procs is None:
            self._start_workers()
        # (sampler iterator, requests in flight) of the epoch being produced; the FOLLOWING epoch's first requests go out as
        # soon as this epoch's sampler is exhausted, so that an epoch does not start with the workers' latency (~2 ms)
        batches, pending = self._ahead or (iter(self.loader.batch_sampler), deque())
        self._ahead = None
        nxt = None

        def fill(it, pend):
            """Send requests while the next worker in turn has a free slot; -> False once `it` is exhausted."""
            while self._procs[self._rr]["free"]:
                try:
                    idx = next(it)
                except StopIteration:
                    return False
                w = self._procs[self._rr]
                sid = w["free"].pop()
                w["conn"].send((sid, list(idx)))
                pend.append((self._rr, sid, idx))
                self._rr = (self._rr + 1) % len(self._procs)
            return True

        done = False
        try:
            while True:
                if not done:
                    done = not fill(batches, pending)
                if done and NEXT_EPOCH_AHEAD:
                    if nxt is None:
                        nxt = (iter(self.loader.batch_sampler), deque())
                    fill(*nxt)
                if not pending:
                    self._ahead, nxt = nxt, None
                    return
                wi, sid, idx = pending.popleft()
                w = self._procs[wi]
                while not w["conn"].poll(1.0):  # (a worker that died -- killed for memory, say -- must not hang the loop)
                    if not w["proc"].is_alive():
                        raise RuntimeError("loader worker {} exited with code {}".format(wi, w["proc"].exitcode))
                got_sid, skel = w["conn"].recv()
                assert got_sid == sid
                if isinstance(skel, BaseException):
                    w["free"].append(sid)
                    raise skel
                if skel is None:  # the batch does not fit a slot: its host half here, in this process
                    batch = self.dds.finish(self.dds.collate_host(self.dds.__getitems__(list(idx))))
                else:
                    batch = self.dds.finish(self.dds.from_slot(skel, w["views"][sid]))  # (copies the slot out: it is free again)
                w["free"].append(sid)
                yield batch
        finally:
            # an abandoned epoch: drain what the workers still owe so that the next one starts clean
            for pend in (pending, nxt[1] if nxt is not None else ()):
                for wi, sid, _ in pend:
                    try:
                        self._procs[wi]["conn"].recv()
                        self._procs[wi]["free"].append(sid)
                    except (EOFError, OSError, TypeError):
                        pass

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if self.n_workers > 0:
            yield from self._iter_workers()
            return
        if not PREFETCH_THREAD:
            for batch in self.loader:
                yield self.dds.finish(batch)
            return
        # one batch ahead on a thread of the loader's own: its ~1.6 ms of Python / numpy per 32-scene batch run while the
        # training thread launches the previous iteration (which needs the GIL for a fraction of that); the draws stay in
        # order (one producer), the consumer orders its stream behind the loader's
        import queue
        import threading

        q, stop, end = queue.Queue(maxsize=2), threading.Event(), object()

        def produce():
            try:
                with torch.cuda.device(self.dds.device):
                    for batch in self.loader:
                        item = self.dds.finish(batch, join=False)
                        while not stop.is_set():
                            try:
                                q.put(item, timeout=0.1)
                                break
                            except queue.Full:
                                pass
                        if stop.is_set():
                            return
                item = end
            except BaseException as exc:  # noqa: BLE001  (re-raised in the consumer)
                item = exc
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return
                except queue.Full:
                    pass

        th = threading.Thread(target=produce, daemon=True, name="mggan-loader")
        th.start()
        try:
            while True:
                item = q.get()
                if item is end:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self.dds.join_stream(item)
        finally:
            stop.set()
