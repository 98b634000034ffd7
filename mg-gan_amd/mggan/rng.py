"""Sources of randomness for the training steps.

HostRNG   -- the reference's draw order on the HOST generators (torch CPU global generator +
             numpy global), SURVEY App. B.  Seed-comparable with the reference; needs the
             PM-network logits on the host for Categorical sampling (one small D2H per G call).
DeviceRNG -- same distributions drawn on the GPU by one Philox launch per iteration (csrc/rng.hip), no host sync;
             statistically equivalent, not seed-identical.  Used by bench.py (`--rng device`).
ReplayRNG -- replays recorded draws (golden fixtures / parity tests).
"""
import numpy as np
import torch

from mggan.utils import get_gan_label_scalars, get_global_noise


class HostRNG:
    on_device = False

    def labels(self):
        return get_gan_label_scalars()

    def noise(self, num_samples, dim, sub_batches, device):
        return torch.stack([get_global_noise(dim, sub_batches, "gaussian") for _ in range(num_samples)]).to(device)

    def randn(self, *shape):
        return torch.randn(*shape)

    def sample_generators(self, logits, num_samples):
        """Categorical(logits=...).sample((K,)).T on the CPU generator (standard.py:223-224)."""
        lg = logits.detach().float().cpu()
        probs = torch.softmax(lg - lg.logsumexp(-1, keepdim=True), -1)
        return torch.multinomial(probs, num_samples, True)


class DeviceRNG:
    """Every draw of an iteration comes from ONE Philox launch (csrc/rng.hip: mggan_draw_iteration) at the start of
    the iteration: the label uniforms, the per-scene noise vectors of the three generator calls (already repeated for
    the pedestrians of each scene) and the uniforms of the categorical sampler.  No host sync, no ATen kernel;
    the iteration counter lives on the device and is advanced by the kernel, so a captured HIP graph draws fresh
    numbers at every replay.  Seeded from torch's CUDA seed (torch.cuda.manual_seed) at the first draw."""
    on_device = True

    def __init__(self, seed=None):
        if seed is not None:
            torch.cuda.manual_seed(seed)
        self.plan = (1, 20, 1)  # noise sample sets of the D / G / PM generator calls (the trainer sets it from its config)
        self.d_steps = 1        # discriminator steps per iteration (1 + num_unrolling_steps)
        self._state = None
        self._pools = None
        self._labels = self._noise = self._unif = None
        self._lab_used = self._noise_used = self._unif_used = 0
        self._shape = None

    def _ensure_state(self, device):
        if self._state is None or self._state.device != torch.device(device):
            seed = int(torch.cuda.initial_seed())
            # every process of a sharded run is seeded alike (abstract_train.py seeds at import) and numbers its scenes
            # from 0: without the rank in the key all ranks would draw the same noise vectors and label uniforms
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_rank() > 0:
                seed ^= dist.get_rank() * 0x9E3779B97F4A7C15
            seed &= 0x7FFFFFFFFFFFFFFF
            self._state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
            self._ticket = torch.zeros(1, dtype=torch.int32, device=device)
            self._bticket = torch.zeros(1, dtype=torch.int32, device=device)  # mggan_sample_bucket_rows (self-resetting)

    def _draw(self, sub_batches, b, Z, device, sets=None, n_unif=None, n_labels=None):
        from mggan.hip import functions as HF
        from mggan.hip.lib import lib

        self._ensure_state(device)
        nd, K, E = self.plan
        sets = nd * self.d_steps + K + E if sets is None else sets
        n_unif = b * (nd * self.d_steps + K) if n_unif is None else n_unif
        n_labels = 4 * self.d_steps + 4 if n_labels is None else n_labels
        shape = (sets, b, Z, n_unif, n_labels, str(device))
        if self._shape != shape:
            # one pool per batch shape, kept: static addresses for graph replay (a trainer that alternates between batch
            # shapes -- its graph cache -- must find the pool a captured iteration was recorded with untouched)
            if self._pools is None:
                self._pools = HF.BoundedCache(8)
            pool = self._pools.get(shape)
            if pool is None:
                pool = self._pools.put(shape, (torch.empty(max(n_labels, 1), dtype=torch.float32, device=device),
                                               torch.empty(max(sets, 1), b, Z, dtype=torch.float32, device=device),
                                               torch.empty(max(n_unif, 1), dtype=torch.float32, device=device)))
            self._labels, self._noise, self._unif = pool
            self._shape = shape
        elif HF._PIN["on"] and self._pools is not None:
            # same shape as the previous draw, which may have run unpinned (an eager batch beyond the graph cache, a
            # masked batch, validation): a capture about to record pointers into this pool must find it pinned, or the
            # bounded cache may evict -- and free -- it under the graph
            self._pools.get(shape)
        tb = HF.scene_tables(sub_batches, b, device)
        lib.mggan_draw_iteration(self._state.data_ptr(), self._ticket.data_ptr(), n_labels, self._labels.data_ptr(), sets,
                                 b, Z, tb.ped_scene.data_ptr(), self._noise.data_ptr(), n_unif, self._unif.data_ptr(),
                                 HF._s())
        self._lab_used = self._noise_used = self._unif_used = 0
        self._sets, self._nu, self._nl = sets, n_unif, n_labels

    def begin_iteration(self, sub_batches=None, b=None, noise_dim=8, device=None):
        """One launch draws everything the iteration will ask for (on the caller's stream, before any fork)."""
        if sub_batches is None:
            self._shape_stale = True
            return
        self._draw(sub_batches, b, noise_dim, device)

    def labels(self):
        """-> ((u, 0.9, 1.0), (u', 0.0, 0.1)): smoothed labels real ~ U(.9,1), fake ~ U(0,.1) as device draws."""
        if self._labels is None or self._lab_used + 2 > self._nl:
            dev = self._state.device if self._state is not None else torch.device("cuda", torch.cuda.current_device())
            self._ensure_state(dev)
            self._extra_labels = torch.empty(8, dtype=torch.float32, device=dev)
            from mggan.hip import functions as HF
            from mggan.hip.lib import lib

            lib.mggan_draw_iteration(self._state.data_ptr(), self._ticket.data_ptr(), 8, self._extra_labels.data_ptr(), 0, 0,
                                     0, 0, 0, 0, 0, HF._s())
            self._labels, self._lab_used, self._nl = self._extra_labels, 0, 8
            self._shape = None
        u = self._labels[self._lab_used:self._lab_used + 2]
        self._lab_used += 2
        return (u[0:1], 0.9, 1.0), (u[1:2], 0.0, 0.1)

    def noise(self, num_samples, dim, sub_batches, device):
        """(num_samples, b, dim): one N(0,1)^dim draw per (sample, scene), repeated for its pedestrians (utils.py:160-165)."""
        b = max((int(e) for _, e in sub_batches), default=0)
        pool = self._noise
        if (pool is None or pool.shape[1:] != (b, dim) or pool.device != torch.device(device)
                or self._noise_used + num_samples > self._sets):
            # outside a planned iteration (a stand-alone step or prediction call): draw exactly this request
            self._draw(sub_batches, b, dim, device, sets=num_samples, n_unif=b * num_samples, n_labels=8)
            pool = self._noise
        out = pool[self._noise_used:self._noise_used + num_samples]
        self._noise_used += num_samples
        return out

    def randn(self, *shape):
        return torch.randn(*shape, device="cuda")

    def peek_uniforms(self, n, device):
        """The uniforms the NEXT sampling call of n picks will consume (not advanced), or None when that call would draw
        afresh -- sharded training counts the generator step's picks ahead of time (mggan_sample_counts)."""
        if self._unif is not None and self._unif.device == torch.device(device) and self._unif_used + n <= self._nu:
            return self._unif[self._unif_used:self._unif_used + n]
        return None

    def sample_generators(self, logits, num_samples):
        """Inverse-CDF categorical sampling in one HIP launch (torch.multinomial costs ~12 tiny kernels)."""
        from mggan.hip import functions as HF
        from mggan.hip.lib import lib

        lg = logits.detach().float().contiguous()
        b, g = lg.shape
        n = b * num_samples
        if self._unif is not None and self._unif.device == lg.device and self._unif_used + n <= self._nu:
            u = self._unif[self._unif_used:self._unif_used + n]
            self._unif_used += n
        else:
            self._ensure_state(lg.device)
            u = torch.empty(n, dtype=torch.float32, device=lg.device)
            lib.mggan_draw_iteration(self._state.data_ptr(), self._ticket.data_ptr(), 0, 0, 0, 0, 0, 0, 0, n, u.data_ptr(),
                                     HF._s())
        self.last_sample_u = (u.data_ptr(), n)
        idx = torch.empty(b, num_samples, dtype=torch.int64, device=lg.device)
        lib.mggan_sample_categorical(b, num_samples, g, lg.data_ptr(), u.data_ptr(), idx.data_ptr(),
                                     HF._s())
        return idx

    def sample_rows(self, logits, num_samples):
        """sample_generators + the rollout-row tables of the picks (HF.device_rollout_rows) behind one entry: one launch
        instead of two up to 2,048 rows (the single-sample rollouts of the discriminator step).
        -> (generator indexes (b, num_samples) int64, HF.RolloutRows)."""
        from mggan.hip import functions as HF
        from mggan.hip.lib import lib

        lg = logits.detach().float().contiguous()
        b, g = lg.shape
        n = b * num_samples
        self._ensure_state(lg.device)
        if self._unif is not None and self._unif.device == lg.device and self._unif_used + n <= self._nu:
            u = self._unif[self._unif_used:self._unif_used + n]
            self._unif_used += n
        else:
            u = torch.empty(n, dtype=torch.float32, device=lg.device)
            lib.mggan_draw_iteration(self._state.data_ptr(), self._ticket.data_ptr(), 0, 0, 0, 0, 0, 0, 0, n, u.data_ptr(),
                                     HF._s())
        self.last_sample_u = (u.data_ptr(), n)  # (the trainer's pre-counted generator picks must have read these very uniforms)
        idx = torch.empty(b, num_samples, dtype=torch.int64, device=lg.device)
        rows, _ = HF.empty_rollout_rows(b, num_samples, g, lg.device)
        # the block counters of the bucketing pass: ONE zeroed buffer per sampler, left zero by every call (the scatter
        # kernel clears what it read) -- no memset node in front of every call, inside a captured graph either
        need = 16 * ((n + 1023) // 1024)
        blk = self.__dict__.get("_blk")
        if blk is None or blk.device != lg.device or blk.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the sampler's block counters must exist before a capture (run an eager iteration first)")
            blk = self._blk = torch.zeros(max(need, 4096), dtype=torch.int32, device=lg.device)
        lib.mggan_sample_bucket_rows(b, num_samples, g, lg.data_ptr(), u.data_ptr(), idx.data_ptr(), rows.row_gen.data_ptr(),
                                     rows.row_ped.data_ptr(), rows.row_slot.data_ptr(), rows.row_pos.data_ptr(),
                                     rows.inv.data_ptr(), rows.seg.data_ptr(), rows.row_gen_pos.data_ptr(), blk.data_ptr(), 1,
                                     self._bticket.data_ptr(), HF._s())
        return idx, rows


class ReplayRNG:
    """Feeds recorded draws in call order: labels -> list of (real,fake); noise / gen_idxs -> lists."""
    on_device = False

    def __init__(self, labels=(), noise=(), gen_idxs=()):
        self._labels, self._noise, self._idx = list(labels), list(noise), list(gen_idxs)

    def labels(self):
        r, f = self._labels.pop(0)
        return float(r), float(f)

    def noise(self, num_samples, dim, sub_batches, device):
        n = self._noise.pop(0)
        assert n.shape[0] == num_samples and n.shape[-1] == dim, (n.shape, num_samples, dim)
        return n.to(device)

    def randn(self, *shape):
        return torch.randn(*shape)

    def sample_generators(self, logits, num_samples):
        idx = self._idx.pop(0)
        assert idx.shape[1] == num_samples
        return idx
