"""Sources of randomness for the training steps.

HostRNG   -- the reference's draw order on the HOST generators (torch CPU global generator +
             numpy global), SURVEY App. B.  Seed-comparable with the reference; needs the
             PM-network logits on the host for Categorical sampling (one small D2H per G call).
DeviceRNG -- same distributions drawn on the GPU (torch.cuda generator) with no host sync;
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
    """Everything is drawn by the default CUDA generator (capturable in a HIP graph, no host sync)."""
    on_device = True

    def __init__(self, seed=0):
        torch.cuda.manual_seed(seed)
        self._lens = {}

    def begin_iteration(self):
        """One launch draws the label uniforms of a whole iteration (three labels() calls)."""
        self._pool, self._used = torch.rand(8, device="cuda"), 0
        self._npool = None  # the noise pool is drawn by the first noise() call of the iteration

    _pool, _used = None, 0

    def labels(self):
        """-> ((u, 0.9, 1.0), (u', 0.0, 0.1)): smoothed labels real ~ U(.9,1), fake ~ U(0,.1) as device draws."""
        if self._pool is None or self._used + 2 > self._pool.numel():
            self.begin_iteration()
        u = self._pool[self._used:self._used + 2]
        self._used += 2
        return (u[0:1], 0.9, 1.0), (u[1:2], 0.0, 0.1)

    def noise(self, num_samples, dim, sub_batches, device):
        key = (id(sub_batches), len(sub_batches))
        lens = self._lens.get(key)
        if lens is None or lens[0] is not sub_batches:
            t = torch.tensor([int(e) - int(s) for s, e in sub_batches], device=device)
            total = int(t.sum())
            scene_of = torch.repeat_interleave(torch.arange(len(sub_batches), device=device), t, output_size=total)
            lens = (sub_batches, scene_of)
            if len(self._lens) > 64:
                self._lens.clear()
            self._lens[key] = lens
        per_scene = self._normals(num_samples, len(sub_batches), dim, device)
        return per_scene[:, lens[1]]  # one draw per scene, repeated for its pedestrians (utils.py:160-165)

    _npool, _nused = None, 0

    def _normals(self, k, s, dim, device):
        """(k, s, dim) standard normals out of a per-iteration pool: the three generator calls of an iteration (1,
        num_samples and 1 sample sets) are served by ONE randn launch instead of three."""
        pool = self._npool
        if pool is None or pool.shape[1:] != (s, dim) or pool.device != torch.device(device) or self._nused + k > pool.shape[0]:
            pool = self._npool = torch.randn(max(2 * k + 8, 32), s, dim, device=device)
            self._nused = 0
        out = pool[self._nused:self._nused + k]
        self._nused += k
        return out

    def randn(self, *shape):
        return torch.randn(*shape, device="cuda")

    def sample_generators(self, logits, num_samples):
        """Inverse-CDF categorical sampling in one HIP launch (torch.multinomial costs ~12 tiny kernels)."""
        from mggan.hip.lib import lib

        lg = logits.detach().float().contiguous()
        b, g = lg.shape
        u = torch.rand(b, num_samples, device=lg.device)
        idx = torch.empty(b, num_samples, dtype=torch.int64, device=lg.device)
        lib.mggan_sample_categorical(b, num_samples, g, lg.data_ptr(), u.data_ptr(), idx.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
        return idx


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
