"""Helper functions of the hot path (surface of /root/reference/mggan/utils.py:18-25,
34-39,134-165,234-248).  Random draws stay on the HOST generators (torch CPU / numpy
globals) in the reference's order so that seeded runs are comparable."""
from collections import defaultdict  # noqa: F401  (re-exported like the reference's utils)

import numpy as np
import torch
from torch import nn


def get_gan_label_scalars(smoothness=0.1):
    """The two numpy draws of utils.py:18-25: fake first, then real -> (real, fake)."""
    fake = np.random.uniform(0, smoothness)
    real = np.random.uniform(1 - smoothness, 1.0)
    return float(np.float32(real)), float(np.float32(fake))


def get_gan_labels(shape, smoothness=0.1, device="cpu"):
    real, fake = get_gan_label_scalars(smoothness)
    return torch.full(shape, real, device=device), torch.full(shape, fake, device=device)


def to_numpy(x):
    return x.detach().cpu().numpy()


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def make_mlp(dim_list, activation="relu", batch_norm=False, dropout=0):
    """utils.py:134-149 -- note: a 2-element dim_list yields a single Linear."""
    layers = []
    if len(dim_list) > 2:
        for dim_in, dim_out in zip(dim_list[:-2], dim_list[1:-1]):
            layers.append(nn.Linear(dim_in, dim_out))
            if batch_norm:
                layers.append(nn.BatchNorm1d(dim_out))
            if activation == "relu":
                layers.append(nn.ReLU())
            elif activation == "leaky_relu":
                layers.append(nn.LeakyReLU())
            if dropout > 0:
                layers.append(nn.Dropout(p=dropout))
    layers.append(nn.Linear(dim_list[-2], dim_list[-1]))
    return nn.Sequential(*layers)


def gan_noise(shape, noise_type):
    if noise_type == "gaussian":
        return torch.randn(*shape, device="cpu")
    elif noise_type == "uniform":
        return torch.rand(*shape, device="cpu").sub_(0.5).mul_(2.0)
    raise ValueError('Unrecognized noise type "%s"' % noise_type)


def get_global_noise(dim, sub_batches, noise_type, device=None):
    """One draw per scene (host generator), repeated for the scene's pedestrians."""
    n_scenes = len(sub_batches)
    lens = torch.tensor([int(e) - int(s) for s, e in sub_batches])
    draws = torch.cat([gan_noise((1, dim), noise_type) for _ in range(n_scenes)]) if n_scenes else torch.zeros(0, dim)
    noise = draws.repeat_interleave(lens, dim=0)
    return noise if device is None else noise.to(device)


def get_selection_indices(sampled_gen_idxs):
    """Occurrence offset of every generator id within its row, e.g. [1,2,3,1] -> [0,0,0,1]
    (utils.py:234-248), vectorised."""
    idx = sampled_gen_idxs
    same = idx[:, :, None] == idx[:, None, :]
    lower = torch.tril(torch.ones(idx.shape[1], idx.shape[1], dtype=torch.bool, device=idx.device), -1)
    return (same & lower[None]).sum(-1)


# ---- generator-selection rules of the prediction strategies (model/train.py:291-470 of the reference) ----
def expected_sample_idxs(probs, num):
    """'expected' strategy (train.py:301-338): every generator gets round(p * num) of the `num` predictions, the
    rounding remainder is handed out (or taken back) one by one in descending order of the counts, and the
    predictions are listed round-robin over that order.  probs (b, g) numpy -> (b, num) int64 generator ids."""
    probs = np.asarray(probs)
    expected = np.round(probs * num).astype(int)
    order = np.argsort(-expected, axis=-1)
    missing = num - expected.sum(1)
    for r, miss in enumerate(missing):
        n = abs(int(miss))
        uniq, counts = np.unique(np.tile(order[r], n)[:n], return_counts=True)
        expected[r, uniq] += np.sign(miss) * counts
    assert (expected.sum(1) == num).all()
    out = np.zeros((probs.shape[0], num), dtype=np.int64)
    for r in range(probs.shape[0]):
        left, ids = expected[r].copy(), []
        for _ in range(num):
            for g in order[r]:
                if left[g] > 0:
                    ids.append(g)
                    left[g] -= 1
        out[r] = ids[:num]
    return out


def thresholded_generators(probs, eps):
    """Generators whose probability exceeds eps, per pedestrian; nobody over the threshold -> everybody
    (train.py:372-376,430-433).  probs (b, g) tensor -> bool (b, g)."""
    over = probs > eps
    over[over.sum(1) < 1] = True
    return over


def uniform_sample_idxs(probs, eps, num):
    """'uniform_expected' / 'smart_expected' (train.py:378-399): the generators over the threshold, in descending
    order of probability, repeated until `num` predictions are listed.  -> (gen (b, num), slot (b, num)) int64:
    prediction j of pedestrian r is generator gen[r, j] run on noise sample slot[r, j]."""
    over = thresholded_generators(probs, eps)
    b, g = probs.shape
    gen = torch.zeros(b, num, dtype=torch.int64)
    slot = torch.zeros(b, num, dtype=torch.int64)
    ids = torch.arange(g)
    for r in range(b):
        sel = over[r].cpu()
        order = ids[sel][torch.argsort(-probs[r].cpu()[sel])]
        m = order.numel()
        j = torch.arange(num)
        gen[r], slot[r] = order[j % m], j // m
    return gen, slot
