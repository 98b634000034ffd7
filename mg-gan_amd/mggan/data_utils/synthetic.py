"""Synthetic trajectory batches with the reference's batch schema.

Schema follows the collate output of the reference
(/root/reference/mggan/data_utils/trajectories_scene.py:68-78): time-major
`in_xy (8,b,2)`, `in_dxdy (7,b,2)`, `gt_xy (12,b,2)`, `gt_dxdy (12,b,2)`,
`features (b,4,33,33)` and `seq_start_end` = python list of [start,end].
Relative offsets follow trajectories_scene.py:343-367 (encoder sees 7 deltas).
Image crops follow BaseTrajectories.py:269-286: RGB in [-1,1) plus a one-hot
centre channel.  Distributions are the ones fixed in SURVEY.md 8(d).
"""
import torch

OBS_LEN, PRED_LEN, CROP = 8, 12, 33


def scene_sizes(num_scenes, peds_per_scene=None, seed=0, lo=1, hi=6):
    """Fixed n per scene, or ragged sizes in [lo,hi] (always containing a 1)."""
    if peds_per_scene is not None:
        return [int(peds_per_scene)] * num_scenes
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(lo, hi + 1, (num_scenes,), generator=g).tolist()
    n[0] = 1
    return n


def make_batch(sizes, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    b = int(sum(sizes))
    p0 = torch.rand(b, 2, generator=g) * 15.0
    v = torch.randn(b, 2, generator=g) * 0.35
    steps = v[None] + torch.randn(OBS_LEN + PRED_LEN, b, 2, generator=g) * 0.05
    steps[0] = 0.0
    xy = p0[None] + torch.cumsum(steps, 0)
    d = xy[1:] - xy[:-1]
    img = torch.rand(b, 4, CROP, CROP, generator=g) * 2.0 - 1.0
    img[:, 3] = 0.0
    img[:, 3, CROP // 2, CROP // 2] = 1.0
    sse, s = [], 0
    for n in sizes:
        sse.append([s, s + int(n)])
        s += int(n)
    batch = {
        "in_xy": xy[:OBS_LEN].contiguous(),
        "gt_xy": xy[OBS_LEN:].contiguous(),
        "in_dxdy": d[: OBS_LEN - 1].contiguous(),
        "gt_dxdy": d[OBS_LEN - 1:].contiguous(),
        "features": img.contiguous(),
        "seq_start_end": sse,
    }
    if device != "cpu":
        batch = {k: (t.to(device) if torch.is_tensor(t) else t) for k, t in batch.items()}
    return batch
