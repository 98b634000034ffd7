"""ADE / FDE / mode-hit metrics (parity metric, CPU or GPU tensors).
Surface of /root/reference/mggan/metrics.py:6-141."""
import numpy as np
import torch


def min_scene_error(error, seq_start_end):
    """Sum over scenes of the minimum (over predictions) of the scene-summed error; error (k, b)."""
    total = 0
    for start, end in seq_start_end:
        total += error[:, start:end].sum(1).min(0)[0].item()
    return total


def displacement_error(pred_traj, pred_traj_gt, consider_ped=None, mode="sum"):
    loss = torch.sqrt(((pred_traj_gt.permute(1, 0, 2) - pred_traj.permute(1, 0, 2)) ** 2).sum(dim=2)).sum(dim=1)
    if consider_ped is not None:
        loss = loss * consider_ped
    return torch.sum(loss) if mode == "sum" else loss


def final_displacement_error(pred_pos, pred_pos_gt, consider_ped=None, mode="sum"):
    loss = torch.sqrt(((pred_pos_gt - pred_pos) ** 2).sum(dim=1))
    if consider_ped is not None:
        loss = loss * consider_ped
    return loss if mode == "raw" else torch.sum(loss)


def compute_metrics_from_batch(preds, gt, sub_batches, mode="mean", mode_thresh=3.0):
    """preds (pred_len, k, b, 2), gt (pred_len, b, 2) -> {FDE, ADE, Mode}; 'raw' returns (sum, count) pairs."""
    pred_len, k, b, _ = preds.shape
    err = (preds - gt[:, None]).norm(dim=-1)  # (T, k, b)
    ades, fdes = err.sum(0), err[-1]
    metrics = {"FDE": np.array([min_scene_error(fdes, sub_batches), b]),
               "ADE": np.array([min_scene_error(ades, sub_batches), pred_len * b]),
               "Mode": np.array([(fdes.min(0)[0] < mode_thresh).float().sum().item(), b])}
    if mode == "mean":
        return {key: (v / c) for key, (v, c) in metrics.items()}
    return metrics
