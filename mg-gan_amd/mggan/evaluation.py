"""ADE/FDE evaluation over a dataset (surface of /root/reference/mggan/evaluation.py:14-78).
The reference passes (None, "raw") positionally into (mode, mode_thresh) and cannot run as
written; the intended semantics -- per scene, raw sums accumulated as
(value, count) -- are implemented, including its NaN filter (pedestrians whose ground truth holds a NaN
are dropped and the scene bounds shifted, :14-27,47-50) and the metre -> pixel scaling of the
`stanford` / `gofp` datasets by 1 / ratio of the scene (:58-62)."""
from collections import defaultdict

import numpy as np
import torch

from mggan.metrics import compute_metrics_from_batch


def adjust_seq_start_end_for_mask(seq_start_end, remove_mask):
    """Scene bounds after the pedestrians flagged in `remove_mask` are taken out (evaluation.py:14-27)."""
    remove_mask = np.asarray(remove_mask, dtype=bool)
    assert seq_start_end[-1][1] == len(remove_mask)
    offsets = [0] + np.cumsum(remove_mask).tolist()
    new_seq = [(int(start) - offsets[int(start)], int(end) - offsets[int(end)]) for start, end in seq_start_end]
    assert new_seq[-1][1] == int(np.sum(~remove_mask))
    return new_seq


def evaluate_ade_fde(eval_ds, preds, n_preds_list):
    """eval_ds: `.pred_traj (N, pred_len, 2)`, `.seq_start_end`, and for the pixel datasets `.dataset_name`,
    `.scene_list`, `.images[scene]["ratio"]`; preds (pred_len, K, N, 2) numpy."""
    gt = eval_ds.pred_traj  # (N, pred_len, 2)
    gt = gt.detach().cpu().numpy() if torch.is_tensor(gt) else np.asarray(gt)
    preds = preds.detach().cpu().numpy() if torch.is_tensor(preds) else np.asarray(preds)
    pred_mask = np.isnan(gt).any(-1).any(-1)
    start_end = adjust_seq_start_end_for_mask(eval_ds.seq_start_end, pred_mask)
    gt = gt[~pred_mask]
    preds = preds[:, :, ~pred_mask]
    name = getattr(eval_ds, "dataset_name", None)
    accum = defaultdict(lambda: np.zeros((2,)))
    for scene_idx, (start, end) in enumerate(start_end):
        if start == end:
            continue
        scaling = 1.0
        if name in ("stanford", "gofp"):  # convert to pixels for these datasets
            scaling = 1.0 / eval_ds.images[eval_ds.scene_list[scene_idx]]["ratio"]
        for n_preds in n_preds_list:
            m = compute_metrics_from_batch(torch.from_numpy(np.ascontiguousarray(preds[:, :n_preds, start:end])) * scaling,
                                           torch.from_numpy(gt[start:end]).transpose(0, 1) * scaling,
                                           [[0, end - start]], mode="raw")
            for key, (value, count) in m.items():
                accum["{} k={}".format(key, n_preds)] += value, count
    return {key: value / count for key, (value, count) in accum.items()}
