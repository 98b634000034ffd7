"""ADE/FDE evaluation over a dataset (surface of /root/reference/mggan/evaluation.py:43-78).
The reference passes (None, "raw") positionally into (mode, mode_thresh) and cannot run as
written; the intended semantics -- per scene, raw sums accumulated as
(value, count) -- are implemented."""
from collections import defaultdict

import numpy as np
import torch

from mggan.metrics import compute_metrics_from_batch


def evaluate_ade_fde(eval_ds, preds, n_preds_list):
    gt = eval_ds.pred_traj  # (N, pred_len, 2)
    gt = gt.detach().cpu().numpy() if torch.is_tensor(gt) else np.asarray(gt)
    seq_start_end = eval_ds.seq_start_end
    accum = defaultdict(lambda: np.zeros((2,)))
    for start, end in seq_start_end:
        if start == end:
            continue
        for n_preds in n_preds_list:
            m = compute_metrics_from_batch(torch.from_numpy(np.ascontiguousarray(preds[:, :n_preds, start:end])),
                                           torch.from_numpy(gt[start:end]).transpose(0, 1), [[0, end - start]],
                                           mode="raw")
            for key, (value, count) in m.items():
                accum["{} k={}".format(key, n_preds)] += value, count
    return {key: value / count for key, (value, count) in accum.items()}
