"""Masked batches (pedestrians with NaN ground truth): one D+G+PM iteration against the golden
vectors the real reference produced (tests/golden/make_golden_masked.py) -- the CPU oracle in both modes, and
the HIP path on the GPU."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import rel_l2, sd_from

STEPS = (("d", "discriminator_step"), ("g", "generator_step"), ("pm", "net_chooser_step"))


@pytest.fixture(scope="module")
def masked():
    return dict(np.load(os.path.join(GOLDEN, "golden_masked_g2.npz")))


def _batch(g, device="cpu"):
    b = {k: torch.from_numpy(g["in/" + k].copy()).to(device) for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy", "features")}
    b["seq_start_end"] = [[int(s), int(e)] for s, e in g["meta/scenes"]]
    return b, torch.from_numpy(g["in/mask"].copy())


def _check(g, metrics, G, D):
    for s, _ in STEPS:
        for k, v in g.items():
            if k.startswith("s_{}/metric/".format(s)) and "probs" not in k:
                name = k.split("/metric/")[1]
                assert abs(metrics[name][0] - float(v)) <= 1e-3 * abs(float(v)) + 1e-6, (name, metrics[name][0], float(v))
    for pre, mod in (("G1", G), ("D1", D)):
        ref = sd_from(g, pre)
        sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        fl = [k for k in ref if ref[k].is_floating_point() and "running" not in k]
        a = torch.cat([sd[k].flatten() for k in fl]).numpy()
        r = torch.cat([ref[k].flatten() for k in fl]).numpy()
        assert rel_l2(a, r) <= 1e-3, (pre, rel_l2(a, r))
        for k in ref:  # BatchNorm running statistics and counters
            if "running_var" in k:
                np.testing.assert_allclose(sd[k].numpy(), ref[k].numpy(), rtol=1e-3, atol=1e-5, err_msg=k)
            elif "running_mean" in k:
                # the conv bias in front of a train-mode BatchNorm has a mathematically zero gradient; what AdamW makes
                # of its rounding noise (|update| <= lr = 1e-3) shifts the batch mean by that much and
                # the running mean by momentum * that: a single forward agrees to 3e-10, an iteration to ~1e-4
                np.testing.assert_allclose(sd[k].numpy(), ref[k].numpy(), rtol=1e-3, atol=3e-4, err_msg=k)
            elif not ref[k].is_floating_point():
                assert int(sd[k]) == int(ref[k]), k


@pytest.mark.parametrize("mode", ["block", "faithful"])
def test_oracle_masked_iteration(masked, mode):
    import mggan_oracle as O

    g = masked
    G, D = O.construct_oracle(int(g["meta/num_gens"]))
    G.load_state_dict(sd_from(g, "G0"))
    D.load_state_dict(sd_from(g, "D0"))
    G.train()
    D.train()
    tr = O.OracleTrainer(G, D, mode=mode)
    bt, mask = _batch(g)
    args = (bt["in_xy"], bt["in_dxdy"], bt["gt_xy"][:, mask], bt["gt_dxdy"][:, mask], bt["seq_start_end"])
    m = defaultdict(list)
    for s, fn in STEPS:
        lab = g.get("s_{}/labels".format(s))
        dr = {"noise": torch.from_numpy(g["s_{}/noise".format(s)].copy()),
              "gen_idxs": torch.from_numpy(g["s_{}/gen_idxs".format(s)].copy())}
        if lab is not None:
            dr.update(labels=tuple(lab[0]), labels1=tuple(lab[0]), labels2=tuple(lab[-1]))
        getattr(tr, fn)(*args, m, mask, bt["features"], draws=dr)
    _check(g, m, G, D)


@pytest.mark.gpu
@pytest.mark.parametrize("poison", [False, True])
def test_hip_masked_iteration(masked, poison):
    """poison=True: every scratch buffer starts as NaN, so reading something no kernel wrote fails the comparison
    (the discriminator's K-times repeated scene list once left the last repetition's pair range unwritten)."""
    from mggan.hip import functions as HF
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN
    from mggan.rng import ReplayRNG

    g = masked
    cfg = get_parser().parse_args(["--num_gens", str(int(g["meta/num_gens"]))])
    G, D = construct_model(cfg)
    G.load_state_dict(sd_from(g, "G0"), strict=True)
    D.load_state_dict(sd_from(g, "D0"), strict=True)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    tr.G.train()
    tr.D.train()
    bt, _ = _batch(g, "cuda")
    labels = [tuple(r) for s, _ in STEPS if "s_{}/labels".format(s) in g for r in g["s_{}/labels".format(s)]]
    tr.rng = tr.G.rng = ReplayRNG(labels=labels, noise=[torch.from_numpy(g["s_{}/noise".format(s)].copy()) for s, _ in STEPS],
                                  gen_idxs=[torch.from_numpy(g["s_{}/gen_idxs".format(s)].copy()) for s, _ in STEPS])
    m = defaultdict(list)
    HF.poison_scratch(poison)
    try:
        tr.train_iteration(bt, m)  # computes the loss mask from the NaNs like abstract_train.py:127-132
        torch.cuda.synchronize()
    finally:
        HF.poison_scratch(False)
    _check(g, m, tr.G, tr.D)
