"""Prediction strategies (reference model/train.py:291-563): the generator-selection rules on the
CPU against the indices the real reference produced, and the full strategies on the GPU against its trajectories.
Fixtures: tests/golden/golden_strategies_g{1,4}.npz (made by tests/golden/make_golden_strategies.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import assert_close, batch_from, sd_from


@pytest.fixture(scope="module", params=["g4", "g1"])
def strat(request):
    tag = request.param
    return (dict(np.load(os.path.join(GOLDEN, "golden_{}.npz".format(tag)))),
            dict(np.load(os.path.join(GOLDEN, "golden_strategies_{}.npz".format(tag)))))


def test_selection_rules_match_reference(strat):
    from mggan.utils import expected_sample_idxs, get_selection_indices, uniform_sample_idxs

    _, s = strat
    probs = s["expected/probs"]
    idx = expected_sample_idxs(probs, 20)
    assert np.array_equal(idx, s["expected/idx"])
    assert (np.bincount(idx[0], minlength=probs.shape[1]).sum() == 20)
    for name in ("uniform_expected", "smart_expected"):
        gen, slot = uniform_sample_idxs(torch.from_numpy(probs), float(s[name + "/eps"]), 20)
        assert np.array_equal(gen.numpy(), s[name + "/idx"]), name
        # the slot of a prediction is the number of earlier predictions of the same generator
        assert torch.equal(slot, get_selection_indices(gen)), name


@pytest.mark.gpu
def test_strategies_match_reference(strat):
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN
    from mggan.rng import HostRNG, ReplayRNG

    golden, s = strat
    g = int(golden["meta/num_gens"])
    cfg = get_parser().parse_args(["--num_gens", str(g)])
    G, D = construct_model(cfg)
    G.load_state_dict(sd_from(golden, "G0"), strict=True)
    with torch.no_grad():
        G.net_chooser[4].bias.copy_(torch.from_numpy(s["pm_bias"]))
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    bt = batch_from(golden, "cuda")
    args = (bt["in_dxdy"], bt["in_xy"], bt["seq_start_end"])
    dev = lambda k: torch.from_numpy(s[k].copy()).cuda()
    b, K = bt["in_xy"].shape[1], 20
    dummy = lambda n: torch.zeros(b, n, dtype=torch.int64)  # the generator's own (unused) draw in all_gen_out mode

    assert tr.get_predict_func("sampling") == tr.predict
    tr.rng = tr.G.rng = ReplayRNG(gen_idxs=[dummy(K)])
    pa, pr, probs, idx = tr.get_predict_func("expected")(*args, img=bt["features"], num=K, noise=dev("expected/noise"))
    assert np.array_equal(idx, s["expected/idx"])
    np.testing.assert_allclose(probs, s["expected/probs"], rtol=1e-3, atol=1e-5)
    assert_close(pa, s["expected/abs"], what="expected abs")
    assert_close(pr, s["expected/rel"], what="expected rel")

    for name in ("uniform_expected", "smart_expected"):
        tr.rng = tr.G.rng = ReplayRNG(gen_idxs=[dummy(K * g)])
        pa, pr, _, idx = tr.get_predict_func(name)(*args, img=bt["features"], num=K, noise=dev("uniform/noise"))
        assert np.array_equal(idx, s[name + "/idx"]), name
        assert_close(pa, s[name + "/abs"], what=name)
        assert_close(pr, s[name + "/rel"], what=name)

    for name in ("smart_sampling", "uniform_sampling"):
        tr.rng = tr.G.rng = ReplayRNG(gen_idxs=[dummy(K * g), torch.from_numpy(s[name + "/idx"].copy())])
        pa, _, _, idx = tr.get_predict_func(name)(*args, img=bt["features"], num=K, noise=dev("uniform/noise"))
        assert np.array_equal(idx, s[name + "/idx"]), name
        assert_close(pa, s[name + "/abs"], what=name)

    if g == 1:  # the Jacobian estimate of the reference is at rounding level (perturbation sigma^2 = 1e-6): structure only
        tr.rng = tr.G.rng = HostRNG()
        torch.manual_seed(105)
        pa, pr, _, idx = tr.get_predict_func("rejection")(*args, img=bt["features"], num=K, noise=dev("rejection/noise"),
                                                         sigma=1e-3, N=2)
        assert pa.shape == s["rejection/abs"].shape and pr.shape == pa.shape and idx.shape == (b, K)
        full = tr.predict_rejection(*args, img=bt["features"], num=K, noise=dev("rejection/noise"), N=1, debug=True)
        kept = (torch.from_numpy(full[3]) == 0).sum(1)
        assert bool((kept == K).all())  # exactly `num` of the total samples survive the truncation
