"""Other GAN objectives and PM-network targets (--gan_obj LS / MM, --weighting_target l2 / endpoint):
one D+G+PM iteration against golden vectors from the real reference (tests/golden/make_golden_masked.py variants);
the initial state is re-created from the seed (seeded initialisation is bit-identical to the reference's)."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_masked import STEPS, _batch, _check

VARIANTS = ["obj_ls_g2", "obj_mm_g2", "wt_l2_g2", "wt_endpoint_g2", "wt_mgan_g2", "pool_sgan_g2", "discrete_g2",
            "masked_sgan_g2", "narrow_h16_g2", "narrow_h24_d8_g2", "narrow_sgan_g2", "narrow_discrete_g2"]
NARROW = [v for v in VARIANTS if v.startswith("narrow")]


def _load(tag):
    g = dict(np.load(os.path.join(GOLDEN, "golden_{}.npz".format(tag))))
    return g, [str(a) for a in g["meta/args"]]


def _opts(args):
    return {args[i].lstrip("-"): args[i + 1] for i in range(0, len(args), 2)}


@pytest.mark.parametrize("tag", VARIANTS)
def test_oracle_variant_iteration(tag):
    import mggan_oracle as O

    g, args = _load(tag)
    torch.manual_seed(int(g["meta/seed"]))
    np.random.seed(int(g["meta/seed"]) + 1)
    o = _opts(args)
    G, D = O.construct_oracle(int(g["meta/num_gens"]), gan_obj=o.get("gan_obj", "NS"), pool_type=o.get("pool_type", "sways"),
                               experiment=o.get("experiment", "multi_generator"), h_dim=int(o.get("h_dim", 32)),
                               decoder_h_dim=int(o.get("decoder_h_dim", 32)), noise_dim=int(o.get("noise_dim", 8)))
    G.train()
    D.train()
    tr = O.OracleTrainer(G, D, mode="block", gan_obj=o.get("gan_obj", "NS"), weighting_target=o.get("weighting_target", "ml"))
    bt, mask = _batch(g)
    a = (bt["in_xy"], bt["in_dxdy"], bt["gt_xy"][:, mask], bt["gt_dxdy"][:, mask], bt["seq_start_end"])
    m = defaultdict(list)
    for s, fn in STEPS:
        lab = g.get("s_{}/labels".format(s))
        dr = {"noise": torch.from_numpy(g["s_{}/noise".format(s)].copy()),
              "gen_idxs": torch.from_numpy(g["s_{}/gen_idxs".format(s)].copy())}
        if lab is not None:
            dr.update(labels=tuple(lab[0]), labels1=tuple(lab[0]), labels2=tuple(lab[-1]))
        getattr(tr, fn)(*a, m, mask, bt["features"], draws=dr)
    _check(g, m, G, D)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", VARIANTS)
def test_hip_variant_iteration(tag):
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN
    from mggan.rng import ReplayRNG

    g, args = _load(tag)
    cfg = get_parser().parse_args(["--num_gens", str(int(g["meta/num_gens"]))] + args)
    torch.manual_seed(int(g["meta/seed"]))
    np.random.seed(int(g["meta/seed"]) + 1)
    G, D = construct_model(cfg)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    tr.G.train()
    tr.D.train()
    bt, _ = _batch(g, "cuda")
    labels = [tuple(r) for s, _ in STEPS if "s_{}/labels".format(s) in g for r in g["s_{}/labels".format(s)]]
    tr.rng = tr.G.rng = ReplayRNG(labels=labels, noise=[torch.from_numpy(g["s_{}/noise".format(s)].copy()) for s, _ in STEPS],
                                  gen_idxs=[torch.from_numpy(g["s_{}/gen_idxs".format(s)].copy()) for s, _ in STEPS])
    m = defaultdict(list)
    tr.train_iteration(bt, m)
    _check(g, m, tr.G, tr.D)
    if tag in NARROW:  # the padding of the wider kernels' parameters stayed exactly zero through D, G and PM updates
        from mggan.model import widths

        assert widths.padding_is_zero(tr.G) and widths.padding_is_zero(tr.D)


@pytest.mark.parametrize("tag", ["pool_sgan_g2", "discrete_g2"] + NARROW)
def test_variant_state_dict_surface(tag):
    """CPU: the variant models expose the reference's state_dict keys and shapes (checkpoint compatibility)."""
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model

    g, args = _load(tag)
    cfg = get_parser().parse_args(["--num_gens", str(int(g["meta/num_gens"]))] + args)
    G, D = construct_model(cfg)
    for pre, mod in (("G1", G), ("D1", D)):
        ref = {k[len(pre) + 1:]: v.shape for k, v in g.items() if k.startswith(pre + "/")}
        sd = mod.state_dict()
        assert list(sd.keys()) == list(ref.keys())
        for k, shp in ref.items():
            assert tuple(sd[k].shape) == tuple(shp), k


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--pool_type", "sgan"], ["--experiment", "discrete"],
                                   ["--experiment", "discrete", "--pool_type", "sgan"]])
def test_variants_graph_replay_and_prediction_strategies(extra):
    """The variants run through the device-RNG path too: eager iterations, HIP-graph replay, every prediction strategy."""
    import contextlib
    import io

    from mggan.data_utils import synthetic
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    cfg = get_parser().parse_args(["--num_gens", "3", "--rng", "device"] + extra)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = construct_model(cfg)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    tr.G.train()
    tr.D.train()
    batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(12, None, seed=3), seed=5))
    batch["loss_mask"] = None
    tr.defer_metrics = True
    m = defaultdict(list)
    tr.train_iteration(batch, m)
    replay = tr.capture_iteration(batch, warmup=1)
    for _ in range(3):
        replay(m, True)
    torch.cuda.synchronize()
    for k, v in m.items():
        assert np.isfinite(v).all(), k
    assert 0.2 < m["train/discr_loss"][-1] < 3.0
    tr.G.eval()
    tr.D.eval()
    b = batch["in_xy"].shape[1]
    for strat in ("sampling", "expected", "uniform_expected", "smart_expected"):
        with torch.no_grad():
            p = tr.get_predict_func(strat)(batch["in_dxdy"], batch["in_xy"], batch["seq_start_end"], img=batch["features"],
                                           num=6)
        p = p[0] if isinstance(p, tuple) else p
        assert tuple(p.shape) == (12, 6, b, 2) and bool(torch.isfinite(p).all()), strat


@pytest.mark.gpu
def test_narrow_model_graph_replay_and_checkpoint_round_trip():
    """--h_dim 16 --decoder_h_dim 24 on the padded kernels (mggan/model/widths.py): eager iteration, HIP-graph replays, the
    padding stays exactly zero, and model + optimizer state round-trip through the reference-shaped state_dicts."""
    import contextlib
    import io

    from mggan.data_utils import synthetic
    from mggan.logging import Experiment
    from mggan.model import widths
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    def make(seed):
        cfg = get_parser().parse_args(["--num_gens", "3", "--rng", "device", "--h_dim", "16", "--decoder_h_dim", "24"])
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            G, D = construct_model(cfg)
        tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
        tr.G.train()
        tr.D.train()
        tr.defer_metrics = True
        return tr

    tr = make(1)
    batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(12, None, seed=3), seed=5))
    batch["loss_mask"] = None
    m = defaultdict(list)
    tr.train_iteration(batch, m)
    replay = tr.capture_iteration(batch, warmup=1)
    for _ in range(3):
        replay(m, True)
    torch.cuda.synchronize()
    for k, v in m.items():
        assert np.isfinite(v).all(), k
    assert widths.padding_is_zero(tr.G) and widths.padding_is_zero(tr.D)
    sd = {"G": tr.G.state_dict(), "D": tr.D.state_dict(), "oG": tr.optimizerG.state_dict(), "oD": tr.optimizerD.state_dict()}
    assert tuple(sd["G"]["encoder.encoder.weight_hh_l0"].shape) == (64, 16)
    assert tuple(sd["G"]["gs.0.hidden2pos.0.weight"].shape) == (12, 24 + 16)
    assert tuple(sd["D"]["discs.0.0.weight"].shape) == (64, 128)
    tr2 = make(2)
    tr2.train_iteration(batch, defaultdict(list))  # (moments exist before they are overwritten)
    tr2.G.load_state_dict(sd["G"])
    tr2.D.load_state_dict(sd["D"])
    tr2.optimizerG.load_state_dict(sd["oG"])
    tr2.optimizerD.load_state_dict(sd["oD"])
    for a, b in ((tr.G, tr2.G), (tr.D, tr2.D)):
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(p, q), k
    for a, b in ((tr.optimizerG, tr2.optimizerG), (tr.optimizerD, tr2.optimizerD)):
        assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
        assert torch.equal(a.seg_step.cpu(), b.seg_step.cpu())
