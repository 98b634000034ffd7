"""checkpoints (abstract_train.py:235-296 of the reference: save / load / load_from_path through
meta_tags.csv, strict=False model load, optimizer restore) and the evaluation path that decides which one is "best"."""
import csv
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

from helpers import batch_from, sd_from, assert_close


def test_meta_tags_round_trip_keeps_gpu_string_and_types(tmp_path):
    """CPU: what Experiment.save() writes, read_meta_tags() reads back with the parser's types; `--gpus 0` (the default)
    must not come back as a falsy 0 (the trainer would refuse to start: 'no CPU compute path')."""
    from mggan.abstract_train import read_meta_tags
    from mggan.logging import Experiment
    from mggan.model.config import get_parser

    args = get_parser().parse_args(["--num_gens", "3", "--g_lr", "0.002", "--pool_type", "sgan", "--debug"])
    ex = Experiment(tmp_path, name="exp", version=7)
    ex.argparse(args)
    ex.save()
    cfg = read_meta_tags(os.path.join(ex.get_data_path("exp", 7), "meta_tags.csv"))
    assert cfg.gpus == "0" and isinstance(cfg.gpus, str)
    assert cfg.num_gens == 3 and isinstance(cfg.num_gens, int)
    assert cfg.g_lr == 0.002 and cfg.pool_type == "sgan" and cfg.debug is True
    assert cfg.checkpoint is None and cfg.unconditional is False
    assert vars(cfg).keys() >= vars(args).keys()
    # a file written by test_tube for the reference holds only the flags it knew: the additions keep their defaults
    with open(tmp_path / "old.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["key", "value"])
        for k, v in (("num_gens", 2), ("gpus", 0), ("dataset", "eth")):
            w.writerow([k, v])
    old = read_meta_tags(tmp_path / "old.csv")
    assert old.num_gens == 2 and old.gpus == "0" and old.rng == "device" and old.dataset == "eth"


def test_evaluate_ade_fde_masks_nan_ground_truth_and_scales_pixels():
    """CPU: evaluation.py:43-78 -- pedestrians whose ground truth holds a NaN are dropped (scene bounds shifted) and the
    stanford / gofp metrics are reported in pixels (1 / ratio of the scene)."""
    from mggan.evaluation import adjust_seq_start_end_for_mask, evaluate_ade_fde
    from mggan.metrics import compute_metrics_from_batch

    g = torch.Generator().manual_seed(0)
    N, K = 7, 5
    gt = torch.randn(N, 12, 2, generator=g)
    preds = torch.randn(12, K, N, 2, generator=g).numpy()
    gt[4, 3:, :] = float("nan")  # one pedestrian of the second scene leaves early
    sse = [(0, 3), (3, 6), (6, 7)]
    assert adjust_seq_start_end_for_mask(sse, np.array([0, 0, 0, 0, 1, 0, 0], bool)) == [(0, 3), (3, 5), (5, 6)]

    class DS:
        pred_traj, seq_start_end, dataset_name = gt, sse, "stanford"
        scene_list = ["a", "b", "a"]
        images = {"a": {"ratio": 0.5}, "b": {"ratio": 0.25}}

    got = evaluate_ade_fde(DS, preds, [K])
    keep = [0, 1, 2, 3, 5, 6]
    acc = defaultdict(lambda: np.zeros(2))
    for (s, e), scale in zip([(0, 3), (3, 5), (5, 6)], (2.0, 4.0, 2.0)):
        m = compute_metrics_from_batch(torch.from_numpy(preds[:, :, keep][:, :, s:e]) * scale,
                                       gt[keep][s:e].transpose(0, 1) * scale, [[0, e - s]], mode="raw")
        for k, (v, c) in m.items():
            acc[k] += v, c
    for k, (v, c) in acc.items():
        assert np.isfinite(got["{} k={}".format(k, K)])
        np.testing.assert_allclose(got["{} k={}".format(k, K)], v / c, rtol=1e-6)
    DS.dataset_name = "eth"  # metres stay metres
    plain = evaluate_ade_fde(DS, preds, [K])
    assert plain["ADE k=5"] < got["ADE k=5"]


def _draws(sizes, g, K, gen):
    b, S = sum(sizes), len(sizes)
    rep = torch.tensor(sizes)
    noise, idx = [], []
    for k in (1, K, 1):
        noise.append(torch.randn(k, S, 8, generator=gen).repeat_interleave(rep, dim=1))
        idx.append(torch.randint(0, g, (b, k), generator=gen))
    return noise, idx


def _run(tr, batch, sizes, g, gen, iters):
    from mggan.rng import ReplayRNG

    for _ in range(iters):
        noise, idx = _draws(sizes, g, 20, gen)
        tr.rng = tr.G.rng = ReplayRNG(labels=[(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)], noise=noise, gen_idxs=idx)
        tr.train_iteration(batch, defaultdict(list))


@pytest.mark.gpu
def test_save_load_from_path_resume_is_bit_identical(tmp_path):
    """Train 2 iterations, save(), load_from_path() (through meta_tags.csv), continue: equal to the uninterrupted run to
    the bit -- parameters, BatchNorm buffers, Adam moments and per-tensor step counts all survive the round trip."""
    from mggan.data_utils import synthetic
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    g, sizes = 3, [1, 4, 2, 5, 3]
    args = get_parser().parse_args(["--num_gens", str(g), "--log_dir", str(tmp_path), "--name", "ck"])
    torch.manual_seed(21)
    G, D = construct_model(args)
    logger = Experiment(tmp_path, name="ck", version=3)
    logger.argparse(args)
    tr = PiNetMultiGeneratorGAN(G, D, args, logger)
    logger.save()
    tr.G.train(); tr.D.train()
    batch = tr.to_device(synthetic.make_batch(sizes, seed=2))
    batch["loss_mask"] = None
    gen = torch.Generator().manual_seed(77)
    _run(tr, batch, sizes, g, gen, 2)
    tr.epoch = 5
    tr.save()  # checkpoints/checkpoint_5.pth
    steps_at_save = (tr.optimizerG.seg_step.cpu().tolist(), tr.optimizerD.seg_step.cpu().tolist())
    state = gen.get_state()
    _run(tr, batch, sizes, g, gen, 2)
    want = torch.cat([tr.G._flat, tr.D._flat]).cpu()
    bufs = lambda t: {m + "." + k: v.cpu().clone() for m, mod in (("G", t.G), ("D", t.D)) for k, v in mod.state_dict().items()
                      if "running" in k or "tracked" in k}
    want_buf = bufs(tr)

    version_dir = tmp_path / "ck" / "version_3"
    assert (version_dir / "meta_tags.csv").exists() and (version_dir / "checkpoints" / "checkpoint_5.pth").exists()
    tr2, cfg2 = PiNetMultiGeneratorGAN.load_from_path(version_dir, checkpoint="latest")
    assert cfg2.num_gens == g and cfg2.gpus == "0"
    tr2.G.train(); tr2.D.train()
    assert (tr2.optimizerG.seg_step.cpu().tolist(), tr2.optimizerD.seg_step.cpu().tolist()) == steps_at_save
    assert max(steps_at_save[0]) == 4 and min(steps_at_save[0]) >= 0  # trunk: two Adam steps per iteration (A.7)
    gen2 = torch.Generator()
    gen2.set_state(state)
    _run(tr2, batch, sizes, g, gen2, 2)
    got = torch.cat([tr2.G._flat, tr2.D._flat]).cpu()
    assert torch.equal(got, want)
    for k, v in bufs(tr2).items():
        assert torch.equal(v, want_buf[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["g4", "g1"])
def test_reference_shaped_checkpoint_loads(tmp_path, which):
    """A checkpoint as the REFERENCE writes it -- {generator, discriminator, gen_opt, disc_opt} with torch.optim.AdamW
    state dicts, meta_tags.csv by test_tube -- built from the recorded reference weights after 3 iterations
    (golden G3 / D3) loads with strict=False, restores the optimizers and predicts what the reference predicted."""
    from mggan.model.train import PiNetMultiGeneratorGAN
    from mggan.rng import ReplayRNG

    golden = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_{}.npz".format(which))))
    n_gens = int(which[1:])
    sdG, sdD = sd_from(golden, "G3"), sd_from(golden, "D3")
    version_dir = tmp_path / "ref" / "version_0"
    (version_dir / "checkpoints").mkdir(parents=True)
    with open(version_dir / "meta_tags.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["key", "value"])
        for k, v in (("num_gens", n_gens), ("gpus", "0"), ("dataset", "eth"), ("name", "ref"), ("epochs", 500),
                     ("h_dim", 32), ("gan_type", "mgan"), ("debug", False), ("checkpoint", None)):
            w.writerow([k, v])

    def torch_opt_state(sd):
        """the optimizer state torch.optim.AdamW would hold for these parameters after one step"""
        gg = torch.Generator().manual_seed(5)
        ps = [torch.nn.Parameter(v.clone().float()) for k, v in sd.items() if v.is_floating_point() and "running" not in k]
        opt = torch.optim.AdamW(ps, lr=1e-3, betas=(0.5, 0.999))
        for p in ps:
            p.grad = torch.randn(p.shape, generator=gg)
        opt.step()
        return opt.state_dict()

    # parameters in module order (state_dict order minus buffers); the duplicate gs.i / G_i entries alias one tensor
    def param_sd(mod_sd, names):
        return {k: mod_sd[k] for k in names}

    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model

    G0, D0 = construct_model(get_parser().parse_args(["--num_gens", str(n_gens)]))
    optG = torch_opt_state(param_sd(sdG, [n for n, _ in G0.named_parameters()]))
    optD = torch_opt_state(param_sd(sdD, [n for n, _ in D0.named_parameters()]))
    torch.save({"generator": sdG, "discriminator": sdD, "gen_opt": optG, "disc_opt": optD},
               version_dir / "checkpoints" / "checkpoint_best.pth")

    tr, cfg = PiNetMultiGeneratorGAN.load_from_path(version_dir)  # default: checkpoint "best"
    for (p, o), (i, st) in zip(tr.G._flat_items, sorted(optG["state"].items())):
        n = p.numel()
        assert torch.equal(tr.optimizerG.exp_avg[o:o + n].cpu(), st["exp_avg"].reshape(-1))
        assert torch.equal(tr.optimizerG.exp_avg_sq[o:o + n].cpu(), st["exp_avg_sq"].reshape(-1))
    assert tr.optimizerD.seg_step.cpu().tolist() == [1] * len(tr.D._flat_items)
    bt = batch_from(golden, "cuda")
    tr.rng = tr.G.rng = ReplayRNG(gen_idxs=[torch.from_numpy(golden["e/gen_idxs"].copy())])
    pa, _, probs, _ = tr.predict(bt["in_dxdy"], bt["in_xy"], bt["seq_start_end"], img=bt["features"], num=20,
                                 noise=torch.from_numpy(golden["e/noise"].copy()))
    assert_close(pa, golden["e/abs"], what="predict from a reference-shaped checkpoint")
    np.testing.assert_allclose(probs, golden["e/probs"], rtol=1e-3, atol=1e-5)
