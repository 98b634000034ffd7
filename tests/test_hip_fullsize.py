"""GPU, BASELINE.json sizes: the HIP path against the CPU oracle on one full D+G+PM iteration at configs[1]
(64 scenes x 20 pedestrians, 4 generators, 20 samples) and configs[2] (256 scenes x 32 pedestrians, 8 generators),
plus size-independent properties at those sizes:
bit-identical repeats (no float atomics, fixed-order reductions) and scene-order equivariance of the forward."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _draws(sizes, g, K, gen):
    b, S = sum(sizes), len(sizes)
    rep = torch.tensor(sizes)
    steps = []
    for k in (1, K, 1):
        noise = torch.randn(k, S, 8, generator=gen).repeat_interleave(rep, dim=1)
        idx = torch.randint(0, g, (b, k), generator=gen)
        steps.append((noise, idx))
    return steps


def _trainers(g, seed=3):
    import mggan_oracle as O
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    cfg = get_parser().parse_args(["--num_gens", str(g)])
    torch.manual_seed(seed)
    G, D = construct_model(cfg)
    Go, Do = O.construct_oracle(g)
    Go.load_state_dict(G.state_dict())
    Do.load_state_dict(D.state_dict())
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    for m in (tr.G, tr.D, Go, Do):
        m.train()
    return tr, O.OracleTrainer(Go, Do, mode="block")


def _iteration(tr, batch, steps, labels):
    from mggan.rng import ReplayRNG

    m = defaultdict(list)
    dbatch = tr.to_device(batch)
    dbatch["loss_mask"] = None
    tr.rng = tr.G.rng = ReplayRNG(labels=[labels[0], labels[1], labels[2]], noise=[s[0] for s in steps],
                                  gen_idxs=[s[1] for s in steps])
    tr.train_iteration(dbatch, m)
    return m


def _snapshot_steps(tr):
    """Wrap the three steps so that the (clipped) gradients each one leaves behind are kept per step."""
    grads = {}
    for name, mod_name in (("discriminator_step", "D"), ("generator_step", "G"), ("net_chooser_step", "G")):
        fn = getattr(tr, name)

        def wrapped(*a, _fn=fn, _name=name, _mod=mod_name, **kw):
            out = _fn(*a, **kw)
            mod = getattr(tr, _mod)
            grads[_name] = {n: q.grad.detach().cpu().clone() for n, q in mod.named_parameters()
                            if id(q) in mod._touched and q.grad is not None}
            return out

        setattr(tr, name, wrapped)
    return grads


# BASELINE.json configs[1] and configs[2] (C2 = 64 x 20, g=4; C3 = 256 x 32, g=8)
FULL_SIZES = [pytest.param(64, 20, 4, id="configs1-64x20-g4"), pytest.param(256, 32, 8, id="configs2-256x32-g8")]
# + BASELINE.json configs[0]'s shape (32 ragged scenes of 1-6 pedestrians incl. n = 1, one generator)
ALL_SIZES = [pytest.param(32, None, 1, id="configs0-32xragged-g1")] + FULL_SIZES
# The ELEMENT-WISE bound of one tensor is stated on top of the f32 oracle's own error: the conv1 weight of the two scene
# CNNs (two coherent sums that cancel behind the train-mode BatchNorm) -- found by shape in the test.
# Every other tensor is held to 1e-3 of its largest entry against f64, as SURVEY A.12 iii says.


def _oracle_iteration(tro, batch, steps, labels, dtype):
    """D, G, PM step of the oracle with the recorded draws -> (metrics, {step: {param: clipped gradient}})."""
    m_cpu, grads = defaultdict(list), {}
    b = batch["in_xy"].shape[1]
    mask = torch.ones(b, dtype=torch.bool)
    cpu_args = tuple(batch[k].to(dtype) for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy")) + (batch["seq_start_end"],)
    for (step, (noise, idx)), lab in zip(zip(("discriminator_step", "generator_step", "net_chooser_step"), steps),
                                         ((labels[0], labels[1]), (labels[2], labels[2]), (labels[2], labels[2]))):
        draws = {"noise": noise.to(dtype), "gen_idxs": idx, "labels": lab[0], "labels1": lab[0], "labels2": lab[1]}
        getattr(tro, step)(*cpu_args, m_cpu, mask, batch["features"].to(dtype), draws=draws)
        ref_mod = tro.D if step == "discriminator_step" else tro.G
        grads[step] = {n: q.grad.detach().clone() for n, q in ref_mod.named_parameters() if q.grad is not None}
    return m_cpu, grads


@pytest.mark.parametrize("scenes,peds,g", ALL_SIZES)
def test_full_size_iteration_matches_oracle(scenes, peds, g):
    """One D+G+PM iteration (the de-duplicated train_iteration path bench.py measures) against the oracle in
    block-diagonal mode: every logged loss rtol 1e-3 and the post-step parameters relL2 1e-3 against the f32 oracle (the
    reference's arithmetic); every parameter gradient of every step per tensor (relL2 1e-3 + element-wise 1e-3 of the
    tensor's largest entry, SURVEY A.12 iii) against the SAME oracle evaluated in f64.  Why f64 for the gradients: at
    these batch sizes the f32 CPU path is itself up to 1.2e-3 away from the exact value of the gradients that pass
    through the train-mode BatchNorm of the scene CNN (measured: conv1 weight, PM step, 8,192 pedestrians: f32 oracle
    vs f64 1.19e-3, HIP vs f64 7.7e-4) -- a comparison of two f32 evaluations at 1e-3 would test the oracle's noise.
    The f32 oracle's own distance to f64 is checked alongside: the HIP path must not be the less accurate of the two
    by more than the tolerance."""
    import copy

    import mggan_oracle as O
    from helpers import assert_grad_close, rel_l2
    from mggan.data_utils import synthetic

    K = 20
    sizes = synthetic.scene_sizes(scenes, peds)
    batch = synthetic.make_batch(sizes, seed=1)
    steps = _draws(sizes, g, K, torch.Generator().manual_seed(5))
    labels = [(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)]
    tr, tro = _trainers(g)
    tro64 = O.OracleTrainer(copy.deepcopy(tro.G).double(), copy.deepcopy(tro.D).double(), mode="block")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gpu_grads = _snapshot_steps(tr)
    m_gpu = _iteration(tr, batch, steps, labels)
    m_cpu, g32 = _oracle_iteration(tro, batch, steps, labels, torch.float32)
    _, g64 = _oracle_iteration(tro64, batch, steps, labels, torch.float64)
    conv1 = [n for n, q in list(tr.G.named_parameters()) + list(tr.D.named_parameters())
             if n.startswith("scene_encoder") and q.dim() == 4 and q.shape[1] == 4]
    assert len(set(conv1)) == 1, conv1  # (the same name in both models: the first convolution of the scene CNN)
    relaxed = set(conv1)
    for step, ref in g64.items():
        got = gpu_grads[step]
        assert set(got) == set(ref), (step, sorted(set(got) ^ set(ref)))
        group = max(float(v.abs().max()) for v in ref.values())
        for n, r in ref.items():
            if float(r.abs().max()) < 1e-4 * group:  # structurally zero up to round-off (conv bias before BatchNorm)
                assert float(got[n].abs().max()) <= 1e-3 * group, (step, n)
                continue
            # per tensor relL2 <= 1e-3; element-wise within 1e-3 of the tensor's largest entry ON TOP OF what the f32 CPU
            # path itself is off by there (the conv1 weight gradient behind the train-mode BatchNorm is two coherent sums
            # that cancel: at 1,280 pedestrians the f32 oracle is 5.6e-4 of the largest entry away from f64 in the
            # PM step, this build 4e-4 ... 1.1e-3 depending on the summation order of the convolutions in front of it)
            ours, theirs = rel_l2(got[n], r), rel_l2(g32[step][n], r)
            assert ours <= 1e-3, (step, n, ours)
            assert ours <= theirs + 1e-3, (step, n, ours, theirs)
            scale = float(r.abs().max())
            el_ours = float((got[n].detach().cpu().double() - r.double()).abs().max())
            el_theirs = float((g32[step][n].double() - r.double()).abs().max()) if n in relaxed else 0.0
            assert el_ours <= el_theirs + 1e-3 * scale, (step, n, el_ours / scale, el_theirs / scale)
    for key, v in m_cpu.items():  # losses: rtol 1e-3 (SURVEY A.12 ii)
        assert abs(m_gpu[key][0] - v[0]) <= 1e-3 * abs(v[0]) + 1e-6, (key, m_gpu[key][0], v[0])
    for mod, ref in ((tr.G, tro.G), (tr.D, tro.D)):  # post-step parameters: relL2 1e-3 (A.12 iv)
        a = torch.cat([p.detach().cpu().flatten() for p in mod.parameters()]).double()
        r = torch.cat([p.detach().flatten() for p in ref.parameters()]).double()
        assert float((a - r).norm() / r.norm()) <= 1e-3


@pytest.mark.parametrize("scenes,peds,g", ALL_SIZES)
def test_full_size_iteration_is_bit_reproducible(scenes, peds, g):
    from mggan.data_utils import synthetic

    K = 20
    sizes = synthetic.scene_sizes(scenes, peds)
    batch = synthetic.make_batch(sizes, seed=2)
    labels = [(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)]
    outs = []
    for _ in range(2):
        tr, _ = _trainers(g, seed=7)
        steps = _draws(sizes, g, K, torch.Generator().manual_seed(9))
        m = _iteration(tr, batch, steps, labels)
        outs.append((torch.cat([tr.G._flat, tr.D._flat]).cpu(), {k: v[0] for k, v in m.items()}))
    assert torch.equal(outs[0][0], outs[1][0])  # every reduction is fixed-order: two runs agree to the bit
    assert outs[0][1] == outs[1][1]


@pytest.mark.parametrize("scenes,peds,g", FULL_SIZES)
def test_generator_is_equivariant_to_scene_order(scenes, peds, g):
    """Scenes are independent in the generator (social attention, noise and rollouts are per scene / per
    pedestrian): feeding the scenes in reverse order permutes the predictions and nothing else."""
    from mggan.data_utils import synthetic
    from mggan.rng import ReplayRNG

    K = 20
    sizes = [peds] * scenes
    batch = synthetic.make_batch(sizes, seed=4)
    tr, _ = _trainers(g, seed=11)
    tr.G.eval()  # BatchNorm statistics of the whole batch do not depend on the order, running stats even less
    b = sum(sizes)
    gen = torch.Generator().manual_seed(13)
    noise = torch.randn(K, len(sizes), 8, generator=gen).repeat_interleave(torch.tensor(sizes), dim=1)
    idx = torch.randint(0, g, (b, K), generator=gen)
    perm = torch.arange(b).view(len(sizes), peds).flip(0).reshape(-1)  # pedestrian order after reversing the scenes
    outs = []
    for order in (None, perm):
        bt = {k: v for k, v in batch.items()}
        n, i = noise, idx
        if order is not None:
            for k in ("in_xy", "in_dxdy"):
                bt[k] = batch[k][:, order]
            bt["features"] = batch["features"][order]
            n, i = noise[:, order], idx[order]
        d = tr.to_device(bt)
        tr.rng = tr.G.rng = ReplayRNG(gen_idxs=[i])
        with torch.no_grad():
            out, logits, _ = tr.G(d["in_xy"], d["in_dxdy"], bt["seq_start_end"], noise=n.cuda(), all_gen_out=False,
                                  img=d["features"], num_samples=K)
        outs.append((out.abs.cpu(), logits.cpu()))
    np.testing.assert_allclose(outs[1][0].numpy(), outs[0][0][:, :, perm].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[1][1].numpy(), outs[0][1][perm].numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("sizes,g", [pytest.param([70, 3, 100, 1, 65], 2, id="scenes-of-65-to-100-peds-g2")])
def test_scenes_larger_than_64_pedestrians_match_oracle(sizes, g):
    """Scenes of more than 64 pedestrians leave the row-structured social kernels (csrc/social_rows.hip walks scenes of up
    to 64) for the per-stage kernels of csrc/social.hip, in the generator's attention AND in the discriminator's
    (/root/reference/mggan/model/modules/social.py:7-123): one full D+G+PM iteration against the oracle -- losses 1e-3,
    every parameter gradient of every step (relL2 and element-wise 1e-3 against the oracle in f64), post-step parameters."""
    import copy

    import mggan_oracle as O
    from helpers import rel_l2
    from mggan.data_utils import synthetic
    from mggan.hip import functions as HF

    K = 20
    assert not HF.SceneTables([[0, 70]], 70, "cpu").rows_ok  # (the premise: such a scene is not a "rows" scene)
    batch = synthetic.make_batch(sizes, seed=6)
    steps = _draws(sizes, g, K, torch.Generator().manual_seed(15))
    labels = [(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)]
    tr, tro = _trainers(g, seed=21)
    tro64 = O.OracleTrainer(copy.deepcopy(tro.G).double(), copy.deepcopy(tro.D).double(), mode="block")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gpu_grads = _snapshot_steps(tr)
    m_gpu = _iteration(tr, batch, steps, labels)
    m_cpu, _ = _oracle_iteration(tro, batch, steps, labels, torch.float32)
    _, g64 = _oracle_iteration(tro64, batch, steps, labels, torch.float64)
    for step, ref in g64.items():
        got = gpu_grads[step]
        assert set(got) == set(ref), (step, sorted(set(got) ^ set(ref)))
        group = max(float(v.abs().max()) for v in ref.values())
        for n, r in ref.items():
            if float(r.abs().max()) < 1e-4 * group:
                assert float(got[n].abs().max()) <= 1e-3 * group, (step, n)
                continue
            assert rel_l2(got[n], r) <= 1e-3, (step, n, rel_l2(got[n], r))
            scale = float(r.abs().max())
            assert float((got[n].detach().cpu().double() - r.double()).abs().max()) <= 1e-3 * scale, (step, n)
    for key, v in m_cpu.items():
        assert abs(m_gpu[key][0] - v[0]) <= 1e-3 * abs(v[0]) + 1e-6, (key, m_gpu[key][0], v[0])
    for mod, ref in ((tr.G, tro.G), (tr.D, tro.D)):
        a = torch.cat([p.detach().cpu().flatten() for p in mod.parameters()]).double()
        r = torch.cat([p.detach().flatten() for p in ref.parameters()]).double()
        assert float((a - r).norm() / r.norm()) <= 1e-3
