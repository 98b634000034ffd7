"""GPU: ragged batches padded to shape buckets (mggan/abstract_train.py: IterationGraphs; reference loader
/root/reference/mggan/data_utils/trajectories_scene.py:40-78, loop /root/reference/mggan/abstract_train.py:114-168).
The phantom pedestrians behind the real ones must be inert: the padded iteration equals the unpadded one up to the order of
the floating-point sums, and it matches the CPU oracle on the UNPADDED batch like any other iteration."""
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(g, seed=3, oracle=False):
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    cfg = get_parser().parse_args(["--num_gens", str(g)])
    torch.manual_seed(seed)
    G, D = construct_model(cfg)
    tro = None
    if oracle:
        import mggan_oracle as O

        Go, Do = O.construct_oracle(g)
        Go.load_state_dict(G.state_dict())
        Do.load_state_dict(D.state_dict())
        Go.train()
        Do.train()
        tro = O.OracleTrainer(Go, Do, mode="block")
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    tr.G.train()
    tr.D.train()
    return tr, tro


def _draws(sizes, g, K, gen, b_pad=None):
    """Recorded draws of one iteration; with b_pad also their padded twins (phantom columns: any values)."""
    b, S = sum(sizes), len(sizes)
    rep = torch.tensor(sizes)
    plain, padded = [], []
    for k in (1, K, 1):
        noise = torch.randn(k, S, 8, generator=gen).repeat_interleave(rep, dim=1)
        idx = torch.randint(0, g, (b, k), generator=gen)
        plain.append((noise, idx))
        if b_pad is not None:
            padded.append((torch.cat([noise, torch.randn(k, b_pad - b, 8, generator=gen)], 1),
                           torch.cat([idx, torch.randint(0, g, (b_pad - b, k), generator=gen)], 0)))
    return plain, padded


def _run(tr, batch, steps, labels, iters):
    from mggan.rng import ReplayRNG

    out = []
    for it in range(iters):
        m = defaultdict(list)
        tr.rng = tr.G.rng = ReplayRNG(labels=list(labels), noise=[s[0] for s in steps[it]], gen_idxs=[s[1] for s in steps[it]])
        tr.train_iteration(batch, m)
        out.append({k: v[0] for k, v in m.items()})
    torch.cuda.synchronize()
    return out


def _grads(mod):
    return {n: q.grad.detach().double().cpu().clone() for n, q in mod.named_parameters()
            if id(q) in mod._touched and q.grad is not None}


@pytest.mark.parametrize("sizes,g", [([3, 1, 5, 2, 6, 4], 3), ([2, 7, 1, 1, 30, 3, 12, 5, 9, 17, 4], 2),
                                     ([5, 3, 6, 2], 2)])  # (the last one fills its bucket exactly: no phantom pedestrian)
def test_padded_steps_equal_the_unpadded_ones(sizes, g):
    """Every step on its own, from identical weights, once on the ragged batch as it is and once padded to its bucket
    (phantom pedestrians, static scene tables, device-side real counts): the logged losses and EVERY parameter gradient
    agree to the order of the floating-point sums (1e-5; the phantoms are inert).  Then three full iterations: an AdamW step
    turns rounding-level gradient differences of near-zero elements into O(lr) weight differences (BASELINE.md section 4:
    the reference differs from itself the same way when only its thread count changes), so those are held to 1e-3."""
    from mggan.abstract_train import IterationGraphs
    from mggan.data_utils import synthetic
    from mggan.hip import functions as HF
    from mggan.rng import ReplayRNG

    K, iters = 20, 3
    labels = [(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)]
    batch = synthetic.make_batch(sizes, seed=5)
    b = sum(sizes)

    def padded(tr):
        ig = IterationGraphs(tr, pad="on", capture=False)
        tr.padding_ok = lambda: True  # (the replayed draws come from the host: this test drives the padded entry itself)
        key, b_pad, S_pad, max_n = ig.bucket_of(batch)
        dbatch = tr.to_device(batch)
        ent = ig._padded_entry(key, b_pad, S_pad, max_n, dbatch)
        ig._load(ent, dbatch)
        assert b_pad >= b and S_pad > len(sizes) and ent.tables.n_real == b and ent.tables.s_real == len(sizes)
        assert [tuple(x) for x in ent.static["seq_start_end"][:len(sizes)]] == [tuple(x) for x in batch["seq_start_end"]]
        return ent, b_pad

    gen = torch.Generator().manual_seed(8)
    for si, (step, mod_name) in enumerate((("discriminator_step", "D"), ("generator_step", "G"), ("net_chooser_step", "G"))):
        tr_a, _ = _trainer(g)
        tr_b, _ = _trainer(g)
        ent, b_pad = padded(tr_b)
        plain, pad = _draws(sizes, g, K, gen, b_pad)
        out = []
        for tr, bt, draws, dims in ((tr_a, tr_a.to_device(batch), plain, None), (tr_b, ent.static, pad, ent.tables)):
            tr.rng = tr.G.rng = ReplayRNG(labels=list(labels), noise=[draws[si][0]], gen_idxs=[draws[si][1]])
            m = defaultdict(list)
            was = HF.set_pad_dims(dims.dims, dims.b) if dims is not None else HF.set_pad_dims(None)
            try:
                getattr(tr, step)(bt["in_xy"], bt["in_dxdy"], bt["gt_xy"], bt["gt_dxdy"], bt["seq_start_end"], m, None,
                                  bt["features"])
            finally:
                HF.set_pad_dims(*was)
            out.append(({k: v[0] for k, v in m.items()}, _grads(getattr(tr, mod_name))))
        (m_a, g_a), (m_b, g_b) = out
        assert set(m_a) == set(m_b) and set(g_a) == set(g_b), step
        for k, v in m_a.items():
            assert abs(m_b[k] - v) <= 1e-5 * abs(v) + 1e-7, (step, k, m_b[k], v)
        top = max(float(v.abs().max()) for v in g_a.values())
        for n, r in g_a.items():
            if float(r.abs().max()) < 1e-4 * top:  # structurally zero up to round-off (a conv bias before BatchNorm)
                assert float(g_b[n].abs().max()) <= 1e-3 * top, (step, n)
                continue
            scale = float(r.abs().max())
            assert float((g_b[n] - r).abs().max()) <= 1e-5 * scale + 1e-9, (step, n, float((g_b[n] - r).abs().max()), scale)

    tr_a, _ = _trainer(g)
    tr_b, _ = _trainer(g)
    ent, b_pad = padded(tr_b)
    steps = [_draws(sizes, g, K, gen, b_pad) for _ in range(iters)]
    dbatch = tr_a.to_device(batch)
    dbatch["loss_mask"] = None
    m_a = _run(tr_a, dbatch, [s[0] for s in steps], labels, iters)
    m_b = _run(tr_b, ent.static, [s[1] for s in steps], labels, iters)
    for it in range(iters):
        assert set(m_a[it]) == set(m_b[it])
        for k, v in m_a[it].items():
            assert abs(m_b[it][k] - v) <= 1e-3 * abs(v) + 1e-6, (it, k, m_b[it][k], v)
    for a, p in ((tr_a.G, tr_b.G), (tr_a.D, tr_b.D)):
        sa, sp = a.state_dict(), p.state_dict()
        for k in sa:
            if not sa[k].is_floating_point():
                assert int(sa[k]) == int(sp[k]), k  # BatchNorm num_batches_tracked
        x, y = a._flat.double(), p._flat.double()
        assert float((x - y).norm() / x.norm()) <= 1e-3, float((x - y).norm() / x.norm())
        for k in sa:
            if "running_" in k:
                ref = sa[k].cpu().numpy()  # (relative to the tensor: an element near zero has no relative error to speak of)
                np.testing.assert_allclose(sp[k].cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * float(np.abs(ref).max()), err_msg=k)


def test_padded_iteration_matches_the_oracle():
    """The padded iteration against the CPU oracle on the UNPADDED batch: losses 1e-3, post-step parameters relL2 1e-3."""
    from mggan.abstract_train import IterationGraphs
    from mggan.data_utils import synthetic

    sizes, g, K = [4, 1, 6, 2, 3, 5, 1, 2], 4, 20
    labels = [(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)]
    batch = synthetic.make_batch(sizes, seed=9)
    b = sum(sizes)
    tr, tro = _trainer(g, seed=12, oracle=True)
    ig = IterationGraphs(tr, pad="on", capture=False, bucket="pow2")
    tr.padding_ok = lambda: True
    key, b_pad, S_pad, max_n = ig.bucket_of(batch)
    assert b_pad == 32 and b == 24
    plain, padded = _draws(sizes, g, K, torch.Generator().manual_seed(2), b_pad)
    dbatch = tr.to_device(batch)
    ent = ig._padded_entry(key, b_pad, S_pad, max_n, dbatch)
    ig._load(ent, dbatch)
    m_gpu = _run(tr, ent.static, [padded], labels, 1)[0]
    m_cpu = defaultdict(list)
    mask = torch.ones(b, dtype=torch.bool)
    cpu_args = tuple(batch[k] for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy")) + (batch["seq_start_end"],)
    for (step, (noise, idx)), lab in zip(zip(("discriminator_step", "generator_step", "net_chooser_step"), plain),
                                         ((labels[0], labels[1]), (labels[2], labels[2]), (labels[2], labels[2]))):
        draws = {"noise": noise, "gen_idxs": idx, "labels": lab[0], "labels1": lab[0], "labels2": lab[1]}
        getattr(tro, step)(*cpu_args, m_cpu, mask, batch["features"], draws=draws)
    for k, v in m_cpu.items():
        assert abs(m_gpu[k] - v[0]) <= 1e-3 * abs(v[0]) + 1e-6, (k, m_gpu[k], v[0])
    for mod, ref in ((tr.G, tro.G), (tr.D, tro.D)):
        a = torch.cat([p.detach().cpu().flatten() for p in mod.parameters()]).double()
        r = torch.cat([p.detach().flatten() for p in ref.parameters()]).double()
        assert float((a - r).norm() / r.norm()) <= 1e-3


def test_padded_iterations_match_the_reference_golden(golden):
    """The padded path against the REAL reference: the golden run's three D+G+PM iterations on its ragged scenes
    (tests/golden/make_golden.py, recorded draws), executed on the batch padded to its shape bucket -- every logged loss
    within 1e-3, the parameters after iterations 1 and 3 relL2 1e-3, BatchNorm counters exact.  (The phantom pedestrians
    get arbitrary noise and generator ids.)"""
    from helpers import batch_from, rel_l2, sd_from
    from mggan.abstract_train import IterationGraphs
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN
    from mggan.rng import ReplayRNG

    g = int(golden["meta/num_gens"])
    cfg = get_parser().parse_args(["--num_gens", str(g)])
    G, D = construct_model(cfg)
    G.load_state_dict(sd_from(golden, "G0"), strict=True)
    D.load_state_dict(sd_from(golden, "D0"), strict=True)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    tr.G.train()
    tr.D.train()
    assert tr.share_context and tr.share_trunk
    bt = batch_from(golden, "cuda")
    b = bt["in_xy"].shape[1]
    ig = IterationGraphs(tr, pad="on", capture=False)
    tr.padding_ok = lambda: True
    key, b_pad, S_pad, max_n = ig.bucket_of(bt)
    assert b_pad > b
    ent = ig._padded_entry(key, b_pad, S_pad, max_n, bt)
    ig._load(ent, bt)
    gen = torch.Generator().manual_seed(4)
    for it in range(1, 4):
        labels, noise, idxs = [], [], []
        for step in ("d", "g", "pm"):
            p = "s{}_{}".format(it, step)
            lab = golden.get(p + "/labels")
            labels += [] if lab is None else [tuple(r) for r in lab]
            n = torch.from_numpy(golden[p + "/noise"].copy())
            i = torch.from_numpy(golden[p + "/gen_idxs"].copy())
            noise.append(torch.cat([n, torch.randn(n.shape[0], b_pad - b, n.shape[2], generator=gen)], 1))
            idxs.append(torch.cat([i, torch.randint(0, g, (b_pad - b, i.shape[1]), generator=gen)], 0))
        tr.rng = tr.G.rng = ReplayRNG(labels=labels, noise=noise, gen_idxs=idxs)
        m = defaultdict(list)
        tr.train_iteration(ent.static, m)
        for step in ("d", "g", "pm"):
            p = "s{}_{}".format(it, step)
            for k in [k for k in golden if k.startswith(p + "/metric/")]:
                name, ref = k.split("/metric/")[1], float(golden[k])
                assert abs(m[name][0] - ref) <= 1e-3 * abs(ref) + 1e-6, (p, name, m[name][0], ref)
        if it in (1, 3):
            for mod, pre in ((tr.G, "G"), (tr.D, "D")):
                ref = sd_from(golden, pre + str(it))
                sd = {k: v.cpu() for k, v in mod.state_dict().items()}
                fl = [k for k in ref if ref[k].is_floating_point()]
                a = torch.cat([sd[k].flatten().double() for k in fl]).numpy()
                r = torch.cat([ref[k].flatten().double() for k in fl]).numpy()
                assert rel_l2(a, r) <= 1e-3, (pre, it, rel_l2(a, r))
                for k in ref:
                    if not ref[k].is_floating_point():
                        assert int(sd[k]) == int(ref[k]), k
