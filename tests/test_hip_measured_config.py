"""GPU: the configuration bench.py TIMES -- hipGraph replay + device RNG + branch streams at BASELINE configs[1]
(64 scenes x 20 pedestrians, 4 generators) and configs[2] (256 x 32, 8 generators) -- tied to the oracle directly:

  (1) graph replays == eager launches, bit for bit, at the full sizes (the same Philox state), with the kernels that are
      only selected at these sizes named one by one from the library's launch log;
  (2) the draws a REPLAY made on the device (label uniforms, per-scene noise, the sampled generator indexes) fed into the CPU
      oracle, started from the very weights / Adam moments / BatchNorm statistics the replay started from: logged losses
      and post-step parameters at 1e-3;
  (3) every C-ABI entry of the iteration maps to a kernel SYMBOL (bench.py's roofline block names one of them).

Reference loop body: /root/reference/mggan/abstract_train.py:114-168."""
import os
from collections import defaultdict

import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [pytest.param(64, 20, 4, id="configs1-64x20-g4"), pytest.param(256, 32, 8, id="configs2-256x32-g8")]


def _trainer(g, seed):
    import bench

    tr = bench.build_trainer(g, "device", torch.device("cuda", 0))
    torch.cuda.manual_seed(seed)
    tr.defer_metrics = True
    tr.zero_grads_in_step = True
    return tr


def _batch(tr, scenes, peds):
    from mggan.data_utils import synthetic

    batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(scenes, peds), seed=0))
    batch["loss_mask"] = None
    return batch


def _weights(tr):
    torch.cuda.synchronize()
    return torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu()


@pytest.mark.parametrize("scenes,peds,g", SIZES)
def test_graph_replay_is_bit_identical_to_eager_at_the_measured_sizes(scenes, peds, g):
    """1 eager + capture + 3 replays against 4 eager iterations from the same Philox state: identical weights, identical
    Adam moments.  The captured iteration's launch log must contain the kernels whose selection thresholds only these
    sizes cross (measurement knobs)."""
    from mggan.hip.lib import launches_of

    iters = 4
    tr = _trainer(g, 4242)
    batch = _batch(tr, scenes, peds)
    m = defaultdict(list)
    for _ in range(iters):
        tr.train_iteration(batch, m)
    tr.flush_metrics()
    ref = _weights(tr)
    ref_m = torch.cat([tr.optimizerG.exp_avg, tr.optimizerD.exp_avg]).cpu()
    assert bool(torch.isfinite(ref).all())
    tr.dist.close()
    del tr

    tr = _trainer(g, 4242)
    batch = _batch(tr, scenes, peds)
    m = defaultdict(list)
    with launches_of() as seen:
        replay = tr.capture_iteration(batch, warmup=1)
    for _ in range(iters - 1):
        replay(m, False)
    tr.flush_metrics()
    assert torch.equal(_weights(tr), ref)
    assert torch.equal(torch.cat([tr.optimizerG.exp_avg, tr.optimizerD.exp_avg]).cpu(), ref_m)

    # --- the kernels of the measured configuration, by symbol (captured iteration = the second half of `seen`) ---
    b = scenes * peds
    syms = defaultdict(list)
    for entry, launches in seen:
        for sym, threads in launches:
            syms[sym].append(threads)
    names = set(syms)
    assert "decoder_fwd_wave_kernel" in names            # MGGAN_DEC_FWD_MIN: one wave per tile from 16,384 rollout rows
    assert "decoder_bwd_pair_kernel" in names            # two waves per tile from 4,096 rollout rows
    assert "sample_slots_scan_kernel" in names           # sampling + bucketing above 2,048 rollout rows
    assert "sample_bucket_small_kernel" in names or b > 2048  # the discriminator step's single-sample call
    conv1 = [k for k in names if k.startswith("conv1_pool_kernel<")]
    assert sorted(conv1) == ["conv1_pool_kernel<16>", "conv1_pool_kernel<8>"], conv1
    if 512 < b <= 2048:  # the balanced 512-workgroup grids of the scene CNN forward / conv1 kernels
        assert set(syms["conv1_pool_kernel<16>"]) == {512 * 256}, syms["conv1_pool_kernel<16>"]
    if b >= 4096:
        assert any(k.startswith("lstm_bwd_mfma_kernel<") for k in names)  # MGGAN_LSTM_MFMA_MIN_B
    assert not [k for k in names if k.startswith("mggan_") or k == "?"], names
    tr.dist.close()


def _record_draws(rng):
    """Wrap the device RNG's methods: the tensors they hand out are kept (inside a capture these are the graph's static
    buffers -- held here, so the capture's allocator cannot reuse them -- and hold a replay's draws after the replay)."""
    rec = {"labels": [], "noise": [], "idx": []}
    o_labels, o_noise, o_rows, o_gen = rng.labels, rng.noise, rng.sample_rows, rng.sample_generators

    def labels():
        out = o_labels()
        rec["labels"].append(out)
        return out

    def noise(*a, **k):
        out = o_noise(*a, **k)
        rec["noise"].append(out)
        return out

    def sample_rows(*a, **k):
        idx, rows = o_rows(*a, **k)
        rec["idx"].append(idx)
        return idx, rows

    def sample_generators(*a, **k):
        idx = o_gen(*a, **k)
        rec["idx"].append(idx)
        return idx

    rng.labels, rng.noise, rng.sample_rows, rng.sample_generators = labels, noise, sample_rows, sample_generators
    return rec


def _label(entry):
    u, lo, hi = entry
    return float(lo + (hi - lo) * float(u.float().cpu()))


@pytest.mark.parametrize("scenes,peds,g", SIZES)
def test_oracle_follows_a_replay_from_its_own_draws(scenes, peds, g):
    import mggan_oracle as O
    from helpers import rel_l2
    from mggan.data_utils import synthetic

    K = 20
    tr = _trainer(g, 977)
    sizes = synthetic.scene_sizes(scenes, peds)
    host_batch = synthetic.make_batch(sizes, seed=0)
    batch = tr.to_device(host_batch)
    batch["loss_mask"] = None
    m = defaultdict(list)
    tr.train_iteration(batch, m)  # eager: per-batch tables; the oracle starts from the state AFTER this iteration
    tr.flush_metrics()
    rec = _record_draws(tr.rng)
    replay = tr.capture_iteration(batch, warmup=0)
    # (the PM step's Categorical draw -- standard.py:184 of the reference, result unused -- is not made on the device path)
    assert len(rec["noise"]) == 3 and len(rec["idx"]) == 2 and len(rec["labels"]) == 3, {k: len(v) for k, v in rec.items()}
    torch.cuda.synchronize()

    # --- the oracle takes over the trainer's state: weights, BatchNorm statistics, Adam moments and step counts ---
    Go, Do = O.construct_oracle(g)
    Go.load_state_dict({k: v.detach().cpu() for k, v in tr.G.state_dict().items()})
    Do.load_state_dict({k: v.detach().cpu() for k, v in tr.D.state_dict().items()})
    Go.train()
    Do.train()
    tro = O.OracleTrainer(Go, Do, mode="block")
    for mod, ref_mod, opt, oopt in ((tr.G, Go, tr.optimizerG, tro.optG), (tr.D, Do, tr.optimizerD, tro.optD)):
        name_of = {id(p): n for n, p in mod.named_parameters()}
        ref_params = dict(ref_mod.named_parameters())
        steps = opt.seg_step.cpu()
        for i, (p, o) in enumerate(mod._flat_items):
            if int(steps[i]) == 0:
                continue
            q = ref_params[name_of[id(p)]]
            n = p.numel()
            oopt.state[q] = {"step": int(steps[i]), "m": opt.exp_avg[o:o + n].view(p.shape).cpu().clone(),
                             "v": opt.exp_avg_sq[o:o + n].view(p.shape).cpu().clone()}

    m_gpu = defaultdict(list)
    replay(m_gpu, True)
    torch.cuda.synchronize()

    # --- what the replay drew, read from the graph's static buffers ---
    noise = [n.float().cpu() for n in rec["noise"]]
    idx = [i.cpu() for i in rec["idx"]]
    b = sum(sizes)
    assert [tuple(n.shape) for n in noise] == [(1, b, 8), (K, b, 8), (1, b, 8)]
    assert [tuple(i.shape) for i in idx] == [(b, 1), (b, K)] and all(0 <= int(i.min()) and int(i.max()) < g for i in idx)
    for n in noise:  # one vector per (sample, scene), repeated for the scene's pedestrians (utils.py:160-165)
        s0 = 0
        for sz in sizes[:4]:
            assert torch.equal(n[:, s0:s0 + 1].expand(-1, sz, -1), n[:, s0:s0 + sz])
            s0 += sz
    assert float(noise[1].std()) > 0.8 and float(noise[1].mean().abs()) < 0.1
    (d_real, _), (_, d_fake), (g_real, g_fake) = rec["labels"]
    lab_d1, lab_d2, lab_g = (_label(d_real), 0.0), (0.0, _label(d_fake)), (_label(g_real), _label(g_fake))
    assert 0.9 <= lab_d1[0] <= 1.0 and 0.0 <= lab_d2[1] <= 0.1 and 0.9 <= lab_g[0] <= 1.0

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m_cpu = defaultdict(list)
    mask = torch.ones(b, dtype=torch.bool)
    args = tuple(host_batch[k] for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy")) + (host_batch["seq_start_end"],)
    for step, n, i, labs in (("discriminator_step", noise[0], idx[0], (lab_d1, lab_d2)),
                             ("generator_step", noise[1], idx[1], (lab_g, lab_g)),
                             ("net_chooser_step", noise[2], None, (lab_g, lab_g))):
        draws = {"noise": n, "labels": labs[0], "labels1": labs[0], "labels2": labs[1]}
        if i is not None:
            draws["gen_idxs"] = i
        getattr(tro, step)(*args, m_cpu, mask, host_batch["features"], draws=draws)

    assert set(m_cpu) <= set(m_gpu), (sorted(m_cpu), sorted(m_gpu))
    for key, v in m_cpu.items():
        assert abs(m_gpu[key][-1] - v[-1]) <= 1e-3 * abs(v[-1]) + 1e-6, (key, m_gpu[key][-1], v[-1])
    for mod, ref_mod in ((tr.G, Go), (tr.D, Do)):
        a = torch.cat([p.detach().cpu().flatten() for p in mod.parameters()]).double()
        r = torch.cat([p.detach().flatten() for p in ref_mod.parameters()]).double()
        assert rel_l2(a.numpy(), r.numpy()) <= 1e-3, rel_l2(a.numpy(), r.numpy())
        # and the step was a step: the parameters moved by more than the tolerance is wide
    tr.dist.close()


def test_every_entry_of_an_iteration_maps_to_a_kernel_symbol():
    """bench.py books an entry's time on the symbol the library's launch log reports for that very call: no entry of the
    iteration may come back without one, and none may be spelled like a C-ABI entry."""
    import bench
    from mggan.hip.lib import start_trace, stop_trace

    tr = _trainer(4, 5)
    batch = _batch(tr, 64, 20)
    m = defaultdict(list)
    tr.train_iteration(batch, m)
    start_trace()
    tr.train_iteration(batch, m)
    trace = stop_trace()
    tr.flush_metrics()
    assert len(trace) > 30
    seen = set()
    for name, (calls, ms, arglist, launchlist) in trace.items():
        assert len(launchlist) == calls
        for launches in launchlist:
            sym = bench.kernel_of(name, launches)
            assert launches, name  # every traced entry of the iteration launches at least one kernel
            assert not sym.startswith("mggan_") and "?" not in sym and not sym.startswith("_Z"), (name, sym)
            seen.add(sym)
    for want in ("decoder_bwd_pair_kernel", "attn_kernel<16,true>", "attn_kernel<8,true>", "attn_kernel<16,false>",
                 "lstm_fwd_mfma_kernel<64>", "lstm_fwd_mfma_kernel<32>", "grad_reduce_multi_kernel", "mlp_chain_kernel"):
        assert want in seen, (want, sorted(seen))
    tr.dist.close()
