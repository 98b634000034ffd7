"""GPU: device-side row bucketing vs the host bookkeeping, and HIP-graph replay of a whole iteration."""
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bucket_rows_matches_host_tables():
    from mggan.hip.functions import RolloutRows, device_rollout_rows
    from mggan.utils import get_selection_indices

    for b, K, g, seed in ((7, 20, 4, 0), (1280, 20, 4, 1), (33, 5, 1, 2), (513, 20, 8, 3), (1280, 1, 4, 4),
                          (2000, 2, 8, 5)):
        idx = torch.randint(0, g, (b, K), generator=torch.Generator().manual_seed(seed))
        off = get_selection_indices(idx)
        host = RolloutRows(idx.t().reshape(-1).numpy(), np.tile(np.arange(b), K), off.t().reshape(-1).numpy(), g, b, "cpu")
        dev = device_rollout_rows(idx.cuda(), g)
        for name in ("row_gen", "row_ped", "row_slot", "row_pos", "inv", "seg", "row_gen_pos"):
            assert torch.equal(getattr(dev, name).cpu(), getattr(host, name)), (name, b, K, g)


def test_graph_replay_trains_like_eager():
    """Same seed: N eager iterations == N graph replays (device RNG consumes the same philox stream)."""
    import bench
    from mggan.data_utils import synthetic

    dev = torch.device("cuda", 0)
    sizes = synthetic.scene_sizes(6, None, seed=4)
    losses = {}
    for mode in ("eager", "graph"):
        tr = bench.build_trainer(2, "device", dev)
        batch = tr.to_device(synthetic.make_batch(sizes, seed=2))
        batch["loss_mask"] = None
        m = defaultdict(list)
        if mode == "eager":
            for _ in range(6):
                tr.train_iteration(batch, m)
        else:
            replay = tr.capture_iteration(batch, warmup=2)  # 2 warm-up + 1 capture iteration are real steps
            for _ in range(3):
                replay(m)
        losses[mode] = m
        assert all(np.isfinite(v).all() for v in m.values())
        assert len(m["train/discr_loss"]) in (3, 6)
    # the graph run logged only its 3 replays; they must look like training (finite, D loss near 2*ln2 early on)
    assert 0.2 < losses["graph"]["train/discr_loss"][-1] < 3.0
    assert abs(losses["graph"]["train/discr_loss"][-1] - losses["eager"]["train/discr_loss"][-1]) < 0.5


def test_device_categorical_sampler_statistics():
    """Inverse-CDF sampler: empirical frequencies follow softmax(logits); indices in range."""
    from mggan.rng import DeviceRNG

    rng = DeviceRNG(seed=3)
    logits = torch.tensor([[0.0, 1.0, -1.0, 0.5]] * 4096, device="cuda")
    idx = rng.sample_generators(logits, 20)
    assert idx.shape == (4096, 20) and idx.dtype == torch.int64
    assert int(idx.min()) >= 0 and int(idx.max()) <= 3
    freq = torch.bincount(idx.flatten(), minlength=4).float() / idx.numel()
    ref = torch.softmax(logits[0], 0)
    assert float((freq.cpu() - ref.cpu()).abs().max()) < 0.01


@pytest.mark.parametrize("b,K,g", [(1280, 20, 4), (8192, 20, 8), (37, 1, 16), (5, 3, 1)])
def test_device_categorical_sampler_is_the_inverse_cdf(b, K, g):
    """mggan_sample_categorical (one lane per (pedestrian, sample)) against the inverse CDF evaluated in float64 on the
    host: the same pick for every uniform that is not within rounding distance of a CDF step."""
    import ctypes

    import numpy as np

    from mggan.hip import lib

    gen = torch.Generator().manual_seed(b + K + g)
    logits = torch.randn(b, g, generator=gen) * 2
    u = torch.rand(b, K, generator=gen)
    idx = torch.empty(b, K, dtype=torch.int64, device="cuda")
    ld, ud = logits.cuda(), u.cuda()
    lib.mggan_sample_categorical(b, K, g, ld.data_ptr(), ud.data_ptr(), idx.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    e = np.exp(logits.double().numpy() - logits.double().numpy().max(1, keepdims=True))
    cdf = np.cumsum(e, 1)
    x = u.double().numpy() * cdf[:, -1:]
    ref = np.minimum((x[:, :, None] >= cdf[:, None, :]).sum(2), g - 1)
    margin = np.abs(x[:, :, None] - cdf[:, None, :]).min(2) / cdf[:, -1:]
    got = idx.cpu().numpy()
    clear = margin > 1e-5
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(got[clear], ref[clear])
    assert got.min() >= 0 and got.max() <= g - 1


@pytest.mark.parametrize("b,K,g", [(1280, 1, 4), (256, 20, 4), (1280, 20, 4), (8192, 20, 8), (37, 3, 16), (409, 20, 5),
                                   (8192, 1, 8), (5, 3, 1)])
def test_fused_sampling_and_bucketing_match_the_separate_launches(b, K, g):
    """mggan_sample_bucket_rows (one launch up to 2,048 rows; above: picks, slots, counts and the scan in one launch with a
    last-ticket scan, then the scatter) against mggan_sample_categorical followed by mggan_bucket_rows on the same logits and
    uniforms: identical picks and identical row tables, twice in a row (the ticket re-arms itself)."""
    from mggan.hip import lib
    from mggan.hip.functions import device_rollout_rows, empty_rollout_rows

    gen = torch.Generator().manual_seed(3 * b + K + g)
    ld = (torch.randn(b, g, generator=gen) * 2).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    keep = torch.zeros(16 * ((b * K + 1023) // 1024) + 16, dtype=torch.int32, device="cuda")  # the self-resetting counters
    for rep in range(3):
        ud = torch.rand(b, K, generator=gen).cuda()
        idx = torch.empty(b, K, dtype=torch.int64, device="cuda")
        lib.mggan_sample_categorical(b, K, g, ld.data_ptr(), ud.data_ptr(), idx.data_ptr(), st)
        ref = device_rollout_rows(idx, g)
        idx2 = torch.full((b, K), -1, dtype=torch.int64, device="cuda")
        rows, blk = empty_rollout_rows(b, K, g, ld.device)
        # rep 0: a fresh buffer behind a memset node; rep 1, 2: the caller-owned buffer every call leaves zeroed
        own = rep > 0
        lib.mggan_sample_bucket_rows(b, K, g, ld.data_ptr(), ud.data_ptr(), idx2.data_ptr(), rows.row_gen.data_ptr(),
                                     rows.row_ped.data_ptr(), rows.row_slot.data_ptr(), rows.row_pos.data_ptr(),
                                     rows.inv.data_ptr(), rows.seg.data_ptr(), rows.row_gen_pos.data_ptr(),
                                     (keep if own else blk).data_ptr(), 1 if own else 0, ticket.data_ptr(), st)
        if own:
            assert int(keep.abs().sum()) == 0
        assert torch.equal(idx, idx2), (b, K, g, rep)
        for name in ("row_gen", "row_ped", "row_slot", "row_pos", "inv", "seg", "row_gen_pos"):
            assert torch.equal(getattr(rows, name), getattr(ref, name)), (name, b, K, g, rep)
        assert int(ticket) == 0


def test_branch_streams_and_graph_replay_are_bit_identical():
    """No races, no order-dependent arithmetic: after several iterations the weights are bit-identical whether the
    step graph runs on one stream, on the branch streams, or as a replayed HIP graph (same seeds)."""
    import bench
    from mggan.data_utils import synthetic
    from mggan.hip import functions as HF

    dev = torch.device("cuda", 0)

    def run(branches, graph, iters=4):
        HF.enable_branches(branches)
        try:
            tr = bench.build_trainer(3, "device", dev)
            torch.cuda.manual_seed(777)
            batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(24, 6), seed=0))
            batch["loss_mask"] = None
            tr.defer_metrics = True
            tr.zero_grads_in_step = True
            m = defaultdict(list)
            if graph:
                replay = tr.capture_iteration(batch, warmup=1)
                for _ in range(iters - 1):
                    replay(m, False)
            else:
                for _ in range(iters):
                    tr.train_iteration(batch, m)
            tr.flush_metrics()
            torch.cuda.synchronize()
            return torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu()
        finally:
            HF.enable_branches(True)

    ref = run(False, False)
    assert bool(torch.isfinite(ref).all())
    for branches, graph in ((True, False), (True, True)):
        assert torch.equal(run(branches, graph), ref), (branches, graph)


def test_generator_forward_beside_the_discriminator_backward_is_bit_identical(monkeypatch):
    """From 2,048 pedestrians on the trainer issues the generator step's forward pass on a held branch stream before the
    discriminator step's backward pass (mggan/model/train.py: _early_generator_forward).  Same arithmetic, same random draws:
    with the threshold forced to 0 the weights after several iterations equal those of the in-step order bit for bit, eagerly
    and as a replayed graph -- and the early path is the one that ran."""
    import bench
    from mggan.data_utils import synthetic
    from mggan.model import train as T

    dev = torch.device("cuda", 0)

    from mggan.hip import functions as HF

    monkeypatch.setattr(HF, "_BRANCH_EAGER", True)  # (eager iterations run on one stream by default: here they fork too)

    def run(min_b, graph, iters=4):
        monkeypatch.setattr(T, "_G_EARLY_MIN_B", min_b)
        calls = []
        real = T.PiNetMultiGeneratorGAN._early_generator_forward
        monkeypatch.setattr(T.PiNetMultiGeneratorGAN, "_early_generator_forward",
                            lambda self, *a: (calls.append(1), real(self, *a))[1])
        tr = bench.build_trainer(3, "device", dev)
        torch.cuda.manual_seed(777)
        batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(24, 6), seed=0))
        batch["loss_mask"] = None
        tr.defer_metrics = True
        tr.zero_grads_in_step = True
        m = defaultdict(list)
        if graph:
            replay = tr.capture_iteration(batch, warmup=1)
            for _ in range(iters - 1):
                replay(m, False)
        else:
            for _ in range(iters):
                tr.train_iteration(batch, m)
        tr.flush_metrics()
        torch.cuda.synchronize()
        monkeypatch.setattr(T.PiNetMultiGeneratorGAN, "_early_generator_forward", real)
        return torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu(), len(calls)

    ref, n = run(1 << 30, False)
    assert n == 0 and bool(torch.isfinite(ref).all())
    for graph in (False, True):
        got, n = run(0, graph)
        assert n >= 2, "the early path did not run"
        assert torch.equal(got, ref), graph


def test_folded_weights_are_cached_per_weight_version():
    """Inside the trainer's iteration the folded LSTM weights are launched once per weight version: four mggan_lstm_fold
    calls per steady-state iteration instead of seven, and the weights after eager iterations, a capture, replays and
    more eager iterations are bit-identical to a run that folds in every forward (MGGAN_PREP_CACHE=0 semantics)."""
    import bench
    from mggan.data_utils import synthetic
    from mggan.hip import functions as HF
    from mggan.hip.lib import start_trace, stop_trace

    dev = torch.device("cuda", 0)

    def run(cache):
        keep = HF._PREP["enabled"]
        HF._PREP["enabled"] = cache
        HF.clear_prep_cache()
        try:
            tr = bench.build_trainer(3, "device", dev)
            torch.cuda.manual_seed(99)
            batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(8, 5), seed=1))
            batch["loss_mask"] = None
            tr.defer_metrics = True
            m = defaultdict(list)
            tr.train_iteration(batch, m)  # (the first iteration folds everything once more: nothing is cached yet)
            start_trace()
            tr.train_iteration(batch, m)
            folds = stop_trace().get("mggan_lstm_fold", (0, [], []))[0]
            replay = tr.capture_iteration(batch, warmup=0)
            for _ in range(2):
                replay(m, False)
            tr.train_iteration(batch, m)  # eager again, right after a replay: every cached fold is stale
            with torch.no_grad():  # a weight written behind the optimizer's back, through torch
                tr.G.encoder.embedding.weight.mul_(1.5)
            tr.train_iteration(batch, m)
            tr.flush_metrics()
            torch.cuda.synchronize()
            out = folds, torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu()
            del replay
            return out
        finally:
            HF._PREP["enabled"] = keep
            HF.clear_prep_cache()
            import gc

            gc.collect()

    f_on, w_on = run(True)
    f_off, w_off = run(False)
    assert (f_on, f_off) == (4, 7), (f_on, f_off)
    assert bool(torch.isfinite(w_on).all()) and torch.equal(w_on, w_off)


def test_replay_after_an_external_weight_write_refolds_the_lstm_weights():
    """A steady-state graph folds a module's LSTM weights only behind the optimizer step that changed them and reads, at its
    start, what the previous iteration left in the folded buffers.  Weights written between two replays
    behind the trainer's back -- load_state_dict here -- must reach the next replay: the trainer sees the write in its
    host-side fingerprint and re-folds first.  Reference run: the same sequence with eager launches."""
    import bench
    from mggan.data_utils import synthetic

    dev = torch.device("cuda", 0)

    def run(graph):
        tr = bench.build_trainer(3, "device", dev)
        torch.cuda.manual_seed(4321)
        batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(10, 5), seed=3))
        batch["loss_mask"] = None
        tr.defer_metrics = True
        tr.zero_grads_in_step = True
        m = defaultdict(list)
        g0 = {k: v.detach().clone() for k, v in tr.G.state_dict().items()}
        d0 = {k: v.detach().clone() for k, v in tr.D.state_dict().items()}
        if graph:
            replay = tr.capture_iteration(batch, warmup=1)
            step = lambda: replay(m, False)
        else:
            tr.train_iteration(batch, m)
            step = lambda: tr.train_iteration(batch, m)
        step()
        step()
        # back to the initial weights (BatchNorm statistics included), Adam moments kept: only a re-fold makes the encoders /
        # decoders of the next replay see them
        tr.G.load_state_dict(g0)
        tr.D.load_state_dict(d0)
        step()
        step()
        tr.flush_metrics()
        torch.cuda.synchronize()
        return torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu()

    ref = run(False)
    got = run(True)
    assert bool(torch.isfinite(ref).all())
    assert torch.equal(got, ref)


def test_next_discriminator_context_ahead_of_time_is_bit_identical(monkeypatch):
    """Cross-iteration pipelining (mggan/model/train.py: _issue_d_context, off by default -- measured slower, kept as a knob):
    the next iteration's discriminator context is issued beside the PM-network step into fixed buffers (HF.ReplayAlloc), the
    next discriminator step takes it.  Same arithmetic in the same order: weights AND D's BatchNorm running statistics after
    five iterations (the last context rolled back, drain_pipeline) equal the in-order schedule's bit for bit -- eagerly and
    with a captured graph that reads the context the previous replay left; the taken path is the one that ran."""
    import bench
    from mggan.data_utils import synthetic
    from mggan.model import train as T

    dev = torch.device("cuda", 0)

    from mggan.hip import functions as HF

    monkeypatch.setattr(HF, "_BRANCH_EAGER", True)  # (the eager variant needs the branch streams an eager iteration no longer uses)

    def run(pipeline, graph, where="pm_begin", iters=5):
        monkeypatch.setattr(T, "_PIPE_AT", where)
        taken = []
        real = T.PiNetMultiGeneratorGAN._take_d_context

        def spy(self, *a):
            out = real(self, *a)
            taken.append(out is not None)
            return out

        monkeypatch.setattr(T.PiNetMultiGeneratorGAN, "_take_d_context", spy)
        tr = bench.build_trainer(3, "device", dev)
        torch.cuda.manual_seed(4321)
        batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(20, 5), seed=2))
        batch["loss_mask"] = None
        tr.defer_metrics = True
        tr._pipe["on"] = pipeline
        m = defaultdict(list)
        if graph:
            replay = tr.capture_iteration(batch, warmup=2, pipeline=pipeline)
            for _ in range(iters - 2):
                replay(m, False)
        else:
            b = dict(batch, next={"in_dxdy": batch["in_dxdy"], "features": batch["features"]}) if pipeline else batch
            for _ in range(iters):
                tr.train_iteration(b, m)
        tr.drain_pipeline()
        tr.flush_metrics()
        torch.cuda.synchronize()
        monkeypatch.setattr(T.PiNetMultiGeneratorGAN, "_take_d_context", real)
        bn = torch.cat([t.detach().double().flatten() for t in tr._d_bn_buffers()]).cpu()
        return torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu(), bn, sum(taken)

    ref, bn_ref, n = run(False, False)
    assert n == 0 and bool(torch.isfinite(ref).all())
    for graph, where in ((False, "pm_begin"), (True, "pm_begin"), (True, "pm_tail")):
        got, bn, n = run(True, graph, where)
        assert n >= 1, "no issued context was taken"
        assert torch.equal(got, ref) and torch.equal(bn, bn_ref), (graph, where)


def test_decoder_prefold_on_a_branch_is_bit_identical(monkeypatch):
    """HF.prefold (MGGAN_PM_PREFOLD, off by default -- measured slower): the decoders' folded weights of the PM-network step
    produced ahead of time on a branch stream, handed to the rollout through an event.  Same weights after four iterations."""
    import bench
    from mggan.data_utils import synthetic
    from mggan.hip import functions as HF
    from mggan.model import train as T

    dev = torch.device("cuda", 0)

    def run(prefold):
        monkeypatch.setattr(T, "_PM_PREFOLD", prefold)
        n = []
        real = HF.prefold
        monkeypatch.setattr(HF, "prefold", lambda owner: (n.append(real(owner)), n[-1])[1])
        tr = bench.build_trainer(3, "device", dev)
        torch.cuda.manual_seed(99)
        batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(12, 5), seed=1))
        batch["loss_mask"] = None
        tr.defer_metrics = True
        replay = tr.capture_iteration(batch, warmup=2)
        for _ in range(2):
            replay(defaultdict(list), False)
        torch.cuda.synchronize()
        monkeypatch.setattr(HF, "prefold", real)
        return torch.cat([tr.G._flat.clone(), tr.D._flat.clone()]).cpu(), sum(bool(x) for x in n)

    ref, n0 = run(False)
    got, n1 = run(True)
    assert n0 == 0 and n1 >= 2, (n0, n1)  # (the fold really ran ahead of time)
    assert torch.equal(got, ref)
