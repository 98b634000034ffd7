"""GPU: the data-parallel HIP trainer (scene sharding + flat-gradient all-reduce + BatchNorm-statistics
all-reduce + global generator counts) equals the single-process trainer.  Two ranks share the one GPU of
the test box, so the process group uses gloo (RCCL refuses two ranks on one device) -- it only carries the start-up
handshake: the collectives of the iteration are the peer-mapped all-reduce kernels of csrc/comm.hip (the two processes
map each other's arenas through hipIpc), the code path bench.py --gpus N runs on a multi-GPU node."""
import os
import socket
from collections import defaultdict

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_COLLECTIVES = 13  # exchanges per sharded iteration with global-batch BatchNorm (round 4: 18)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _draws(sizes, g, K, seed):
    gen = torch.Generator().manual_seed(seed)
    b, S = sum(sizes), len(sizes)
    rep = torch.tensor(sizes)
    out = {"labels": [(0.95, 0.05), (0.93, 0.07), (0.97, 0.02)]}
    out["noise"] = [torch.randn(k, S, 8, generator=gen).repeat_interleave(rep, dim=1) for k in (1, K, 1)]
    out["idx"] = [torch.randint(0, g, (b, k), generator=gen) for k in (1, K, 1)]
    return out


def _run(rank, world, port, sizes, q, nan_peds=(), iters=2):
    import sys

    for p in (os.path.join(ROOT, "mg-gan_amd"), ROOT):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import bench
    from mggan.data_utils import synthetic
    from mggan.parallel import shard_batch, shard_scenes
    from mggan.rng import ReplayRNG

    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dev = torch.device("cuda", 0)
    g, K = 2, 20
    tr = bench.build_trainer(g, "host", dev)
    full = synthetic.make_batch(sizes, seed=9)
    dr = _draws(sizes, g, K, 5)
    _, p0, p1, _ = shard_scenes(full["seq_start_end"], rank, world)
    valid = torch.ones(sum(sizes), dtype=torch.bool)
    if len(nan_peds):  # pedestrians without ground truth: train_iteration computes the loss mask from the NaNs
        valid[list(nan_peds)] = False
        full["gt_xy"][:, ~valid] = float("nan")
        full["gt_dxdy"][:, ~valid] = float("nan")
    batch = tr.to_device(shard_batch(full, rank, world))
    if not len(nan_peds):
        batch["loss_mask"] = None
    m = defaultdict(list)
    for it in range(iters):
        tr.rng = tr.G.rng = ReplayRNG(labels=list(dr["labels"]), noise=[n[:, p0:p1] for n in dr["noise"]],
                                      gen_idxs=[i[p0:p1][valid[p0:p1]] for i in dr["idx"]])
        tr.train_iteration(batch, m)
    flat = torch.cat([tr.G._flat.cpu(), tr.D._flat.cpu()])
    bn = float(tr.G.scene_encoder.CNN.encoder.ConvBlock_1.Block.BN_1.running_var.sum().cpu())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    q.put((rank, flat.numpy(), bn, {k: v for k, v in m.items() if "probs" not in k}, tr.G._flat.numel()))


def _launch(world, sizes, nan_peds=(), iters=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, sizes, q, nan_peds, iters)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_ranks_match_single_process():
    sizes = [3, 2, 4, 1, 3, 5]  # ranks get 9 pedestrians each (3 scenes each)
    single = _launch(1, sizes)[0]
    two = _launch(2, sizes)
    for r in two:
        rel = np.linalg.norm(r[1] - single[1]) / np.linalg.norm(single[1])
        assert rel <= 1e-3, rel                    # replicas stay in lock-step with the single-process run
        assert abs(r[2] - single[2]) <= 1e-3 * abs(single[2])  # BatchNorm running stats from GLOBAL statistics
        for k, v in single[3].items():              # logged losses are global means on every rank
            np.testing.assert_allclose(r[3][k], v, rtol=2e-3, atol=1e-6, err_msg=k)
    assert np.array_equal(two[0][1], two[1][1])    # bit-identical replicas


def test_two_ranks_match_single_process_on_a_masked_batch():
    """Pedestrians with NaN ground truth: the masked discriminator step runs D's scene encoder once for the real and once
    for the fake pass (no shared context), so a root carries TWO conv1 gradient tails per optimizer step -- the first rides
    with the gradient all-reduce, the second is an exchange of its own (mggan/parallel.py: all_reduce_grads) -- and a rank
    WITHOUT a NaN of its own takes the masked path too (abstract_train.py: the `mask.any` exchange)."""
    sizes = [3, 2, 4, 1, 3, 5]
    # (a) the NaN sits on the last pedestrian of the batch (rank 1 only): two iterations, everything equal
    single = _launch(1, sizes, (17,))[0]
    two = _launch(2, sizes, (17,))
    for r in two:
        rel = np.linalg.norm(r[1] - single[1]) / np.linalg.norm(single[1])
        assert rel <= 1e-3, rel
        assert abs(r[2] - single[2]) <= 1e-3 * abs(single[2])
        for k, v in single[3].items():
            np.testing.assert_allclose(r[3][k], v, rtol=2e-3, atol=1e-6, err_msg=k)
    assert np.array_equal(two[0][1], two[1][1])
    # (b) NaNs in the middle of the batch.  The reference slices the MASKED predictions of the generator step with the
    # UNMASKED scene bounds (train.py:67-68): every scene behind a masked pedestrian reads shifted rows -- a
    # property of the global pedestrian numbering that a shard cannot reproduce.  The discriminator step has no such
    # quirk: after ONE iteration D's weights equal the single-process run's, and the replicas are bit-identical.
    nan_peds = (1, 7, 12)
    single = _launch(1, sizes, nan_peds, iters=1)[0]
    two = _launch(2, sizes, nan_peds, iters=1)
    nG = single[4]
    for r in two:
        rel = np.linalg.norm(r[1][nG:] - single[1][nG:]) / np.linalg.norm(single[1][nG:])
        assert rel <= 1e-3, rel
        assert np.isfinite(r[1]).all()
        np.testing.assert_allclose(r[3]["train/discr_loss"], single[3]["train/discr_loss"], rtol=2e-3)
    assert np.array_equal(two[0][1], two[1][1])


def _run_graph(rank, world, port, sizes, q, device_comm, backend="gloo", own_device=False, rccl_graph=True, eager=False,
               env=None):
    """Sharded iteration captured on `world` ranks (sharing the box's one GPU, handles exchanged over gloo -- or, with
    own_device, one GPU per rank and any backend: tests/test_hip_multigpu.py):
    device_comm=True: peer-mapped all-reduce kernels inside ONE graph; False: graph segments around eager collectives."""
    import sys

    for p in (os.path.join(ROOT, "mg-gan_amd"), ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      MGGAN_DEVICE_COMM="1" if device_comm else "0", MGGAN_RCCL_GRAPH="1" if rccl_graph else "0")
    if world == 1:
        os.environ["MGGAN_FORCE_DIST"] = "1"
    os.environ.update(env or {})
    import torch.distributed as dist

    import bench
    from mggan.data_utils import synthetic
    from mggan.parallel import replicas_in_sync, shard_batch

    dev = torch.device("cuda", rank if own_device else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = bench.build_trainer(2, "device", dev, seed=rank)
    assert tr.dist.enabled
    tr.dist.equal_shards = True
    full = synthetic.make_batch(sizes, seed=9)
    batch = tr.to_device(shard_batch(full, rank, world))
    batch["loss_mask"] = None
    tr.defer_metrics = True
    m = defaultdict(list)
    if eager:  # the same five iterations as eager launches (same Philox state: the device RNG counts iterations)
        for i in range(5):
            tr.train_iteration(batch, m)
        tr.flush_metrics()
        torch.cuda.synchronize()
        flat = torch.cat([tr.G._flat.cpu(), tr.D._flat.cpu()])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, flat.numpy(), 0, 5, {}, tr.dist.devcomm is not None, True, "eager", [], tr.dist.transport))
        return
    replay = tr.capture_iteration(batch, warmup=2)
    for i in range(3):
        replay(m, True)
    torch.cuda.synchronize()
    tr.dist.check(sync=True)
    sync = replicas_in_sync(tr.G, tr.D)
    flat = torch.cat([tr.G._flat.cpu(), tr.D._flat.cpu()])
    steps = int(tr.optimizerD.seg_step.max().cpu())
    one_graph = isinstance(replay.graph, torch.cuda.CUDAGraph)
    # the exchange schedule of one iteration, counted on an eager iteration behind the replays
    tr.dist.reset_count(log=True)
    tr.train_iteration(batch, defaultdict(list))
    torch.cuda.synchronize()
    schedule = list(tr.dist.collective_log)
    # ... and the generator counts that rode with the discriminator step's gradients are the counts of the picks
    os.environ["MGGAN_CHECK_RIDERS"] = "1"
    tr.train_iteration(batch, defaultdict(list))
    torch.cuda.synchronize()
    os.environ["MGGAN_CHECK_RIDERS"] = "0"
    assert "count (check)" in tr.dist.collective_log, tr.dist.collective_log
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, flat.numpy(), 1 if one_graph else replay.graph.n_graphs, steps,
           {k: v for k, v in m.items() if "probs" not in k}, tr.dist.devcomm is not None, sync, tr.launch_mode, schedule,
           tr.dist.transport))


def _launch_graph(world, sizes, device_comm, backend="gloo", own_device=False, rccl_graph=True, eager=False, env=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_graph, args=(r, world, port, sizes, q, device_comm, backend, own_device, rccl_graph, eager,
                                                  env)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=200 + 40 * world) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def _check_replicas(res, replays=3):
    (_, f0, n0, s0, m0, peer_mapped, sync0, _, sched0, transport0) = res[0]
    if transport0 == "rccl-segments":  # torch.distributed between graph segments: the f64 tail of a gradient exchange is a second call
        assert sum("second call" in w for w in sched0) == 3, sched0
        sched0 = [w for w in sched0 if "second call" not in w]
    # <= 13 exchanges per iteration (round 4: 18): ONE Gram all-reduce serves the layer-1 forward statistics of the passes
    # issued behind it (default schedule `late`: two passes come before it and exchange their own sums; MGGAN_GRAM_SCHEDULE
    # =first: none, 11 exchanges, the iteration waits for the matrix); the three layer-1 adjoints and the generator counts
    # ride with the three gradient all-reduces; layer 2 keeps its 4 + 3 synchronisation points (the floor with
    # global-batch BatchNorm is 10: DESIGN section 6)
    assert len(sched0) <= MAX_COLLECTIVES, sched0
    assert sum("gradients" in w for w in sched0) == 3 and sum(w.startswith("bn1") for w in sched0) <= 2, sched0
    assert "count" not in sched0 and sched0.count("gram") == 1, sched0
    assert not [w for w in sched0 if w == "bn1.backward"], sched0
    assert np.isfinite(f0).all() and 0.2 < m0["train/discr_loss"][-1] < 3.0
    for (_, f1, n1, s1, m1, _, sync1, _, sched1, _) in res[1:]:
        assert [w for w in sched1 if "second call" not in w] == sched0  # the same exchanges in the same order on every rank
        assert n0 == n1 and sync0 and sync1
        assert s0 == s1 == 2 + replays         # 2 eager warm-up iterations + the replays (capturing executes nothing)
        assert np.array_equal(f0, f1)          # replicas stay bit-identical through the replays
        for k, v in m0.items():
            assert len(v) == replays and np.isfinite(v).all(), k
            np.testing.assert_allclose(v, m1[k], rtol=1e-6)      # logged losses are global means on every rank


def test_two_ranks_graph_segments():
    """Fallback without peer-mapped memory: every torch.distributed collective cuts the capture."""
    res = _launch_graph(2, [3, 3, 3, 3], device_comm=False)  # equal shards: 6 pedestrians per rank
    _check_replicas(res)
    assert res[0][2] > 10 and not res[0][5]  # the collectives cut the iteration into graph segments


def test_two_ranks_one_graph_with_device_allreduce():
    """Default on one node: the collectives are peer-mapped all-reduce kernels INSIDE the iteration graph (two
    processes map each other's arenas through hipIpc; both replay ONE graph with the branch streams on)."""
    res = _launch_graph(2, [3, 3, 3, 3], device_comm=True)
    _check_replicas(res)
    assert res[0][5] and res[0][2] == 1 and "peer-mapped" in res[0][7]
    seg = _launch_graph(2, [3, 3, 3, 3], device_comm=False)
    # same arithmetic as the segmented replay with eager gloo collectives (other reduction order across ranks: 1e-5)
    rel = np.linalg.norm(res[0][1] - seg[0][1]) / np.linalg.norm(seg[0][1])
    assert rel <= 1e-4, rel


def test_large_shards_put_the_gram_matrix_first_eleven_exchanges():
    """From MGGAN_GRAM_FIRST_MIN_B images per rank on (4,096; forced down to 1 here) the iteration starts with the Gram matrix
    of the image patches and its ONE all-reduce: every conv1 pass takes its BatchNorm statistics from it, no pass exchanges
    layer-1 sums of its own -- 11 exchanges: Gram + 4 layer-2 forward + 3 layer-2 backward + 3 gradient buffers (each with
    its conv1 tail and, in the discriminator step's, the generator counts).  The decision is taken from the shard's image
    count, which equal shards share: every rank decides alike (why the two discriminator-step layer-2
    forward exchanges are not merged into one)."""
    res = _launch_graph(2, [3, 3, 3, 3], device_comm=True, env={"MGGAN_GRAM_FIRST_MIN_B": "1"})
    _check_replicas(res)
    sched = res[0][8]
    assert sched[0] == "gram" and len(sched) == 11, sched
    assert sorted(sched) == sorted(["gram"] + ["bn2.forward"] * 4 + ["bn2.backward"] * 3 + ["gradients+conv1.tail"] * 3), sched
    # every step ends with its gradient exchange, its layer-2 adjoint exchange right before it
    ends = [i for i, w in enumerate(sched) if w.startswith("gradients")]
    assert ends[-1] == len(sched) - 1 and all(sched[i - 1] == "bn2.backward" for i in ends), sched


def test_eight_ranks_one_graph_like_the_eight_gpu_configs():
    """BASELINE configs[3] / configs[4] shard the batch over EIGHT ranks.  No 8-GPU node here: eight processes share the one
    GPU, map each other's arenas (8 slots, 8 flags per chunk, rank-ordered sums over 8 terms) and replay ONE graph each --
    replicas bit-identical after two eager iterations and three replays, finite losses that are the same global means on every
    rank, the 13-exchange schedule in the same order on every rank.  (Not compared with another world size: the device RNG
    mixes the rank into its Philox key, so 8 ranks draw other noise than 2.)"""
    sizes = [3] * 16  # two scenes of three pedestrians per rank
    res = _launch_graph(8, sizes, device_comm=True)
    _check_replicas(res)
    assert all(r[5] and r[2] == 1 and "peer-mapped" in r[7] for r in res), [(r[0], r[5], r[2]) for r in res]


def test_one_rank_forced_collectives_stay_in_one_graph():
    """MGGAN_FORCE_DIST=1: the collective hooks run with one rank -- what the sharded launch mode costs can be measured
    on a one-GPU box (bench.py); here: it is one graph and trains."""
    (r,) = _launch_graph(1, [3, 3, 3, 3], device_comm=True)
    assert r[5] and r[2] == 1 and r[6] and np.isfinite(r[1]).all()


def test_one_rank_rccl_inside_the_one_graph_is_bit_identical_to_eager_launches():
    """The north-star transport as a launch of this library (csrc/rccl.hip: ncclAllReduce bound from librccl.so, issued on
    the capturing stream): with the peer-mapped kernels switched off the sharded iteration is still ONE graph -- no
    segments --, and 2 eager + 3 replayed iterations leave the weights 5 eager iterations leave, to the bit.  (One rank:
    RCCL wants a GPU per rank, the box has one; tests/test_hip_multigpu.py repeats it across devices.)"""
    sizes = [3, 3, 3, 3]
    (r,) = _launch_graph(1, sizes, device_comm=False, backend="nccl")
    assert r[9] == "rccl-graph" and r[2] == 1 and "RCCL all-reduce" in r[7], (r[9], r[2], r[7])
    assert r[6] and np.isfinite(r[1]).all()
    sched = [w for w in r[8] if "second call" not in w]
    assert len(sched) == len(r[8]) and len(sched) <= MAX_COLLECTIVES, r[8]  # the f64 tail rides in the gradients' RCCL group
    (e,) = _launch_graph(1, sizes, device_comm=False, backend="nccl", eager=True)
    assert e[9] == "rccl-graph"
    assert np.array_equal(r[1], e[1])
    # ... and the segmented replay (torch.distributed between graph segments) is still there when asked for
    (sg,) = _launch_graph(1, sizes, device_comm=False, backend="nccl", rccl_graph=False)
    assert sg[9] == "rccl-segments" and sg[2] > 5
    # (one stream and the Gram matrix first there, branch streams and the `late` schedule here: the layer-1 statistics of
    #  two passes come from other sums -- the same model up to rounding)
    rel = np.linalg.norm(r[1] - sg[1]) / np.linalg.norm(sg[1])
    assert rel <= 1e-4, rel


def test_rccl_binding_sums_a_vector_and_its_tail_eager_and_captured():
    """csrc/rccl.hip on its own (world size 1: the sum is the identity): every dtype, with and without the f64 tail, eager
    and replayed from a graph; the entry refuses a null communicator."""
    import ctypes

    import torch.distributed as dist

    from mggan import devcomm
    from mggan.hip.lib import HipError, lib

    assert lib.mggan_rccl_available() == 1
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        comm = devcomm.RcclComm(None, dev)
        for dt, n in ((torch.float32, 90001), (torch.float64, 33), (torch.int32, 8)):
            x = (torch.arange(n, device=dev) % 97).to(dt)
            tail = torch.arange(17, dtype=torch.float64, device=dev) if dt == torch.float32 else None
            want, want_t = x.clone(), None if tail is None else tail.clone()
            comm.all_reduce_(x, tail)
            torch.cuda.synchronize()
            assert torch.equal(x, want) and (tail is None or torch.equal(tail, want_t))
        x = torch.full((1000,), 3.0, device=dev)
        tail = torch.full((9,), 2.0, dtype=torch.float64, device=dev)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            comm.all_reduce_(x, tail)
            with torch.cuda.graph(g, stream=s):
                comm.all_reduce_(x, tail)
                x.mul_(2.0)
        for k in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert bool((x == 24.0).all()) and bool((tail == 2.0).all())
        comm.check()
        with pytest.raises(HipError):
            lib.mggan_rccl_allreduce(0, x.data_ptr(), 4, 0, 0, 0, 0)
        comm.close()
    finally:
        dist.destroy_process_group()


def _run_allreduce(rank, world, port, q, backend="gloo", own_device=False):
    import sys

    for p in (os.path.join(ROOT, "mg-gan_amd"), ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from mggan import devcomm

    dev = torch.device("cuda", rank if own_device else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    comm = devcomm.create(None, dev)
    assert comm is not None, "peer mapping failed"
    ok = True
    gen = torch.Generator().manual_seed(3)  # every rank draws the same table and takes its row
    from mggan.hip import functions as HF

    side = HF._BR["streams"][0] = torch.cuda.Stream()  # a branch stream has a channel (arena) of its own
    for rnd, (n, dt) in enumerate([(5, torch.float64), (90000, torch.float32), (32, torch.float64), (8, torch.int32),
                                   (2049, torch.float32), (131072, torch.float32), (1, torch.float32)] * 3):
        if dt == torch.int32:
            table = torch.randint(-50, 50, (world, n), generator=gen, dtype=torch.int32)
        else:
            table = torch.randn(world, n, generator=gen, dtype=dt)
        want = table[0].clone()
        for j in range(1, world):  # rank order, like the kernel
            want += table[j]
        x = table[rank].to(dev)
        with torch.cuda.stream(side if rnd % 2 else torch.cuda.current_stream()):  # two channels
            comm.all_reduce_(x)
        torch.cuda.synchronize()
        ok = ok and torch.equal(x.cpu(), want)
    # captured: the sequence counter lives on the device, a replay is a fresh collective
    x = torch.full((1000,), float(rank + 1), device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        comm.all_reduce_(x)  # warm the channel of this stream outside the capture
        x.fill_(float(rank + 1))
        with torch.cuda.graph(g, stream=s):
            comm.all_reduce_(x)
    for _ in range(3):
        x.fill_(float(rank + 1))
        g.replay()
        torch.cuda.synchronize()
        ok = ok and bool((x == sum(range(1, world + 1))).all())
    comm.check()
    ok = ok and not comm.failed()
    dist.barrier()
    comm.close()
    dist.destroy_process_group()
    q.put((rank, ok))


def _launch_allreduce(world, backend="gloo", own_device=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_allreduce, args=(r, world, port, q, backend, own_device)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_device_allreduce_is_exact_and_capturable():
    """csrc/comm.hip on its own: f32 / f64 / i32 vectors from 1 element to a full slot, on two channels, eager and
    replayed from a graph, equal the rank-ordered sum to the bit."""
    assert all(ok for _, ok in _launch_allreduce(2))


def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts two ranks itself and prints n_gpus: 2 (here
    the ranks share the box's one GPU over gloo -- a functional check of the launcher and of the line's transport A/B)."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, MGGAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--config", "c1", "--also", "", "--no-cpu-baseline", "--transport-ab"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    kinds = set().union(*[set(t) - {"config"} for t in line["collective_transports"]])
    assert kinds == {"peer-mapped", "rccl-segments"}, kinds  # (two ranks on one device: RCCL inside the graph is refused)


def _run_lost_peer(rank, world, port, q):
    import sys

    for p in (os.path.join(ROOT, "mg-gan_amd"), ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      MGGAN_COMM_TIMEOUT_S="2")
    import torch.distributed as dist

    from mggan import devcomm

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = devcomm.create(None, dev)
    assert comm is not None and not comm.failed()
    x = torch.ones(5000, device=dev)
    if rank == 0:  # rank 1 never issues this collective: a lost peer
        comm.all_reduce_(x)
        torch.cuda.synchronize()
    dist.barrier()
    raised = False
    try:
        comm.check(sync=False)
    except RuntimeError:
        raised = True
    q.put((rank, bool(torch.isnan(x).all().cpu()), comm.failed(), raised))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_lost_peer_poisons_the_result_and_is_reported_without_a_sync():
    """A wait that exceeds the bound must not hand back a sum over stale slots: the vector comes back NaN and the
    host-mapped error word (read every iteration by the training loop) is set."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_lost_peer, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, rest) for r, *rest in [q.get(timeout=200) for _ in range(2)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == [True, True, True]      # rank 0 waited 2 s for a peer that never came
    assert res[1] == [False, False, False]   # rank 1 did nothing and saw nothing
