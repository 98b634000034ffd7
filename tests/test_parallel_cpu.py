"""CPU, world_size 2 over gloo: scene sharding + the three collectives of the data-parallel path
(C1 flat gradient all-reduce, C2 BatchNorm-statistics all-reduce, C3 global counts) reproduce the
single-process result.  The compute under test here is the ORACLE (tests may use it): what is
verified is the sharding / reduction logic in mggan.parallel, which the HIP trainer calls unchanged."""
import os
import socket
from collections import defaultdict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mggan.parallel import DistContext, shard_batch, shard_scenes


def test_shard_scenes_partitions_contiguously():
    sse = [[0, 3], [3, 4], [4, 9], [9, 10], [10, 16], [16, 18]]
    for world in (1, 2, 3, 4, 8):
        seen, covered = [], 0
        for r in range(world):
            sl, p0, p1, local = shard_scenes(sse, r, world)
            seen += list(range(sl.start, sl.stop))
            covered += p1 - p0
            if local:
                assert local[0][0] == 0 and local[-1][1] == p1 - p0
                assert [e - s for s, e in local] == [e - s for s, e in sse[sl]]
        assert seen == list(range(len(sse))) and covered == 18


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mg-gan_amd"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    import mggan_oracle as O
    from mggan.data_utils import synthetic

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ctx = DistContext()
    assert ctx.enabled and ctx.world_size == world

    sizes = [3, 1, 4, 2, 5, 3]
    full = synthetic.make_batch(sizes, seed=5)
    mine = shard_batch(full, rank, world)
    b_glob = full["in_xy"].shape[1]

    # ---- C2: BatchNorm statistics: (sum, sumsq, count) all-reduced == whole-batch statistics
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(4, 8, 3, 1, 1)
    y_loc = conv(mine["features"]).detach()
    sums = torch.cat([y_loc.double().sum((0, 2, 3)), (y_loc.double() ** 2).sum((0, 2, 3))])
    n = ctx.all_reduce_stats(sums, float(y_loc.shape[0]))
    y_all = conv(full["features"]).detach().double()
    assert n == float(b_glob)
    cnt = n * 33 * 33
    mean, var = sums[:8] / cnt, sums[8:] / cnt - (sums[:8] / cnt) ** 2
    assert torch.allclose(mean, y_all.mean((0, 2, 3)), atol=1e-10)
    assert torch.allclose(var, y_all.var((0, 2, 3), unbiased=False), atol=1e-10)

    # ---- C3: generator counts
    idx = torch.randint(0, 4, (b_glob, 20), generator=torch.Generator().manual_seed(1))
    _, p0, p1, _ = shard_scenes(full["seq_start_end"], rank, world)
    counts = torch.bincount(idx[p0:p1].flatten(), minlength=4).to(torch.int32)
    ctx.all_reduce_(counts)
    assert torch.equal(counts, torch.bincount(idx.flatten(), minlength=4).to(torch.int32))

    # ---- C1: a loss that is a mean over the GLOBAL batch, computed shard-wise with 1/N_global scaling,
    # gives the single-process gradient after one all-reduce(sum) of the flat gradient buffer
    torch.manual_seed(3)
    G, _ = O.construct_oracle(2)
    G.eval()  # BN in eval mode: this check isolates the gradient reduction (C2 is checked above)

    def loss_of(batch, norm):
        enc_h, _ = G.trunk(batch["in_xy"], batch["in_dxdy"], batch["seq_start_end"], batch["features"], "block")
        return (G.logits(enc_h) ** 2).sum() / norm

    G.zero_grad()
    loss_of(full, b_glob).backward()
    ref = torch.cat([p.grad.flatten() for p in G.parameters() if p.grad is not None])
    G.zero_grad()
    loss_of(mine, b_glob).backward()

    class Flat:
        pass

    params = [p for p in G.parameters() if p.grad is not None]
    f = Flat()
    f._flat_grad = torch.cat([p.grad.flatten() for p in params])
    ctx.all_reduce_grads(f)
    assert torch.allclose(f._flat_grad, ref, rtol=1e-4, atol=1e-6), float((f._flat_grad - ref).abs().max())
    # ---- the epoch-end metric flush of train()'s graph cache: ONE exchange issued by every rank, whatever it replayed
    # (rank 0 replayed three iterations of one bucket, the others none: before, they skipped the collective and rank 0 hung)
    from mggan.abstract_train import IterationGraphs

    class StubTrainer:
        dist = ctx

    ig = IterationGraphs(StubTrainer())
    if rank == 0:
        ig._acc[17] = [torch.zeros(4), torch.tensor([3.0, 6.0, 9.0, 0.0]), 3, {(("loss/a", 0), ("loss/b", (1, 2)))}]
        ig.replays = 3
    m = defaultdict(list)
    ig.flush(m)
    if rank == 0:
        assert m["loss/a"] == [1.0 * world] * 3 and m["loss/b"] == [5.0 * world] * 3, dict(m)
        assert ig._acc[17][2] == 0 and float(ig._acc[17][1].abs().sum()) == 0.0 and ig.history == [(3, 0, 0)]
    else:
        assert not m
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharded_equals_single_process(world):
    """world 2 and world 4 (six scenes over four ranks: uneven shards, 1-2 scenes each)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, p.exitcode
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, "ok") for r in range(world)]
