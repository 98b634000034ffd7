"""GPU, two or more physical devices (skipped on a one-GPU box): one rank per GPU, backend nccl (= RCCL over xGMI).
What tests/test_hip_ddp.py checks with two processes on ONE device is repeated across real devices: the peer-mapped
all-reduce (csrc/comm.hip -- system-scope visibility of uncached peer stores over xGMI, hipIpcMemLazyEnablePeerAccess,
flag fan-out to every rank), the sharded iteration as one graph, and the same iteration on RCCL collectives between graph
segments (MGGAN_DEVICE_COMM=0)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_hip_ddp import ROOT, _check_replicas, _launch_allreduce, _launch_graph

N_DEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_DEV < 2, reason="needs two or more HIP devices")]
WORLDS = sorted({2, N_DEV} if N_DEV >= 2 else {2})


@pytest.mark.parametrize("world", WORLDS)
def test_peer_mapped_allreduce_across_devices(world):
    assert all(ok for _, ok in _launch_allreduce(world, backend="nccl", own_device=True))


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_iteration_one_graph_across_devices(world):
    sizes = [3] * (2 * world)  # equal shards: two scenes of three pedestrians per rank
    res = _launch_graph(world, sizes, device_comm=True, backend="nccl", own_device=True)
    _check_replicas(res)
    assert res[0][5] and res[0][2] == 1 and "peer-mapped" in res[0][7]
    # the same global batch in ONE process (collective hooks forced on): same weights up to the reduction order
    (single,) = _launch_graph(1, sizes, device_comm=True)
    rel = np.linalg.norm(res[0][1] - single[1]) / np.linalg.norm(single[1])
    assert rel <= 1e-3, rel


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_iteration_rccl_segments_across_devices(world):
    sizes = [3] * (2 * world)
    res = _launch_graph(world, sizes, device_comm=False, backend="nccl", own_device=True, rccl_graph=False)
    _check_replicas(res)
    assert res[0][2] > 10 and not res[0][5] and res[0][9] == "rccl-segments"  # RCCL collectives between graph segments
    peer = _launch_graph(world, sizes, device_comm=True, backend="nccl", own_device=True)
    rel = np.linalg.norm(res[0][1] - peer[0][1]) / np.linalg.norm(peer[0][1])
    assert rel <= 1e-4, rel  # both transports train the same model (other summation order across the ranks)


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_iteration_rccl_inside_one_graph_across_devices(world):
    """ncclAllReduce issued by the library on the capturing stream (csrc/rccl.hip): one graph per rank, replicas
    bit-identical, the same model as the peer-mapped kernels train."""
    sizes = [3] * (2 * world)
    res = _launch_graph(world, sizes, device_comm=False, backend="nccl", own_device=True)
    _check_replicas(res)
    assert all(r[9] == "rccl-graph" and r[2] == 1 and "RCCL all-reduce" in r[7] for r in res), [(r[9], r[2]) for r in res]
    peer = _launch_graph(world, sizes, device_comm=True, backend="nccl", own_device=True)
    rel = np.linalg.norm(res[0][1] - peer[0][1]) / np.linalg.norm(peer[0][1])
    assert rel <= 1e-4, rel


def test_bench_self_launch_across_devices():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--also", "", "--no-cpu-baseline", "--transport-ab"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert {"peer-mapped", "rccl-graph", "rccl-segments"} <= set().union(*[set(t) for t in line["collective_transports"]])
