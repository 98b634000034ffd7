#!/usr/bin/env python
"""Where one iteration's time goes, measured without a profiler: MGGAN_MARKS=1 makes the trainer drop one-lane
timestamp kernels (mggan_timestamp: the device's 100 MHz wall clock) at the step boundaries and the fork / join
points of its stream graph; they are captured into the iteration's HIP graph and read back after a replay.
(rocprofv3 --kernel-trace serialises parts of a multi-stream graph and stretches the iteration by ~15 %;
the marks cost ~2 us each.)      python tools/time_marks.py [scenes] [peds] [num_gens]"""
import os
import sys
import time
from collections import defaultdict

os.environ["MGGAN_MARKS"] = "1"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "mg-gan_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from mggan.data_utils import synthetic  # noqa: E402
from mggan.hip import functions as HF  # noqa: E402


def mark_every_backward_node():
    """MGGAN_MARKS_NODES=1: one more mark behind every autograd node's backward (on the stream it ran on)."""
    import inspect

    for name, cls in list(vars(HF).items()):
        if inspect.isclass(cls) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            inner = cls.backward

            def wrapped(ctx, *g, _inner=inner, _name=name):
                out = _inner(ctx, *g)
                HF.mark("node." + _name)
                return out

            cls.backward = staticmethod(wrapped)


def main(scenes=64, peds=20, num_gens=4):
    if os.environ.get("MGGAN_MARKS_NODES") == "1":
        mark_every_backward_node()
    dev = torch.device("cuda", 0)
    if os.environ.get("MGGAN_FORCE_DIST", "0") == "1":  # the sharded launch mode on one rank (as bench.py sets it up)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev)
        dist.init_process_group(os.environ.get("MGGAN_DIST_BACKEND", "nccl"), rank=0, world_size=1, device_id=dev)
    tr = bench.build_trainer(num_gens, "device", dev)
    if tr.dist.enabled:
        tr.dist.equal_shards = True  # (every rank holds the same number of images: no count exchange, capturable)
    batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(scenes, peds), seed=0))
    batch["loss_mask"] = None
    tr.defer_metrics = True
    tr.zero_grads_in_step = True
    m = defaultdict(list)
    HF._MARKS["on"] = False
    for _ in range(3):
        tr.train_iteration(batch, m)
    torch.cuda.synchronize()
    HF._MARKS["on"] = True
    replay = tr.capture_iteration(batch, warmup=0)
    HF._MARKS["on"] = False
    for _ in range(10):
        replay(m, False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        replay(m, False)
    torch.cuda.synchronize()
    print("# {} scenes x {} pedestrians, {} generators: {:.3f} ms per iteration with the marks in the graph".format(
        scenes, peds, num_gens, (time.perf_counter() - t) / 50 * 1e3))
    print("# microseconds since the first mark of the iteration; one replay")
    replay(m, False)
    replay(m, False)
    torch.cuda.synchronize()
    for name, us in HF.read_marks():
        print("{:9.1f}  {}".format(us, name))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
