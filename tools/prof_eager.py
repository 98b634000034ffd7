#!/usr/bin/env python
"""Host-side profile of EAGER iterations (what train() pays on batch shapes its graph cache has not captured):
   python tools/prof_eager.py [c1|c2] [host|device]  -> cProfile top functions by own time and by cumulative time."""
import cProfile
import os
import pstats
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mg-gan_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from mggan.data_utils import synthetic  # noqa: E402

if os.environ.get("MGGAN_AUTOGRAD_ONE_THREAD") == "1":  # backward nodes on the calling thread: visible to cProfile
    torch.autograd.set_multithreading_enabled(False)
tag = sys.argv[1] if len(sys.argv) > 1 else "c1"
rng = sys.argv[2] if len(sys.argv) > 2 else "device"
c = bench.CONFIGS[tag]
dev = torch.device("cuda", 0)
torch.set_num_threads(4)
tr = bench.build_trainer(c["num_gens"], rng, dev)
batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(c["scenes"], c["peds"]), seed=0))
batch["loss_mask"] = None
tr.defer_metrics = True
tr.zero_grads_in_step = True
m = defaultdict(list)
for _ in range(5):
    tr.train_iteration(batch, m)
tr.flush_metrics()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    tr.train_iteration(batch, m)
t_host = time.perf_counter() - t0
tr.flush_metrics()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("{} {}: host {:.3f} ms / iteration to enqueue, {:.3f} ms / iteration until the GPU is done".format(
    tag, rng, t_host / n * 1e3, t_all / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    tr.train_iteration(batch, m)
pr.disable()
tr.flush_metrics()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    print("=" * 30, key)
    pstats.Stats(pr).strip_dirs().sort_stats(key).print_stats(45)
