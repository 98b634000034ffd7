#!/usr/bin/env python
"""Which Python lines still launch ATen (PyTorch) kernels inside one training iteration?  north_star: PyTorch tensors
are storage only, so every entry listed here is a defect to remove.  Runs one eager iteration under the torch
profiler with Python stacks and prints, per ATen operator that launched a device kernel, the innermost frame of
this repository.      python tools/aten_sites.py [scenes] [peds] [num_gens]"""
import os
import sys
from collections import Counter, defaultdict

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "mg-gan_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from mggan.data_utils import synthetic  # noqa: E402


def main(scenes=64, peds=20, num_gens=4):
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(num_gens, "device", dev)
    batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(scenes, peds), seed=0))
    batch["loss_mask"] = None
    tr.defer_metrics = True
    tr.zero_grads_in_step = True
    m = defaultdict(list)
    for _ in range(3):
        tr.train_iteration(batch, m)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.train_iteration(batch, m)
        torch.cuda.synchronize()
    sites = Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        kern = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
        if not kern:
            continue
        frame = next((s for s in ev.stack if "mg-gan_amd" in s or "bench.py" in s), ev.stack[0] if ev.stack else "?")
        frame = frame.replace(R + "/", "")
        sites[(ev.name, frame, kern[0].name[:60])] += 1
    print("# ATen operators that launched a device kernel in ONE iteration ({} scenes x {} peds, g={}): {} launches".format(
        scenes, peds, num_gens, sum(sites.values())))
    for (name, frame, k), c in sorted(sites.items(), key=lambda x: -x[1]):
        print("{:3d}  {:<28s} {:<70s} {}".format(c, name, frame, k))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
