#!/usr/bin/env python
"""BASELINE.md section 4 promises that the oracle's *faithful* mode (the reference's own operator sequence) reproduces the
reference's step times (the survey container: 8 vCPU Xeon @ 2.1 GHz, 8 OpenMP threads) within
+-15 %.  This script times, in the BUILD container and under the same protocol (child process pinned to 8 cores, one
warm-up, median of 5 iterations timed one by one):
  * the oracle in faithful mode (bench.py's CPU-baseline worker), and
  * the REAL reference (imported from /root/reference with the stubs of tests/golden/_refload.py), on the same batch,
and writes both next to the survey's figure into profiles/r05_cpu_baseline_validation.txt.  Round 5: 15 iterations per
row instead of 5 and the ratio of the MINIMA beside the ratio of the medians -- round 4's medians of five sat inside a
2x spread of the reference's own step time (0.18..0.43 s at b = 6), which is what put two of its three rows outside +-15 %.
    python tools/validate_cpu_baseline.py"""
import json
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [  # (num_gens, scenes of 3 pedestrians, survey seconds per iteration)
    (1, 2, 0.111),
    (4, 2, 0.180),
    (4, 8, 4.03),
]
THREADS = 8
ITERS = 15


def ref_worker(spec):
    spec = json.loads(spec)
    os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:spec["threads"]])
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import importlib.util
    from collections import defaultdict

    import numpy as np
    import torch
    import _refload

    torch.set_num_threads(spec["threads"])
    ref_train, ref_config = _refload.load_reference()
    import test_tube

    sp = importlib.util.spec_from_file_location("_synth", os.path.join(ROOT, "mg-gan_amd", "mggan", "data_utils", "synthetic.py"))
    synth = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(synth)
    args = ref_config.get_parser().parse_args(["--gpus", "", "--num_gens", str(spec["num_gens"])])
    torch.manual_seed(145325)
    np.random.seed(435346)
    G, D = ref_train.construct_model(args)
    model = ref_train.PiNetMultiGeneratorGAN(G, D, args, test_tube.Experiment())
    G.train()
    D.train()
    batch = synth.make_batch(spec["sizes"], seed=0)
    a = [batch[k] for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy")] + [batch["seq_start_end"]]
    mask = ~batch["gt_xy"].isnan().any(2).any(0)

    def iteration():  # the loop body, abstract_train.py:136-159
        m = defaultdict(list)
        model.discriminator_step(*a, m, mask, batch["features"])
        model.generator_step(*a, m, mask, batch["features"])
        model.net_chooser_step(*a, m, mask, batch["features"])

    iteration()
    secs = []
    for _ in range(spec["iters"]):
        t0 = time.perf_counter()
        iteration()
        secs.append(time.perf_counter() - t0)
    print(json.dumps({"seconds": secs}), flush=True)


def child(cmd, env):
    r = subprocess.run(cmd, env=env, check=True, capture_output=True, text=True)
    secs = sorted(json.loads(r.stdout.strip().splitlines()[-1])["seconds"])
    return secs[len(secs) // 2], secs[0], secs[-1]


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--ref-worker":
        return ref_worker(sys.argv[2])
    lines = ["CPU oracle (faithful mode) vs the real reference, timed side by side in the build container, and vs the survey's",
             "figures (BASELINE.md section 2).  Protocol: child process pinned to {} cores, 1 warm-up, median of {} (min..max).".format(THREADS, ITERS),
             "host: {} | nproc {} | python {}".format(platform.machine(), os.cpu_count(), platform.python_version()), "",
             "g  scenes b   survey_s  reference_now_s         oracle_faithful_s       oracle/reference_now  (of minima)  oracle/survey"]
    env = dict(os.environ, OMP_NUM_THREADS=str(THREADS), MKL_NUM_THREADS=str(THREADS), HIP_VISIBLE_DEVICES="")
    for g, scenes, survey_s in ROWS:
        spec = {"sizes": [3] * scenes, "num_gens": g, "iters": ITERS, "mode": "faithful", "threads": THREADS}
        o = child([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", json.dumps(spec)], env)
        r = child([sys.executable, os.path.abspath(__file__), "--ref-worker", json.dumps(spec)], env)
        lines.append("{:<2d} {:<6d} {:<3d} {:<9.3f} {:<7.3f} ({:.3f}..{:.3f})   {:<7.3f} ({:.3f}..{:.3f})   {:<21.3f} {:<12.3f} {:.3f}".format(
            g, scenes, 3 * scenes, survey_s, r[0], r[1], r[2], o[0], o[1], o[2], o[0] / r[0], o[1] / r[1], o[0] / survey_s))
        print(lines[-1], flush=True)
    lines += ["", "Reading: `oracle/reference_now` within 0.85..1.15 = the port costs what the reference costs on the same cores",
              "at the same moment; `oracle/survey` also carries the difference between this container's load and the survey's."]
    out = os.path.join(ROOT, "profiles", "r05_cpu_baseline_validation.txt")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
