#!/usr/bin/env python
"""bench.py -- MG-GAN train-step trajectories/sec on MI355X (BASELINE.json metric).

One "step" = one full iteration of the reference loop body (abstract_train.py:114-168):
discriminator step + generator step + PM-network step, each with forward, backward, gradient
clipping and an AdamW update, on a synthetic batch already resident in HBM.
Headline workload: BASELINE.json configs[1] -- 64 scenes x 20 pedestrians, num_gens=4, K=20 samples; the same
run also measures configs[2] (256 scenes x 32 pedestrians, num_gens=8) with the same protocol into `configs`.
N>1 (launched by torch.distributed.run, one rank per GPU): every rank owns that many scenes of an N-times larger
global batch (weak scaling: the 64x20 shard is configs[4]'s, the 256x32 g=8 shard configs[3]'s); gradients /
BatchNorm statistics / generator counts are exchanged between the ranks.
value = N * b_local * steps / max-over-ranks(time).

Prints ONE JSON line (rank 0) with `roofline` for the dominant C-ABI entry (timed live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle, block-diagonal mode, timed on
this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mg-gan_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

F32_PEAK_TFLOPS = 157.3  # MI355X dense f32 (vector == f32-MFMA rate), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def kernel_of(name, launches):
    """C-ABI entry -> the HIP kernel SYMBOL its time is booked on, spelled as the rocprofv3 summaries and the counter
    tables under profiles/ spell it (mggan/hip/ksym.py: name<every template argument>).  `launches` is what the library's
    launch log noted for this very call ([(symbol, threads)], include/mggan_hip.h: mggan_launch_log): the kernel with the
    most threads is the entry's kernel, its other launches are finalize / fold tails.  No table to keep in step with the
    launchers' thresholds: the name is the one the runtime launched."""
    from mggan.hip.ksym import primary

    sym = primary(launches)
    if sym is None:  # an entry that launched nothing (e.g. an empty batch of weight gradients)
        return "(no launch: {})".format(name[len("mggan_"):] if name.startswith("mggan_") else name)
    return sym


def operator_flops_of(name, a):
    """FLOPs of the REFERENCE operator an entry stands for (SURVEY App. D shapes) where the entry executes fewer: the
    kernels that compute a per-pedestrian part once for the K rows of a pedestrian.  The iteration total is counted in
    these (the same accounting as earlier rounds); the per-kernel roofline uses what the kernel executes (flops_of)."""
    heads = lambda g: 2.0 * (2 * 192 * 96 + 96 * (1 + g))
    pe = 2.0 * (24 * 64 + 64 * 32)
    if name == "mggan_dheads_lean_fwd":
        return (a[4] - a[3]) * heads(a[6])
    if name == "mggan_dheads_lean_bwd":
        return (a[5] - a[4]) * heads(a[6])
    if name == "mggan_d_rows_lean_fwd":
        return (a[3] - a[2]) * (pe + heads(a[5]))
    if name == "mggan_d_rows_lean_bwd":
        return (a[6] - a[5]) * (pe + heads(a[7]))
    if name in ("mggan_dheads_shared", "mggan_decoder_e2d_shared"):
        return 0.0  # (their products are part of the row operators counted above / below)
    if name == "mggan_decoder_rollout_fwd" and a[31]:
        return flops_of(name, a) + float(a[0]) * 2 * a[4] * a[3]
    if name == "mggan_decoder_rollout_bwd_fused" and not a[24]:
        return flops_of(name, a) + float(a[21]) * 2 * a[4] * a[3]
    return flops_of(name, a)


def flops_of(name, a):
    """Algorithmic FLOPs (2*MAC, reference operator shapes) of one C-ABI call."""
    if name == "mggan_linear_fwd":
        return 2.0 * a[6] * a[7] * a[8]
    if name == "mggan_linear_bwd_data":
        return 2.0 * a[6] * a[7] * a[8]
    if name == "mggan_wgrad":
        return 2.0 * a[7] * a[8] * a[9]
    if name == "mggan_wgrad_multi":  # a batch of weight gradients: the launcher notes the FLOPs of each batch
        from mggan.hip import functions as HF

        notes = HF.TRACE_NOTES["wgrad_multi_flops"]
        return notes.pop(0) if notes else 0.0
    if name == "mggan_mlp_chain":  # fused MLP chain: the launcher notes the FLOPs of every launch
        from mggan.hip import functions as HF

        notes = HF.TRACE_NOTES["mlp_chain_flops"]
        return notes.pop(0) if notes else 0.0
    if name == "mggan_lstm_encoder_fwd":
        T, b, H = a[1], a[2], a[3]
        E = H // 2 if H == 32 else H
        return float(T) * b * (2 * 2 * E + 2 * (E + H) * 4 * H)
    if name == "mggan_lstm_encoder_bwd":
        T, b, H = a[2], a[3], a[4]
        return float(T) * b * 2 * (4 * H * H)
    if name == "mggan_decoder_rollout_fwd":
        R, T, H, EIN, Z = a[0], a[1], a[3], a[4], a[5]
        e2d = Z if a[31] else EIN + Z  # (Qe given: the enc_h part of h0 came once per pedestrian, mggan_decoder_e2d_shared)
        return float(R) * (2 * e2d * H + T * (2 * 2 * 16 + 2 * (16 + H) * 4 * H + 2 * (2 * H * (H // 2) + (H // 2) * 2)))
    if name == "mggan_decoder_e2d_shared":
        return 2.0 * a[2] * a[3] * 32
    if name == "mggan_decoder_rollout_bwd_fused":  # BPTT data path + the fused per-generator weight gradients
        T, H, EIN, R = a[2], a[3], a[4], a[21]
        data = T * 2 * (4 * H * H + 2 * 4 * H + H * (H // 2) + (H // 2) * 2) + 2 * ((EIN if a[24] else 0) * H + H * (H // 2))
        wgrads = T * 2 * (4 * H * (H + 3) + (H // 2) * H + 2 * (H // 2))
        return float(R) * (data + wgrads)
    if name == "mggan_conv1_pool":
        return float(a[1]) * 2 * 33 * 33 * a[2] * 36
    if name == "mggan_conv2_fwd2":
        return float(a[1]) * 2 * 256 * a[2] * a[2] * 9
    if name == "mggan_conv2_bwd":
        return float(a[1]) * 2 * 2 * 256 * a[2] * a[2] * 9
    if name == "mggan_conv1_wgrad":  # the reference operator: a dense (C x 36) x (33*33 positions) weight gradient
        return float(a[1]) * 2 * 33 * 33 * a[2] * 36
    if name == "mggan_dheads_fwd":  # both heads: 2 x Linear(192,96) + Linear(96,1) + Linear(96,g)
        return 2.0 * a[2] * (2 * 192 * 96 + 96 * (1 + a[3]))
    if name == "mggan_dheads_bwd_data":  # their input gradient
        return 2.0 * a[5] * (2 * 192 * 96 + 96 * (1 + a[6]))
    # the lean heads (sample blocks >= 1 of the generator step's pass): the products they EXECUTE -- the per-pedestrian
    # part is shared by the K rows of a pedestrian, so this is less than the reference operator's 192-wide row product
    if name == "mggan_dheads_lean_fwd":
        return 2.0 * (a[4] - a[3]) * (32 * 192 + 96 * (1 + a[6]))
    if name == "mggan_dheads_lean_bwd":
        return 2.0 * (a[5] - a[4]) * (32 * 192 + 96 * (1 + a[6]))
    if name == "mggan_dheads_shared":
        return 2.0 * a[2] * 96 * 192
    if name == "mggan_d_rows_lean_fwd":  # + the pred_encoder (24 -> 64 -> 32) in front
        return 2.0 * (a[3] - a[2]) * (24 * 64 + 64 * 32 + 32 * 192 + 96 * (1 + a[5]))
    if name == "mggan_d_rows_lean_bwd":
        return 2.0 * (a[6] - a[5]) * (24 * 64 + 64 * 32 + 32 * 192 + 96 * (1 + a[7]))
    if name == "mggan_image_gram":  # not in the reference's operator list (bookkeeping of the factorised conv1 gradient)
        return 0.0
    if name == "mggan_scene_attention_fwd":
        return float(a[1]) * 64 * 2 * (a[2] * 32 * 2)
    if name == "mggan_scene_attention_bwd":  # two data adjoints + the two weight gradients (the forward recomputation is not counted)
        return float(a[2]) * 64 * 2 * (a[3] * 32 * 2) * 2  # (ysel, ycode, B, C, ...)
    if name == "mggan_social_pairs_fwd":
        return float(a[0]) * 2 * (96 + 2048 + 64)
    if name == "mggan_social_pairs_bwd":
        return float(a[0]) * 2 * (64 + 2048)
    if name in ("mggan_social_rows_fwd", "mggan_social_rows_bwd"):
        # per in-scene ordered pair (the launcher notes the pair count): forward 3->32->64 pair MLP + score + pooling;
        # backward softmax / score / layer-2 adjoints and, when the pair MLP trains (partials given), its weight gradients
        from mggan.hip import functions as HF

        notes = HF.TRACE_NOTES[name[6:] + "_pairs"]
        P, H = (notes.pop(0) if notes else 0), a[2]
        if name.endswith("fwd"):
            return float(P) * 2 * (96 + 2048 + 64 + H)
        return float(P) * 2 * (64 + 3 * H + 65 + (2048 + 2048 + 96 if a[28] else 0))
    return 0.0


def bytes_of(name, a):
    """Algorithmic HBM bytes of one C-ABI call for the entries that stream large operands once (DESIGN.md section 7
    states the per-unit figures): every saved activation / operand read or written exactly once, weights ignored."""
    if name == "mggan_wgrad_multi":  # both operands of every product of the batch, once
        from mggan.hip import functions as HF

        notes = HF.TRACE_NOTES["wgrad_multi_bytes"]
        return notes.pop(0) if notes else 0.0
    if name == "mggan_decoder_rollout_fwd":  # saves per (row, step): gates 4H, (c,h) 2H, activations H/2, input 2 (+ outputs)
        R, T, H = a[0], a[1], a[3]
        save = 0.0 if not a[25] else 4.0 * (4 * H + 2 * H + H // 2 + 2)
        return float(R) * T * (save + 4.0 * 4)
    if name == "mggan_decoder_rollout_bwd_fused":  # reads the same record back, plus the two output gradients
        T, H, R = a[2], a[3], a[21]
        return float(R) * T * 4.0 * (4 * H + 2 * H + H // 2 + 2 + 4)
    if name == "mggan_image_gram":  # the batch's images, once
        return float(a[1]) * 4 * 33 * 33 * 4.0
    return 0.0


def build_trainer(num_gens, rng, device, seed=0):
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    # (MGGAN_BENCH_BN_SYNC=local: per-rank BatchNorm statistics in sharded runs -- an exploration knob; the default
    #  keeps the global-batch statistics, i.e. the single-process results)
    cfg = get_parser().parse_args(["--num_gens", str(num_gens), "--rng", rng, "--epochs", "500", "--bn_sync",
                                   os.environ.get("MGGAN_BENCH_BN_SYNC", "global")])
    torch.manual_seed(145325)
    np.random.seed(435346)
    import io
    import contextlib

    with contextlib.redirect_stdout(io.StringIO()):
        G, D = construct_model(cfg)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    if rng == "device":
        torch.cuda.manual_seed(1234 + seed)  # replicas share the weights, every rank draws its own noise
    tr.G.train()
    tr.D.train()
    return tr


def marked_replay(tr, batch, entries, replays=8):
    """Durations of the launches of `entries` (C-ABI names) inside graph replay: {kernel symbol: ms per iteration}.
    A second capture of the same iteration carries a mggan_timestamp launch before and after every call of those entries
    on the call's own stream; the fixed cost of a pair of marks (two back-to-back marks on an idle stream segment measure
    it: ~2-3 us) is subtracted per launch."""
    from mggan.hip.lib import load

    L = load()
    buf = torch.zeros(2 * 512, dtype=torch.int64, device=batch["in_xy"].device)
    L.marks = {"names": set(entries), "buf": buf, "calls": []}
    L.launch_log(True)
    try:
        replay = tr.capture_iteration(batch, warmup=0)
    finally:
        mk, L.marks = L.marks, None
        L.launch_log(False)
    calls = mk["calls"]
    # calibration: the gap between two consecutive marks with nothing between them
    cal = torch.zeros(2, dtype=torch.int64, device=buf.device)
    s = torch.cuda.current_stream().cuda_stream
    gaps = []
    for _ in range(5):
        L._c.mggan_timestamp(cal.data_ptr(), s)
        L._c.mggan_timestamp(cal.data_ptr() + 8, s)
        torch.cuda.synchronize()
        t = cal.cpu().tolist()
        gaps.append((t[1] - t[0]) / 100.0)  # 100 MHz wall clock -> us
    gap = sorted(gaps)[len(gaps) // 2]
    acc = defaultdict(float)
    for _ in range(replays):
        replay(None, False)
        torch.cuda.synchronize()
        t = buf[:2 * len(calls)].cpu().tolist()
        for i, (name, a, launches) in enumerate(calls):
            acc[kernel_of(name, launches)] += max((t[2 * i + 1] - t[2 * i]) / 100.0 - gap, 0.0) * 1e-3
    del replay
    return {k: v / replays for k, v in acc.items()}


def cpu_baseline_worker(spec):
    """Child process of cpu_baseline(): pins itself to `threads` cores BEFORE torch spawns its OpenMP team, times every
    iteration on its own and prints one JSON line."""
    spec = json.loads(spec)
    threads = spec["threads"]
    try:
        cores = sorted(os.sched_getaffinity(0))[:threads]
        os.sched_setaffinity(0, cores)
    except (AttributeError, OSError):
        cores = []
    torch.set_num_threads(threads)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mggan_oracle as O
    from mggan.data_utils import synthetic

    torch.manual_seed(145325)
    np.random.seed(435346)
    G, D = O.construct_oracle(spec["num_gens"])
    tr = O.OracleTrainer(G, D, mode=spec["mode"])
    batch = synthetic.make_batch(spec["sizes"], seed=0)
    m = defaultdict(list)
    for _ in range(spec.get("warmup", 1)):
        tr.iteration(batch, m)
    secs = []
    for _ in range(spec["iters"]):
        t0 = time.perf_counter()
        tr.iteration(batch, m)
        secs.append(time.perf_counter() - t0)
    print(json.dumps({"seconds": secs, "b": int(batch["in_xy"].shape[1]), "pinned_cores": len(cores),
                      "threads": torch.get_num_threads()}), flush=True)


def cpu_baseline(sizes, num_gens, iters, mode="block", tag="same workload"):
    """The CPU oracle timed on this box's host cores (test infrastructure used as the reported CPU baseline only):
    a child process pinned to `threads` cores, one warm-up, then `iters` (>= 5) iterations timed one by one;
    value = b / median, `spread` = (min, max) of the per-iteration rates."""
    import subprocess

    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    iters = max(5, iters)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    spec = json.dumps({"sizes": [int(x) for x in sizes], "num_gens": num_gens, "iters": iters, "mode": mode,
                       "threads": threads})
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", spec], env=env, check=True,
                       capture_output=True, text=True)
    w = json.loads(r.stdout.strip().splitlines()[-1])
    secs = sorted(w["seconds"])
    med = secs[len(secs) // 2]
    b = w["b"]
    return {"value": b / med, "unit": "trajectories/s", "cores": w["threads"], "kind": "port", "mode": mode,
            "spread": [round(b / secs[-1], 2), round(b / secs[0], 2)], "s_per_iteration": round(med, 4),
            "iterations": iters, "pinned_cores": w["pinned_cores"], "nproc": os.cpu_count(),
            "sample": "median of {} full D+G+PM iterations of {} ({} scenes, {} peds, num_gens={}) on oracle/mggan_oracle.py, "
                      "{} mode, {} pinned threads".format(iters, tag, len(sizes), b, num_gens, mode, w["threads"])}


def sharded_one_rank_leg(args, single_ms):
    """What the sharded launch mode costs on ONE rank, before a byte crosses xGMI: this command again with the collective
    hooks forced on (MGGAN_FORCE_DIST=1, world size 1) on each in-graph transport, against the single-GPU graphs of this
    run.  single_ms: {config tag: ms_per_step}.  One child process per transport (the process group is per process)."""
    import socket
    import subprocess

    out = {"note": "world size 1: every exchange kernel / RCCL call of the sharded iteration is issued, nothing waits for a "
                   "peer (RCCL's in-place all-reduce over one rank launches no kernel at all, so its column is the cost of "
                   "the unfused BatchNorm folds and tails around the calls, not of RCCL's kernels)"}
    for name, env_add in (("peer-mapped", {"MGGAN_DEVICE_COMM": "1"}),
                          ("rccl-graph", {"MGGAN_DEVICE_COMM": "0", "MGGAN_RCCL_GRAPH": "1"})):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MGGAN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                   LOCAL_RANK="0", MGGAN_BENCH_NO_DETAIL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_add)
        cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--also", args.also, "--no-floor",
               "--no-cpu-baseline", "--no-profile", "--steps", str(max(args.steps, 20)), "--warmup", str(max(args.warmup, 5))]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
            continue
        rows = []
        for c in line["configs"]:  # (the compact line spells the config tag "workload")
            base = single_ms.get(c["workload"])
            rows.append({"config": c["workload"], "ms_per_step": c["ms_per_step"], "single_graph_ms_per_step": base,
                         "overhead": round(c["ms_per_step"] / base - 1.0, 4) if base else None,
                         "collective": c.get("collective"), "exchanges_per_step": c.get("exchanges_per_step")})
        out[name] = rows
    return out


def train_loop_disk_leg(args, dev, head_ms, frames=3219, peds=40, scenes_per_batch=32, epochs=5, workers=2):
    """train() as the README runs it on an ON-DISK dataset in the reference's format (one 640 x 480 scene image, `frames`
    frames x `peds` pedestrians written as an ETH-style text file), --augment 1 (the reference's default), no --cache_device:
    every batch goes through the loader -- trajectories on the host, the augmented scene crops on the GPU
    (mggan/data_utils/device_crops.py) -- and the trainer's graph cache.  32 scenes x 40 pedestrians = the headline's 1,280
    pedestrians per batch.  The figure is the median over the later epochs of (epoch wall time) / iterations."""
    import contextlib
    import io
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_loader as BL

    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    tmp = tempfile.mkdtemp(prefix="mggan_disk_")
    was = os.environ.get("MGGAN_DATA_ROOT")
    os.environ["MGGAN_DATA_ROOT"] = tmp
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            BL.make_dataset(tmp, frames=frames, peds=peds, phase="train")
            BL.make_dataset(tmp, frames=40, peds=peds, phase="val")
            cfg = get_parser().parse_args([
                "--num_gens", str(CONFIGS["c2"]["num_gens"]), "--rng", "device", "--graph", "auto", "--dataset", "eth",
                "--augment", "1", "--epochs", str(epochs), "--batch_size", str(scenes_per_batch), "--val_every", "1000000",
                "--save_every", "1000000", "--workers", str(workers)])
            torch.manual_seed(145325)
            np.random.seed(435346)
            G, D = construct_model(cfg)
            tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
            torch.cuda.manual_seed(1234)
            tr.zero_grads_in_step = True
            last = tr.train()
    finally:
        if was is None:
            os.environ.pop("MGGAN_DATA_ROOT", None)
        else:
            os.environ["MGGAN_DATA_ROOT"] = was
    ig = tr.iteration_graphs
    per_it = sorted(s / n for s, n in list(zip(tr.epoch_seconds, tr.epoch_iterations))[2:])
    ms = per_it[len(per_it) // 2] * 1e3
    tr.dist.close()
    return {"workload": "train() on an on-disk dataset (reference format: {} frames x {} pedestrians, one 640x480 scene image), "
                        "--augment 1, --workers {} (the reference's flag: loader processes for the host half of a batch), batches of {} scenes, "
                        "num_gens={}, {} epochs; crops on the GPU, no device cache".format(
                            frames, peds, workers, scenes_per_batch, CONFIGS["c2"]["num_gens"], epochs),
            "ms_per_step": round(ms, 4), "vs_captured_iteration": round(ms / head_ms, 3),
            "iterations_per_epoch": tr.epoch_iterations[-1], "replayed_iterations": ig.replays if ig else 0,
            "eager_iterations": (ig.eager if ig else sum(tr.epoch_iterations)), "graphs": len(ig.entries) if ig else 0,
            "ms_per_step_by_epoch": [round(s / n * 1e3, 4) for s, n in zip(tr.epoch_seconds, tr.epoch_iterations)],
            "last_losses": {k: round(v, 6) for k, v in sorted(last.items()) if "probs" not in k}}


def train_loop_leg(tag, args, dev, batches=25, epochs=6):
    """MultiGeneratorGAN.train() itself (the reference's loop, abstract_train.py:114-168) on the synthetic loader at the
    headline shape: device-resident batches, --rng device, the loop's own graph cache.  Epoch 1 produces the data and runs
    the shape's first iteration eagerly, its second iteration captures; the figure is the median over the later epochs of
    (wall time of the epoch's training loop, device-synchronised) / iterations."""
    import contextlib
    import io

    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    c = CONFIGS[tag]
    cfg = get_parser().parse_args([
        "--num_gens", str(c["num_gens"]), "--rng", "device", "--graph", "auto", "--cache_device", "1", "--epochs", str(epochs),
        "--batch_size", str(c["scenes"]), "--synthetic_scenes", str(c["scenes"] * batches), "--synthetic_peds",
        str(c["peds"] or 0), "--val_every", "1000000", "--save_every", "1000000"])
    torch.manual_seed(145325)
    np.random.seed(435346)
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = construct_model(cfg)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    torch.cuda.manual_seed(1234)
    tr.zero_grads_in_step = True
    last = tr.train()
    ig = tr.iteration_graphs
    per_it = sorted(s / n for s, n in list(zip(tr.epoch_seconds, tr.epoch_iterations))[2:])
    ms = per_it[len(per_it) // 2] * 1e3
    b = c["scenes"] * (c["peds"] or 3)
    if not c["peds"]:  # ragged batches: the mean pedestrian count of the loader's batches
        from mggan.data_utils import synthetic

        b = float(np.mean([sum(synthetic.scene_sizes(c["scenes"], None, seed=i)) for i in range(batches)]))
    tr.dist.close()
    return {"workload": "train() on the synthetic loader: {} batches of {} scenes x {} peds per epoch, num_gens={}, {} epochs "
                        "(batches resident in HBM, --rng device, graph cache)".format(batches, c["scenes"], c["peds"],
                                                                                      c["num_gens"], epochs),
            "ms_per_step": round(ms, 4), "value": round(b / ms * 1e3, 2), "unit": "trajectories/s",
            "replayed_iterations": ig.replays if ig else 0, "eager_iterations": (ig.eager if ig else sum(tr.epoch_iterations)),
            "graphs": len(ig.entries) if ig else 0, "padded_iterations": ig.padded if ig else 0,
            "mean_b": round(float(b), 1),
            "ms_per_step_by_epoch": [round(s / n * 1e3, 4) for s, n in zip(tr.epoch_seconds, tr.epoch_iterations)],
            "last_losses": {k: round(v, 6) for k, v in sorted(last.items()) if "probs" not in k}}


CONFIGS = {  # BASELINE.json configs
    "c1": dict(scenes=32, peds=None, num_gens=1, name="BASELINE configs[0] shape: 32 ragged scenes (1-6 peds), num_gens=1"),
    "c2": dict(scenes=64, peds=20, num_gens=4, name="BASELINE configs[1] / per-GPU shard of configs[4]"),
    "c3": dict(scenes=256, peds=32, num_gens=8, name="BASELINE configs[2] / per-GPU shard of configs[3]"),
}


def _load_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return {}


def measure(tag, scenes, peds, num_gens, args, world, rank, dev, profile=True, graph=True, rng=None, transport=None):
    """Build a trainer for one workload, time exactly args.steps iterations between barriers, then (profile) time
    every C-ABI entry with HIP events over three eager iterations.  -> dict (rank 0 keeps it).
    transport (sharded runs): None = the default (peer-mapped all-reduce kernels inside the one graph when every rank
    could map its peers, else ncclAllReduce issued by the library inside the one graph, else RCCL between graph segments),
    "peer-mapped" = the kernels of csrc/comm.hip, "rccl-graph" = csrc/rccl.hip (RCCL inside the one graph),
    "rccl" / "rccl-segments" = torch.distributed (backend nccl == RCCL) collectives between graph segments."""
    import torch.distributed as dist
    from mggan.data_utils import synthetic
    from mggan.hip.lib import start_trace, stop_trace

    rng = rng or args.rng
    if transport is not None:  # read by mggan.devcomm.create / create_rccl when the trainer attaches its DistContext
        os.environ["MGGAN_DEVICE_COMM"] = "1" if transport == "peer-mapped" else "0"
        os.environ["MGGAN_RCCL_GRAPH"] = "0" if transport in ("rccl", "rccl-segments") else "1"
    tr = build_trainer(num_gens, rng, dev, seed=rank)
    tr.dist.equal_shards = True  # every rank holds the same number of scenes/pedestrians
    sizes = synthetic.scene_sizes(scenes, peds)
    batch = tr.to_device(synthetic.make_batch(sizes, seed=rank))
    b = batch["in_xy"].shape[1]
    batch["loss_mask"] = None  # synthetic data has no NaN ground truth: every pedestrian is valid
    tr.defer_metrics = True
    tr.zero_grads_in_step = True  # AdamW zeroes what it consumed: no separate memset per step
    metrics = defaultdict(list)
    use_graph = graph and rng == "device" and not args.no_graph
    sharded = tr.dist.enabled
    replay = None
    verbose = os.environ.get("MGGAN_BENCH_VERBOSE", "0") == "1"
    if verbose:
        print("[bench] {}: use_graph={} sharded={} transport={}".format(tag, use_graph, sharded, tr.dist.transport),
              file=sys.stderr, flush=True)
    if use_graph:
        try:
            replay = tr.capture_iteration(batch)
            if verbose:
                print("[bench] captured: {}".format(getattr(tr, "launch_mode", "?")), file=sys.stderr, flush=True)
        except Exception as exc:  # noqa: BLE001
            if not sharded:
                raise
            # every rank runs the same program on the same shapes, so they all end up here together
            print("[bench] rank {}: graph capture of the sharded iteration failed ({}: {}); launching eagerly".format(
                rank, type(exc).__name__, exc), file=sys.stderr)
            tr.dist.recorder = None
            use_graph = False
    if use_graph:
        def run_step(fetch):
            replay(metrics, fetch)
    else:
        def run_step(fetch):
            tr.train_iteration(batch, metrics)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run_step(i == args.warmup - 1)
        if verbose:
            torch.cuda.synchronize()
            print("[bench] warm-up step {} done".format(i), file=sys.stderr, flush=True)
    tr.flush_metrics()
    barrier()
    if sharded and tr.dist.stream_safe:
        # the in-graph all-reduce has never met this node before: the replicas must still hold identical weights
        # and no wait may have timed out; otherwise every rank falls back to the next transport together
        # (peer-mapped kernels -> RCCL inside the graph -> torch.distributed between graph segments)
        from mggan.parallel import replicas_in_sync

        was = tr.dist.transport
        ok = 1.0
        try:
            tr.dist.check(sync=True)
            ok = 1.0 if replicas_in_sync(tr.G, tr.D) else 0.0
        except Exception as exc:  # noqa: BLE001
            print("[bench] rank {}: {}".format(rank, exc), file=sys.stderr)
            ok = 0.0
        v = torch.tensor([ok], device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if float(v.item()) < 0.5:
            nxt = "rccl-graph" if was == "peer-mapped" else "rccl-segments"
            print("[bench] rank {}: {} all-reduce rejected; re-measuring with {}".format(rank, was, nxt), file=sys.stderr)
            tr.dist.close()
            del tr, replay
            torch.cuda.empty_cache()
            res = measure(tag, scenes, peds, num_gens, args, world, rank, dev, profile, graph, rng, transport=nxt)
            res["collective_note"] = "{} all-reduce rejected by the post-warm-up check; fell back to {}".format(
                was, res.get("collective"))
            return res
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i == args.steps - 1)  # logged losses are fetched once (one D2H) inside the timed region
    tr.flush_metrics()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launch = ("eager launches" if not use_graph else "hipGraph replay of the whole iteration" if not sharded
              else getattr(tr, "launch_mode", "hipGraph replay"))
    res = {"config": tag, "workload": "{} scenes x {} peds per GPU, num_gens={}, num_samples=20, D+G+PM steps ({})".format(
               scenes, peds if peds else "1-6 (ragged)", num_gens, CONFIGS.get(tag, {}).get("name", "custom shape")),
           "b_per_gpu": b, "ms_per_step": round(dt / args.steps * 1e3, 4),
           "value": round(world * b * args.steps / dt, 2), "unit": "trajectories/s", "rng": rng, "launch": launch,
           "bn_sync": tr.config.bn_sync,
           "collective": (None if not sharded else tr.dist.transport if (tr.dist.stream_safe or use_graph) else "rccl-eager"),
           "last_losses": {k: round(v[-1], 6) for k, v in sorted(metrics.items()) if "probs" not in k}}
    if sharded:
        tr.dist.check(sync=True)  # a timed-out wait inside a peer-mapped all-reduce would have flagged the arena
        # the exchange schedule of one iteration, counted on an eager iteration behind the timed region
        tr.dist.reset_count(log=True)
        tr.train_iteration(batch, defaultdict(list))
        log = list(tr.dist.collective_log)  # (the read-back of the logged losses behind it is not part of the iteration)
        tr.flush_metrics()
        torch.cuda.synchronize()
        res["exchanges_per_step"] = len([w for w in log if "second call" not in w])
        res["exchange_schedule"] = log
        tr.dist.reset_count()
    if not profile:
        tr.dist.close()
        del tr, replay
        torch.cuda.empty_cache()
        return res

    # ---- live per-entry timing for the roofline block ----
    # (1) stand-alone: HIP events around every C-ABI entry over three EAGER iterations (host-bound: kernels run alone);
    # (2) in the regime the headline runs in: a second captured iteration with device-clock marks (mggan_timestamp, on the
    #     entry's own stream, so they become graph nodes) around the entries with the largest stand-alone share; its
    #     replays give the duration of those kernels under the contention of graph replay
    start_trace()
    n_prof = 3
    for _ in range(n_prof):
        tr.train_iteration(batch, metrics)
    tr.flush_metrics()
    trace = stop_trace()
    rows = []
    operator_flops = 0.0
    for name, (calls, ms_list, arglist, launchlist) in trace.items():
        # one row per HIP kernel SYMBOL: an entry such as mggan_scene_attention_bwd launches a different instantiation per
        # channel count, and two instantiations are never summed into one row
        by_kernel = {}
        for a, ms, launches in zip(arglist, ms_list, launchlist):
            r = by_kernel.setdefault(kernel_of(name, launches), [0, 0.0, 0.0, 0.0])
            r[0] += 1
            r[1] += ms
            r[2] += flops_of(name, a)
            r[3] += bytes_of(name, a)
            operator_flops += operator_flops_of(name, a) / n_prof
        for sym, (c, ms, fl, by) in by_kernel.items():
            rows.append((ms / n_prof, name, c / n_prof, fl / n_prof, sym, by / n_prof))
    # a symbol launched from several entries (gemm_kernel<...>, dheads_bwd_kernel) is ONE row, named after the entry that
    # holds most of its time
    merged = {}
    for ms, name, c, fl, sym, by in sorted(rows, reverse=True):
        m = merged.get(sym)
        merged[sym] = (ms, name, c, fl, sym, by) if m is None else (m[0] + ms, m[1], m[2] + c, m[3] + fl, sym, m[5] + by)
    rows = list(merged.values())
    replay_ms = {}
    if use_graph and not sharded:
        replay_ms = marked_replay(tr, batch, [r[1] for r in sorted(rows, reverse=True)[:10]])
    rows.sort(reverse=True)
    gpu_ms = sum(r[0] for r in rows)
    executed_flops = sum(r[3] for r in rows)
    total_flops = operator_flops  # the reference operators' FLOPs; executed: fewer where a part is shared
    # HBM bytes per launch and MFMA-pipe utilisation from the committed rocprofv3 --pmc passes
    traffic_tab = _load_json("hbm_traffic_{}.json".format(tag)) or (_load_json("hbm_traffic.json") if tag == "c2" else {})
    mfma_tab = _load_json("mfma_util_{}.json".format(tag))

    def family(tab, symbol, key, how):
        """Counter value of a kernel, or of the launches of a family (wgrad_stream_kernel -> <0>, <2>) taken together."""
        if symbol in tab:
            return tab[symbol].get(key)
        vals = [e.get(key) for k, e in tab.items() if k.startswith(symbol + "<") and e.get(key) is not None]
        return how(vals) if vals else None

    def roof(row):
        ms_alone, name, calls, fl, symbol, by = row
        ms = replay_ms.get(symbol, ms_alone)  # per iteration, all launches of this kernel
        per_launch_s = ms / max(calls, 1) * 1e-3
        tf = fl / max(calls, 1) / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
        gbs = by / max(calls, 1) / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        common = {"kernel": symbol, "entry": name, "traffic": family(traffic_tab, symbol, "bytes_per_launch", sum),
                  "mfma_util": family(mfma_tab, symbol, "mfma_util", max),
                  "launches_per_step": round(calls, 2), "avg_launch_ms": round(per_launch_s * 1e3, 4),
                  "timed": "graph replay (device-clock marks)" if symbol in replay_ms else "eager (HIP events)",
                  "standalone_ms": round(ms_alone / max(calls, 1), 4),
                  "tflops": round(tf, 3), "algorithmic_gbs": round(gbs, 1),
                  # BOTH fractions, whatever `bound` says: where they are close the single label hides the other roof
                  "frac_mfma": round(tf / F32_PEAK_TFLOPS, 5), "frac_hbm": round(gbs / HBM_PEAK_GBS, 5)}
        tb = common["traffic"]
        if tb and per_launch_s > 0:  # ... and the fraction of the HBM roof by MEASURED bytes (re-reads included)
            common["frac_hbm_measured"] = round(tb / per_launch_s / 1e9 / HBM_PEAK_GBS, 5)
        # the binding roof is the one the kernel sits closer to
        if gbs / HBM_PEAK_GBS > tf / F32_PEAK_TFLOPS:
            return dict(common, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(gbs / HBM_PEAK_GBS, 5))
        return dict(common, bound="mfma", achieved=round(tf, 3), peak=F32_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(tf / F32_PEAK_TFLOPS, 5))

    hbm_step = None
    if traffic_tab:  # measured bytes per launch (committed PMC table) x this run's launches per iteration, kernel by kernel
        tot = 0.0
        for _, _, calls, _, sym, _ in rows:
            e = traffic_tab.get(sym)
            if isinstance(e, dict) and e.get("bytes_per_launch") is not None:
                tot += float(e["bytes_per_launch"]) * calls
        hbm_step = round(tot) if tot > 0 else None
    # the dominant kernel = the one with the largest share of the step's GPU WORK: its launches' time when each runs alone
    # on the GPU (rows[.][0]).  Its duration for `achieved` is the one inside graph replay (roof()); ranking by that duration
    # instead made the block flip from run to run between the rollout adjoint and whichever small CNN kernel the replay had
    # stretched most by running others beside it (conv1_pool_kernel<8>: 35 us alone, 60-100 us in the graph)
    rows.sort(key=lambda r: r[0], reverse=True)
    roofline = roof(rows[0])
    roofline["note"] = ("f32 (exact) -- mfma: peak is the dense f32 vector/MFMA rate, achieved = algorithmic FLOPs per launch "
                        "(SURVEY App. D shapes) / average duration of a launch of this kernel INSIDE graph replay (device-clock "
                        "marks around the entry in a second captured iteration; standalone_ms = the same kernel alone on the "
                        "GPU, HIP events over eager iterations); hbm: achieved = "
                        "algorithmic bytes per launch (every operand once, DESIGN.md section 7) / the same duration, peak 8 TB/s; "
                        "the bound reported is the roof the kernel sits closer to; traffic = HBM "
                        "bytes per launch, mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs), both "
                        "from the rocprofv3 --pmc passes committed under profiles/")
    res.update({
        "roofline": roofline, "roofline_top_kernels": [roof(r) for r in rows[:8]],
        "iteration_flops_algorithmic_g": round(total_flops / 1e9, 2),
        "iteration_flops_executed_g": round(executed_flops / 1e9, 2),
        "iteration_tflops": round(total_flops / (dt / args.steps) / 1e12, 3),
        "iteration_frac_of_f32_peak": round(total_flops / (dt / args.steps) / 1e12 / F32_PEAK_TFLOPS, 4),
        # HBM bytes of one iteration: the sum over the committed per-kernel PMC table (profiles/hbm_traffic_<cfg>.json:
        # bytes per launch x launches per iteration), and what that is of the 8 TB/s roof at this run's iteration time
        "hbm_bytes_per_step": hbm_step,
        "iteration_frac_of_hbm_peak": round(hbm_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if hbm_step else None,
        "gpu_ms_per_step_sum_of_entries": round(gpu_ms, 3),
        "launches_per_step": round(sum(r[2] for r in rows), 1),
        "breakdown": [{"entry": n, "kernel": sym, "ms_per_step": round(ms, 4), "calls": round(c, 2), "gflop": round(fl / 1e9, 3),
                       "algorithmic_mb": round(by / 1e6, 1)} for ms, n, c, fl, sym, by in rows[:60]]})
    tr.dist.close()
    del tr, replay
    torch.cuda.empty_cache()
    return res


LINE_LIMIT = 4096  # the driver keeps the last 8 KB of stdout: the one JSON line must fit with room to spare

ROOFLINE_KEYS = ("kernel", "entry", "bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_util", "avg_launch_ms",
                 "frac_mfma", "frac_hbm", "frac_hbm_measured")


def _clean(x):
    """NaN / inf -> None (the line is strict JSON), numpy scalars -> Python numbers."""
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        x = x.item()
    if isinstance(x, float) and not np.isfinite(x):
        return None
    return x


def compact_line(full):
    """The ONE line the driver parses: the contract's keys, the roofline block of the dominant kernel, the CPU baseline and
    one short entry per measured workload.  Everything else (per-kernel tables, breakdowns, floors, train() leg, notes)
    goes to bench_detail.json."""
    full = _clean(full)
    cfg = full.get("config", {})
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "b_per_gpu", "parallelism", "launch", "collective") if k in cfg}
    if full.get("roofline"):
        line["roofline"] = {k: full["roofline"].get(k) for k in ROOFLINE_KEYS}
    if full.get("cpu_baseline"):
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "mode", "sample", "spread") if k in cb}
        line["cpu_baseline"]["value"] = round(cb["value"], 2)
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:200]
    confs = []
    for r in full.get("configs", []):
        e = {"workload": r.get("config"), "b_per_gpu": r.get("b_per_gpu"), "ms_per_step": r.get("ms_per_step"),
             "value": r.get("value")}
        if r.get("roofline"):
            e.update(kernel=r["roofline"].get("kernel"), frac=r["roofline"].get("frac"), bound=r["roofline"].get("bound"))
        if r.get("iteration_frac_of_f32_peak") is not None:
            e["iteration_frac_of_f32_peak"] = r["iteration_frac_of_f32_peak"]
        if r.get("exchanges_per_step") is not None:  # sharded runs: transport and exchanges per iteration
            e.update(collective=r.get("collective"), exchanges_per_step=r["exchanges_per_step"])
        confs.append(e)
    if confs:
        line["configs"] = confs
    for k in ("iteration_frac_of_f32_peak", "hbm_bytes_per_step", "iteration_frac_of_hbm_peak", "launches_per_step"):
        if full.get(k) is not None:
            line[k] = full[k]
    if full.get("collective_transports"):
        line["collective_transports"] = [
            {"config": t.get("config"), **{k: {"ms_per_step": v.get("ms_per_step"), "value": v.get("value")}
                                             for k, v in t.items() if isinstance(v, dict)}}
            for t in full["collective_transports"]]
    line["detail"] = "bench_detail.json"
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # never expected: drop the optional blocks rather than print an unparseable line
        for k in ("collective_transports", "configs"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(full):
    """Write the full record to bench_detail.json (repo root, and gpurun_out/ so that it travels back from a GPU box) and
    return the compact line."""
    text = compact_line(full)
    blob = json.dumps(_clean(full), allow_nan=False, indent=1)
    for d in () if os.environ.get("MGGAN_BENCH_NO_DETAIL") == "1" else (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as fh:
                fh.write(blob)
        except OSError:
            pass
    return text


def self_launch(n):
    """Re-exec this command line as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py ...`
    (rendezvous on 127.0.0.1, a free port).  -> exit code of the job."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    # (MGGAN_DIST_BACKEND=gloo: ranks may share a GPU -- the functional check of this path on a one-GPU box)
    if have < n and not (have >= 1 and os.environ.get("MGGAN_DIST_BACKEND") == "gloo"):
        print("[bench] --gpus {} asked for, {} HIP device(s) visible".format(n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", help="headline workload: c1 | c2 | c3 (= the per-GPU shard of configs[3]; "
                    "alias c4) | custom (uses --scenes/--peds/--num_gens)")
    ap.add_argument("--also", default="c3", help="comma list of further configs measured into the `configs` list ('' = none)")
    ap.add_argument("--scenes", type=int, default=None)
    ap.add_argument("--peds", type=int, default=None)
    ap.add_argument("--num_gens", type=int, default=None)
    ap.add_argument("--rng", choices=["host", "device"], default="device")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-floor", action="store_true", help="skip the C1-shaped eager / host-RNG / graph floor timings")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-entry timing behind the timed region (no roofline block)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the forced-one-rank sharded-mode legs of the N=1 run")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--transport-ab", action="store_true",
                    help="N>1: repeat both workloads on the OTHER collective transport (RCCL between graph segments when the "
                         "default is the peer-mapped kernels) into `collective_transports`.  Opt-in: the N>1 line the "
                         "driver parses must not depend on a second trainer and a second communicator coming up")
    ap.add_argument("--no-transport-ab", action="store_true", help=argparse.SUPPRESS)  # (round-4 command lines)
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_baseline_worker(args.cpu_worker)
    if args.config == "c4":
        args.config = "c3"
    if args.scenes is not None:  # explicit shape (round-1 command lines keep working)
        CONFIGS["custom"] = dict(scenes=args.scenes, peds=args.peds, num_gens=args.num_gens or 4, name="custom shape")
        for k, v in CONFIGS.items():
            if k != "custom" and (v["scenes"], v["peds"], v["num_gens"]) == (args.scenes, args.peds, args.num_gens or 4):
                args.config = k
                break
        else:
            args.config = "custom"
        args.also = ""

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("MGGAN_FORCE_DIST", "0") != "1":
        # `python bench.py --gpus N` on its own: start N ranks (one per GPU) of this very command under
        # torch.distributed.run and hand its output (the one JSON line of rank 0) through
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("MGGAN_FORCE_DIST", "0") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # MGGAN_DIST_BACKEND=gloo lets two ranks share ONE GPU (RCCL refuses that): a functional check of the
        # sharded path on a single-GPU box, never a measurement
        backend = os.environ.get("MGGAN_DIST_BACKEND", "nccl")
        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # host-side bookkeeping is a handful of tiny torch CPU ops: a large OpenMP team only adds fork/join latency
    torch.set_num_threads(min(4, os.cpu_count() or 1))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from mggan.data_utils import synthetic
    from mggan.hip import lib as hiplib_mod  # noqa: F401

    head_cfg = CONFIGS[args.config]
    prof = not args.no_profile
    head = measure(args.config, head_cfg["scenes"], head_cfg["peds"], head_cfg["num_gens"], args, world, rank, dev, profile=prof)
    others = []
    for tag in [t for t in args.also.split(",") if t and t != args.config]:
        c = CONFIGS[tag]
        others.append(measure(tag, c["scenes"], c["peds"], c["num_gens"], args, world, rank, dev, profile=prof))
    transports = None
    if world > 1 and args.transport_ab and not args.no_transport_ab:
        # the same workloads on the other transport (north_star names RCCL; the default keeps the iteration ONE graph
        # with the exchange points as peer-mapped kernels): both numbers in the line, same timing protocol
        transports = []
        for tag in [args.config] + [t for t in args.also.split(",") if t and t != args.config]:
            c = CONFIGS[tag]
            base = head if tag == args.config else [o for o in others if o["config"] == tag][0]
            row = {"config": tag, base["collective"]: {"ms_per_step": base["ms_per_step"], "value": base["value"],
                                                       "launch": base["launch"]}}
            for other in ("peer-mapped", "rccl-graph", "rccl-segments"):
                if other in row:
                    continue
                alt = measure(tag, c["scenes"], c["peds"], c["num_gens"], args, world, rank, dev, profile=False,
                              transport=other)
                row.setdefault(alt["collective"], {"ms_per_step": alt["ms_per_step"], "value": alt["value"],
                                                   "launch": alt["launch"]})
            transports.append(row)
        os.environ.pop("MGGAN_DEVICE_COMM", None)
        os.environ.pop("MGGAN_RCCL_GRAPH", None)

    out = None
    if rank == 0:
        keep = ("config", "workload", "b_per_gpu", "ms_per_step", "value", "unit", "launch", "collective", "roofline",
                "roofline_top_kernels", "iteration_flops_algorithmic_g", "iteration_flops_executed_g", "iteration_tflops",
                "iteration_frac_of_f32_peak", "hbm_bytes_per_step", "iteration_frac_of_hbm_peak", "launches_per_step", "breakdown",
                "exchanges_per_step", "exchange_schedule",
                "collective_note")
        out = {
            "metric": "train-step trajectories/sec", "value": head["value"], "unit": "trajectories/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["workload"], "b_per_gpu": head["b_per_gpu"], "parallelism": "dp{}".format(world),
                       "rng": head["rng"], "bn_sync": head["bn_sync"], "launch": head["launch"],
                       "collective": head.get("collective"), "last_losses": head["last_losses"]},
            "roofline": head.get("roofline"), "roofline_top_kernels": head.get("roofline_top_kernels"),
            "iteration_flops_algorithmic_g": head.get("iteration_flops_algorithmic_g"),
            "iteration_flops_executed_g": head.get("iteration_flops_executed_g"),
            "iteration_tflops": head.get("iteration_tflops"),
            "iteration_frac_of_f32_peak": head.get("iteration_frac_of_f32_peak"),
            "hbm_bytes_per_step": head.get("hbm_bytes_per_step"),
            "iteration_frac_of_hbm_peak": head.get("iteration_frac_of_hbm_peak"),
            "gpu_ms_per_step_sum_of_entries": head.get("gpu_ms_per_step_sum_of_entries"),
            "launches_per_step": head.get("launches_per_step"), "breakdown": head.get("breakdown"),
            # every measured workload, the headline first (same timing protocol: warmup, barrier, K steps, barrier)
            "configs": [{k: r[k] for k in keep if k in r} for r in [head] + others],
        }
        if transports is not None:
            out["collective_transports"] = transports
        if world == 1 and os.environ.get("MGGAN_FORCE_DIST", "0") != "1" and not args.no_floor and not args.no_sharded_leg:
            # (detail file) the sharded launch mode on one rank, both in-graph transports: the part of the N > 1 points that
            # is measurable here -- exchanges per step, the schedule, the overhead over this run's single-GPU graphs
            out["sharded_mode_one_rank"] = sharded_one_rank_leg(args, {r["config"]: r["ms_per_step"] for r in [head] + others})
    if world == 1 and os.environ.get("MGGAN_FORCE_DIST", "0") != "1" and not args.no_floor:
        # what a user of train() gets on ragged ETH-shaped batches (configs[0] shape): the graph floor, eager launches
        # with the device RNG, and eager launches with the seed-comparable host RNG (one D2H + host multinomial per G call)
        c1 = CONFIGS["c1"]
        floor_args = argparse.Namespace(**vars(args))
        floor_args.steps, floor_args.warmup = max(args.steps, 20), max(args.warmup, 5)
        g1 = measure("c1", c1["scenes"], c1["peds"], c1["num_gens"], floor_args, world, rank, dev, profile=False)
        e1 = measure("c1", c1["scenes"], c1["peds"], c1["num_gens"], floor_args, world, rank, dev, profile=False, graph=False)
        h1 = measure("c1", c1["scenes"], c1["peds"], c1["num_gens"], floor_args, world, rank, dev, profile=False, graph=False,
                     rng="host")
        out["c1_shaped"] = {"workload": g1["workload"], "b": g1["b_per_gpu"], "graph_ms_per_step": g1["ms_per_step"],
                            "graph_value": g1["value"], "eager_ms_per_step": e1["ms_per_step"], "eager_value": e1["value"],
                            "host_rng_ms_per_step": h1["ms_per_step"], "host_rng_value": h1["value"],
                            "note": "train() replays a captured graph per batch shape with --rng device (`train_loop` below); "
                                    "a shape's first batch, shapes beyond its cache and --rng host (the seed-comparable "
                                    "mode) launch eagerly"}
        e2 = measure(args.config, head_cfg["scenes"], head_cfg["peds"], head_cfg["num_gens"], floor_args, world, rank, dev,
                     profile=False, graph=False)
        out["eager_ms_per_step"] = e2["ms_per_step"]
        if head_cfg["peds"]:
            tl = train_loop_leg(args.config, args, dev)
            tl["vs_graph_headline"] = round(tl["value"] / out["value"], 3)
            out["train_loop"] = tl
        if args.config == "c2":
            try:
                out["train_loop_disk"] = train_loop_disk_leg(args, dev, head["ms_per_step"])
            except Exception as exc:  # noqa: BLE001  (a detail leg must not cost the line)
                out["train_loop_disk"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}
        # train() on RAGGED batches of the configs[0] shape (32 scenes of 1-6 pedestrians, a new tuple of sizes every batch,
        # like the reference loader's): padded to shape buckets, one captured graph per bucket replays them all
        tr_ = train_loop_leg("c1", args, dev)
        tr_["vs_c1_graph_floor"] = round(tr_["ms_per_step"] / g1["ms_per_step"], 3)
        tr_["vs_c1_eager"] = round(tr_["ms_per_step"] / e1["ms_per_step"], 3)
        out["train_loop_ragged"] = tr_
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sizes = synthetic.scene_sizes(head_cfg["scenes"], head_cfg["peds"])
        out["cpu_baseline"] = cpu_baseline(sizes, head_cfg["num_gens"], args.cpu_iters, "block", "the headline workload")
        # (detail file only: a ratio against the PORT says nothing about kernel quality -- the roofline fraction does)
        out["gpu_over_cpu_port"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        # the reference's own operator sequence (dense all-pairs social features, per-pedestrian select loop) is cubic in
        # the batch: at the full configs[0] shape (b~100) one iteration takes minutes, so the bounded sample is its first
        # 8 scenes; the rate at the full shape is lower still (0.30 trajectories/s at b=96)
        c1s = synthetic.scene_sizes(CONFIGS["c1"]["scenes"], None)[:8]
        fb = cpu_baseline(c1s, 1, max(3, args.cpu_iters), "faithful", "the first 8 scenes of the configs[0] shape")
        out["cpu_baseline_faithful_c1"] = fb
    if world > 1 or os.environ.get("MGGAN_FORCE_DIST", "0") == "1":
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # the JSON line is the LAST thing on stdout: librccl prints a version banner through C stdio, which sits in
        # the C buffer until it is flushed (at exit, i.e. after anything Python printed earlier)
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(emit(out), flush=True)


if __name__ == "__main__":
    main()
