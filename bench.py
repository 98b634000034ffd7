#!/usr/bin/env python
"""bench.py -- MG-GAN train-step trajectories/sec on MI355X (BASELINE.json metric).

One "step" = one full iteration of the reference loop body (abstract_train.py:114-168):
discriminator step + generator step + PM-network step, each with forward, backward, gradient
clipping and an AdamW update, on a synthetic batch already resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- 64 scenes x 20 pedestrians, num_gens=4, K=20 samples.
N>1 (launched by torch.distributed.run, one rank per GPU): every rank owns 64 scenes of a
N*64-scene global batch (weak scaling); gradients / BatchNorm statistics / generator counts are
all-reduced over RCCL.  value = N * b_local * steps / max-over-ranks(time).

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


def kernel_of(name, a):
    """C-ABI entry -> the HIP kernel that does its work, named as tools/hbm_traffic.py and the rocprofv3
    summaries under profiles/ name it."""
    fixed = {
        "mggan_decoder_rollout_bwd_fused": "decoder_bwd_mfma_kernel",
        "mggan_decoder_rollout_fwd": "decoder_fwd_mfma_kernel",
        "mggan_wgrad_multi": "gemm_multi_kernel<true,true>",
        "mggan_wgrad": "gemm_kernel<true,true,false>",
        "mggan_linear_fwd": "gemm_kernel<false,false,false>",
        "mggan_linear_bwd_data": "gemm_kernel<false,true,false>",
        "mggan_mlp_chain": "mlp_chain_kernel",
    }
    if name in fixed:
        return fixed[name]
    if name == "mggan_conv1_fwd":
        return "conv1_fwd_mfma_kernel<{}>".format(a[2])
    if name == "mggan_conv2_bwd":
        return "conv2_bwd_mfma_kernel" if a[2] == 16 else "conv2_bwd_kernel<8>"
    if name in ("mggan_conv1_bwd", "mggan_conv2_fwd"):
        return "{}_kernel<{}>".format(name[len("mggan_"):], a[2])
    return name


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
        return float(R) * (2 * (EIN + Z) * H + T * (2 * 2 * 16 + 2 * (16 + H) * 4 * H + 2 * (2 * H * (H // 2) + (H // 2) * 2)))
    if name == "mggan_decoder_rollout_bwd_fused":  # BPTT data path + the fused per-generator weight gradients
        T, H, EIN, R = a[2], a[3], a[4], a[22]
        data = T * 2 * (4 * H * H + 2 * 4 * H + H * (H // 2) + (H // 2) * 2) + 2 * (EIN * H + H * (H // 2))
        wgrads = T * 2 * (4 * H * (H + 3) + (H // 2) * H + 2 * (H // 2))
        return float(R) * (data + wgrads)
    if name == "mggan_conv1_fwd":
        return float(a[1]) * 2 * 33 * 33 * a[2] * 36
    if name == "mggan_conv2_fwd":
        return float(a[1]) * 2 * 256 * a[2] * a[2] * 9
    if name == "mggan_conv2_bwd":
        return float(a[1]) * 2 * 2 * 256 * a[2] * a[2] * 9
    if name == "mggan_conv1_bwd":
        return float(a[1]) * 2 * 33 * 33 * a[2] * 36
    if name == "mggan_scene_attention_fwd":
        return float(a[1]) * 64 * 2 * (a[2] * 32 * 2)
    if name == "mggan_scene_attention_bwd":
        return float(a[1]) * 64 * 2 * (a[2] * 32 * 2) * 3
    if name == "mggan_social_pairs_fwd":
        return float(a[0]) * 2 * (96 + 2048 + 64)
    if name == "mggan_social_pairs_bwd":
        return float(a[0]) * 2 * (64 + 2048)
    if name == "mggan_social_attention_fwd":
        return float(a[2]) * 2 * (96 + 2048 + 64 + a[3])
    if name == "mggan_social_attention_bwd":
        return float(a[2]) * 2 * (64 + 2048 + 3 * a[4] + 65)
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


def cpu_baseline(sizes, num_gens, iters):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mggan_oracle as O
    from mggan.data_utils import synthetic

    threads = min(16, os.cpu_count() or 1)  # more threads are slower for these small operators
    torch.set_num_threads(threads)
    torch.manual_seed(145325)
    np.random.seed(435346)
    G, D = O.construct_oracle(num_gens)
    tr = O.OracleTrainer(G, D, mode="block")
    batch = synthetic.make_batch(sizes, seed=0)
    m = defaultdict(list)
    tr.iteration(batch, m)  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.iteration(batch, m)
    dt = (time.perf_counter() - t0) / iters
    b = batch["in_xy"].shape[1]
    return {"value": b / dt, "unit": "trajectories/s", "cores": threads, "kind": "port",
            "sample": "{} full iterations (D+G+PM steps) of the same {}-scene x {}-ped, num_gens={} workload on the CPU "
                      "oracle (oracle/mggan_oracle.py, block-diagonal mode, torch CPU {} threads, nproc={}); "
                      "{:.2f} s/iteration".format(iters, len(sizes), sizes[0], num_gens, threads, os.cpu_count(), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenes", type=int, default=64)
    ap.add_argument("--peds", type=int, default=20)
    ap.add_argument("--num_gens", type=int, default=4)
    ap.add_argument("--rng", choices=["host", "device"], default="device")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=2)
    args = ap.parse_args()

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
    from mggan.hip.lib import start_trace, stop_trace

    tr = build_trainer(args.num_gens, args.rng, dev, seed=rank)
    tr.dist.equal_shards = True  # every rank holds the same number of scenes/pedestrians
    sizes = synthetic.scene_sizes(args.scenes, args.peds)
    batch = tr.to_device(synthetic.make_batch(sizes, seed=rank))
    b = batch["in_xy"].shape[1]
    batch["loss_mask"] = None  # synthetic data has no NaN ground truth: every pedestrian is valid
    tr.defer_metrics = True
    tr.zero_grads_in_step = True  # AdamW zeroes what it consumed: no separate memset per step
    metrics = defaultdict(list)
    use_graph = args.rng == "device" and not args.no_graph
    replay = None
    sharded = tr.dist.enabled
    if use_graph and sharded and os.environ.get("MGGAN_GRAPH_COLLECTIVES", "0") != "0":
        # opt-in (MGGAN_GRAPH_COLLECTIVES=1|auto, experimental): the RCCL collectives captured INSIDE one graph.  The
        # attempt is checked (two replays must leave every rank with identical weights) and bounded in time; if it fails,
        # a fresh trainer is built and the capture is cut into segments around eager collectives instead.
        import threading

        from mggan.parallel import replicas_in_sync

        import torch.distributed as dist

        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)  # first collective of the process: RCCL loads its kernels here, however long that takes
        torch.cuda.synchronize()
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("MGGAN_GRAPH_TRIAL_TIMEOUT", "300"))):
                print("[bench] rank {}: graph with captured collectives did not finish in time; rerun with "
                      "MGGAN_GRAPH_COLLECTIVES=0".format(rank), file=sys.stderr, flush=True)
                os._exit(17)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            replay = tr.capture_iteration(batch)
            if tr.graph_collectives:
                for _ in range(2):
                    replay(metrics, False)
                torch.cuda.synchronize()
                if not replicas_in_sync(tr.G, tr.D):
                    raise RuntimeError("ranks diverged after replaying the captured collectives")
        except Exception as exc:  # noqa: BLE001
            print("[bench] rank {}: one-graph capture with RCCL inside failed ({}: {}); cutting the capture into "
                  "segments".format(rank, type(exc).__name__, exc), file=sys.stderr)
            replay = None
            os.environ["MGGAN_GRAPH_COLLECTIVES"] = "0"
            torch.cuda.synchronize()
            tr = build_trainer(args.num_gens, args.rng, dev, seed=rank)
            tr.dist.equal_shards = True
            tr.defer_metrics = True
            tr.zero_grads_in_step = True
        finally:
            done.set()
    if use_graph and replay is None:
        try:
            replay = tr.capture_iteration(batch)
        except Exception as exc:  # noqa: BLE001
            if not sharded:
                raise
            # every rank runs the same program on the same shapes, so they all end up here together
            print("[bench] rank {}: graph-segment capture failed ({}: {}); launching eagerly".format(
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
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run_step(i == args.warmup - 1)
    tr.flush_metrics()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i == args.steps - 1)  # logged losses are fetched once (one D2H) inside the timed region
    tr.flush_metrics()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- live per-entry timing (HIP events on the launch stream) for the roofline block ----
    start_trace()
    n_prof = 3
    for _ in range(n_prof):
        tr.train_iteration(batch, metrics)
    tr.flush_metrics()
    trace = stop_trace()
    rows = []
    for name, (calls, ms_list, arglist) in trace.items():
        # one row per HIP kernel: an entry such as mggan_conv1_bwd launches a different template per channel count
        by_kernel = {}
        for a, ms in zip(arglist, ms_list):
            r = by_kernel.setdefault(kernel_of(name, a), [0, 0.0, 0.0])
            r[0] += 1
            r[1] += ms
            r[2] += flops_of(name, a)
        for sym, (c, ms, fl) in by_kernel.items():
            rows.append((ms / n_prof, name, c // n_prof, fl / n_prof, sym))
    rows.sort(reverse=True)
    gpu_ms = sum(r[0] for r in rows)
    total_flops = sum(r[3] for r in rows)
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    traffic_tab = {}
    if os.path.exists(tpath):  # HBM bytes per launch from the committed rocprofv3 --pmc passes
        with open(tpath) as fh:
            traffic_tab = json.load(fh)

    def roof(row):
        ms, name, calls, fl, symbol = row
        per_launch_s = ms / max(calls, 1) * 1e-3
        achieved = fl / max(calls, 1) / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
        return {"bound": "mfma", "kernel": symbol, "entry": name, "achieved": round(achieved, 3),
                "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / F32_PEAK_TFLOPS, 5),
                "traffic": traffic_tab.get(symbol, {}).get("bytes_per_launch"), "launches_per_step": calls,
                "avg_launch_ms": round(per_launch_s * 1e3, 4)}

    # the dominant kernel = the one with the largest share of the step's GPU time (summed over its launches)
    roofline = roof(rows[0])
    roofline["note"] = ("f32 (exact) -- peak is the dense f32 vector/MFMA rate; achieved = algorithmic FLOPs per launch "
                        "(SURVEY App. D shapes) / average HIP-event duration of a launch of this kernel; traffic = HBM "
                        "bytes per launch from the rocprofv3 --pmc passes committed under profiles/")
    roofline_top = [roof(r) for r in rows[:6]]
    breakdown = [{"entry": n, "ms_per_step": round(ms, 4), "calls": c, "gflop": round(fl / 1e9, 3)}
                 for ms, n, c, fl, _ in rows[:12]]

    if rank == 0:
        out = {
            "metric": "train-step trajectories/sec", "value": round(world * b * args.steps / dt, 2),
            "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "{} scenes x {} peds per GPU, num_gens={}, num_samples=20, D+G+PM steps ({})".format(
                           args.scenes, args.peds, args.num_gens,
                           {(64, 20, 4): "BASELINE configs[1]", (256, 32, 8): "BASELINE configs[2]"}.get(
                               (args.scenes, args.peds, args.num_gens), "custom shape")),
                       "b_per_gpu": b, "parallelism": "dp{}".format(world), "rng": args.rng,
                       "bn_sync": tr.config.bn_sync,
                       "launch": ("eager" if not use_graph else "hipGraph replay of the whole iteration" if not sharded
                                  else "hipGraph replay of the whole iteration, RCCL collectives captured inside it"
                                  if getattr(tr, "graph_collectives", False)
                                  else "{} hipGraph segments per iteration, RCCL collectives between them".format(
                                      replay.graph.n_graphs)),
                       "last_losses": {k: round(v[-1], 6) for k, v in sorted(metrics.items()) if "probs" not in k}},
            "roofline": roofline,
            "roofline_top_kernels": roofline_top,
            "iteration_flops_algorithmic_g": round(total_flops / 1e9, 2),
            "iteration_tflops": round(total_flops / (dt / args.steps) / 1e12, 3),
            "gpu_ms_per_step_sum_of_entries": round(gpu_ms, 3),
            "breakdown": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sizes, args.num_gens, args.cpu_iters)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    else:
        out = None
    if world > 1 or sharded:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # the JSON line is the LAST thing on stdout: librccl prints a version banner through C stdio, which sits in
        # the C buffer until it is flushed (at exit, i.e. after anything Python printed earlier)
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
