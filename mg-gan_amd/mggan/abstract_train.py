"""Training driver with the reference's surface (/root/reference/mggan/abstract_train.py:25-296):
MultiGeneratorGAN.{train, save, load, load_from_path} and the optimizer / schedule setup.
One iteration = discriminator step -> generator step -> PM-network step (:136-159)."""
import abc
import ctypes
import math
import os
import sys
import time
from argparse import Namespace
from collections import defaultdict
from pathlib import Path
from statistics import mean

import numpy as np
import torch

from mggan.data_utils.data_loaders import get_dataloader
from mggan.logging import Experiment
from mggan.model.config import get_parser
from mggan.optim import FlatAdamW, CosineAnnealingLR
from mggan.parallel import DistContext
from mggan.rng import HostRNG, DeviceRNG

# Set seeds for reproducibility (abstract_train.py:14-15)
torch.random.manual_seed(145325)
np.random.seed(435346)


class MultiGeneratorGAN(abc.ABC):
    def __init__(self, generator, discriminator, config, writer):
        self.writer = writer
        self.config = config
        # `--gpus` is a device-id string ("0" by default); meta_tags.csv read-back may hand it over as int 0 -- only an
        # explicitly empty / None / False value asks for the CPU path, which this build does not have
        if config.gpus is None or config.gpus is False or (isinstance(config.gpus, str) and config.gpus.strip() == ""):
            raise RuntimeError("the MI355X build has no CPU compute path: pass --gpus 0 (use the oracle for CPU runs)")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible; the hot path runs only on the GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.D = discriminator.to(self.device).flatten_parameters_()
        self.G = generator.to(self.device).flatten_parameters_()
        self.l2_weight = self.config.l2_loss_weight
        self.gan_type = self.config.gan_type
        if self.config.gan_obj not in ("NS", "MM", "LS"):
            raise ValueError("Objective not supported on the HIP path ('NS' (default), 'MM', 'LS'; 'W' needs the "
                             "unbounded discriminator output)")

        self.log_dir = Path(self.writer.get_data_path(self.writer.name, self.writer.version))
        self.model_save_dir = self.log_dir / "checkpoints"
        self.model_save_dir.mkdir(exist_ok=True, parents=True)

        self.optimizerD = FlatAdamW(self.D, lr=self.config.d_lr, betas=(config.beta1, 0.999))
        self.optimizerG = FlatAdamW(self.G, lr=self.config.g_lr, betas=(config.beta1, 0.999))
        self.lr_schedulerD = CosineAnnealingLR(self.optimizerD, config.epochs, eta_min=0)
        self.lr_schedulerG = CosineAnnealingLR(self.optimizerG, config.epochs, eta_min=0)
        self.epoch = 0
        self.total_iterations = 0  # abstract_train.py:104 (gates the discriminator step with --num_gen_steps)

        self.rng = DeviceRNG() if getattr(config, "rng", "device") == "device" else HostRNG()
        self.G.rng = self.rng
        self.dist = DistContext()
        self.dist.attach(self.G, self.D, bn_sync=getattr(config, "bn_sync", "global"))
        if self.dist.enabled and not self.dist.stream_safe and os.environ.get("MGGAN_BRANCH_SHARDED", "0") != "1":
            # sharded runs are replayed as ~19 short graph segments (one per collective); forks that have to be
            # joined at every cut measured slower than one stream (3.49 vs 3.24 ms per iteration with dummy
            # collectives on one MI355X), so the branch streams stay off unless asked for
            from mggan.hip import functions as HF

            HF.enable_branches(False)

    def to_device(self, batch):
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def train_iteration(self, batch, metrics):
        """Loop body of abstract_train.py:114-168 for one collated batch (already on the device)."""
        from mggan.hip import functions as HF

        pad = batch.get("pad")  # a bucket's static scene tables (IterationGraphs): the batch ends in phantom pedestrians
        pad_was = HF.set_pad_dims(pad.dims, pad.b) if pad is not None else HF.set_pad_dims(None)
        try:
            self._train_iteration(batch, metrics)
        finally:
            HF.set_pad_dims(*pad_was)

    def _train_iteration(self, batch, metrics):
        from mggan.hip import functions as HF

        try:
            self._train_iteration_body(batch, metrics, HF)
        finally:
            HF.force_masked(False)

    def _train_iteration_body(self, batch, metrics, HF):
        in_xy, in_dxdy = batch["in_xy"], batch["in_dxdy"]
        b = in_xy.size(1)
        sub_batches = batch["seq_start_end"] if "seq_start_end" in batch else list(zip(range(b), range(1, b + 1)))
        gt_xy, gt_dxdy = batch["gt_xy"], batch["gt_dxdy"]
        if "loss_mask" in batch and not batch.get("_local_mask"):  # None == every pedestrian valid (no device sync, capturable)
            loss_mask = batch["loss_mask"]
        else:
            if "loss_mask" in batch:  # this rank's verdict, taken by the loader on the host copy (device_crops.py)
                loss_mask = batch["loss_mask"]
                all_valid = loss_mask is None
                if all_valid:
                    loss_mask = torch.ones(b, dtype=torch.bool, device=gt_xy.device)
            else:
                loss_mask = ~gt_xy.isnan().any(2).any(0)
                all_valid = bool(loss_mask.all())
            if self.dist.enabled and self.dist.world_size > 1:
                # the masked steps issue other collectives than the unmasked ones (no shared discriminator context: two
                # scene-CNN passes per step): every rank takes the masked path as soon as ANY rank holds a NaN
                import torch.distributed as dist

                flag = torch.tensor([0.0 if all_valid else 1.0], device=gt_xy.device)
                self.dist.count_collective("mask.any")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.dist.group)
                all_valid = float(flag.item()) == 0.0
            if all_valid:
                loss_mask = None
            HF.force_masked(loss_mask is not None and self.dist.enabled and self.dist.world_size > 1)
        if loss_mask is not None:
            gt_dxdy, gt_xy = gt_dxdy[:, loss_mask], gt_xy[:, loss_mask]
        img = batch["features"] if "features" in batch else None
        args = (in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, metrics, loss_mask, img)
        if hasattr(self.rng, "begin_iteration"):  # device RNG: ONE launch draws every random number of the iteration
            self.rng.plan = (1, self.config.num_samples, self.config.num_expectation_samples)
            self.rng.d_steps = self.config.num_unrolling_steps + 1
            self.rng.begin_iteration(sub_batches, b, self.config.noise_dim, self.device)
        from mggan.hip import functions as HF

        if not (self.dist.enabled and not self.dist.stream_safe):  # (segmented sharded replay: the trainer set them off)
            HF.auto_branches(b, for_graph=torch.cuda.is_current_stream_capturing() or getattr(self, "_warming_up", False)
                             or self.dist.enabled)
        cfg = self.config
        run_d = self.total_iterations % max(int(cfg.num_gen_steps), 1) == 0 or self.epoch >= cfg.keep_gen_steps
        # sharded training with global-batch BatchNorm: the Gram matrix of the image patches comes FIRST and is all-reduced
        # -- the one exchange that serves every conv1 forward pass (statistics) and weight gradient of the iteration
        bn_sync = self.dist if (self.dist.enabled and getattr(cfg, "bn_sync", "global") == "global") else None
        if bn_sync is not None:
            HF.begin_images(img, sync=bn_sync)
        shared = None
        if getattr(self, "share_trunk", False) and loss_mask is None and cfg.num_unrolling_steps == 0 and run_d:
            # G is not updated between the no-grad generator call of the D step and the G step: one trunk
            # forward (with its backward graph) serves both; BatchNorm running stats still move twice (A.8)
            shared = {"g_trunk": self.G.trunk(in_xy, in_dxdy, sub_batches, img, passes=2)}
        # the Gram matrix of the image patches (image-only part of every conv1 weight gradient of the iteration) goes to
        # a side stream beside the latency-bound ROW PASS of the discriminator step (discriminator_step starts it once
        # its history context -- scene CNN and LSTM -- and the fake trajectories are queued; beside those it stretched all
        # of them: conv1_pool<8> 100 us instead of 30 at configs[1], the PM-network's 8 us chain 343 us at configs[2]);
        # its first reader is the scene CNN's adjoint of that step
        if bn_sync is None:
            HF.begin_images(img, defer=os.environ.get("MGGAN_GRAM_EARLY", "0") != "1")
        # abstract_train.py:136-150: the discriminator step runs when total_iterations % num_gen_steps == 0 or
        # epoch >= keep_gen_steps, and num_unrolling_steps + 1 times.  The reference's unrolling "backup" is
        # `self.D.state_dict()` -- references to the live parameters, not copies -- so its load_state_dict(backup)
        # after the generator / PM steps (:163-164) copies every tensor onto itself: the discriminator KEEPS all
        # unrolled updates.  That literal behaviour is what runs here (nothing to restore).
        if hasattr(self, "_open_iteration"):
            self._open_iteration()
        prep_was = HF.prep_cache(True)  # folded LSTM weights are kept per weight version inside the iteration
        # cross-iteration pipelining: the caller names the batch of the NEXT iteration (batch["next"]: its in_dxdy / features,
        # already on the device; capture_iteration(pipeline=True) names the static batch itself); its discriminator context is
        # issued beside this iteration's PM-network step (mggan/model/train.py: _issue_d_context)
        pipe = getattr(self, "_pipe", None)
        if pipe is not None:
            nxt = batch.get("next")
            ok = nxt is not None and run_d and self.pipeline_ok(loss_mask)
            pipe["next"] = (nxt["in_dxdy"], nxt["features"]) if ok else None
            if not ok and pipe["ctx"] is not None and not (run_d and loss_mask is None):
                self.drain_pipeline()
        try:
            if run_d:
                for _ in range(cfg.num_unrolling_steps + 1):
                    self.discriminator_step(*args, shared=shared)
            HF.launch_images()  # (no discriminator step in this iteration, or one that did not start it)
            self.generator_step(*args, shared=shared)
            self.net_chooser_step(*args)
            if pipe is not None and pipe["next"] is not None:  # (no PM-network step: --weighting_target none)
                pipe["next"] = None
            if pipe is not None and pipe["ctx"] is not None and torch.cuda.is_current_stream_capturing():
                # a capture ends with every fork joined: the held stream meets the main chain at the END of the iteration
                # (eager iterations leave it in flight until the next discriminator step takes the context)
                from mggan.model.train import _PIPE_BRANCH

                HF.join_branch(which=_PIPE_BRANCH, force=True)
        finally:
            HF.prep_cache(prep_was)
            HF.end_images()
            if hasattr(self, "_close_iteration"):
                self._close_iteration()
        self.total_iterations += 1
        self.dist.check()

    def capture_iteration(self, batch, warmup=3, pool=None, pipeline=None):
        """Capture one full D+G+PM iteration on `batch` into a HIP graph (needs --rng device: no host
        sync anywhere in the iteration).  Returns replay(metrics) which re-runs the iteration on the
        same static batch tensors (copy new data into them to change the input).  `warmup` eager iterations run
        first: per-batch tables, streams and scratch are created lazily (with host copies), which a capture cannot
        contain -- pass warmup=0 only when this trainer has already run an iteration on this batch."""
        if not getattr(self.rng, "on_device", False):
            raise RuntimeError("graph capture needs the device RNG (--rng device): the host RNG path reads the "
                               "PM-network logits back to the CPU")
        # (--num_gen_steps k: the discriminator step runs in some iterations only; the captured iteration is the one of
        #  self.total_iterations as it stands -- train()'s graph cache keeps one graph per (shape, with / without the step))
        from mggan.hip import functions as HF

        batch = dict(batch)
        batch["loss_mask"] = None
        batch.pop("_local_mask", None)
        # pipeline (None: MGGAN_PIPELINE): the captured iteration reads its discriminator context from where the previous
        # replay left it and ends with the context of the next one -- the static batch's own: replays feed the same buffers
        pipe = getattr(self, "_pipe", None)
        if pipe is not None:
            if pipeline is not None:
                pipe["on"] = bool(pipeline)
            if pipe["on"] and not self.dist.enabled and "features" in batch:
                batch["next"] = {"in_dxdy": batch["in_dxdy"], "features": batch["features"]}
        in_graph = False
        if self.dist.enabled:

            # peer-mapped all-reduce kernels (mggan/devcomm.py): the collectives are ordinary launches, the sharded
            # iteration is ONE graph like the single-GPU one, branch streams on.  Without them (ranks on several nodes,
            # IPC mapping refused, a gradient buffer larger than an arena slot, unequal shards) every torch.distributed
            # collective cuts the capture into graph segments.
            in_graph = self.dist.graph_safe(self.G, self.D)
            if not in_graph and not self.dist.equal_shards:
                raise RuntimeError("graph capture of a sharded iteration needs equal shards (dist.equal_shards)")
            if not in_graph and self.dist.stream_safe:
                # the decision is per trainer, not per collective: a torch.distributed call inside the single-graph
                # capture would fail, so the in-graph transports are set aside for this trainer's captures
                self.dist.close()
            HF.enable_branches(in_graph or os.environ.get("MGGAN_BRANCH_SHARDED", "0") == "1")
        self.graph_collectives = in_graph
        keep, self.defer_metrics = self.defer_metrics, True
        n_pending = len(self._pending)
        try:
            run, graph, captured = self._capture(batch, warmup, pool, in_graph, HF)
        except BaseException:
            # a failed capture must not leak into the eager iterations that follow: metric items recorded during the aborted
            # capture, queued reductions / weight-gradient GEMMs whose operands are gone, side-stream bookkeeping
            self.defer_metrics = keep
            del self._pending[n_pending:]
            HF.reset_deferred()
            # during a capture nothing executes, but every fold it RECORDED was booked as done: the buffers still hold the
            # previous weight version.  The eager iteration that follows a failed capture must fold again.
            HF.bump_weight_version(self.G)
            HF.bump_weight_version(self.D)
            raise
        pending, self._pending = self._pending, []
        self.defer_metrics = keep
        # nothing executed during the capture: the folded-weight buffers it allocated are empty until the first replay
        HF.bump_weight_version(self.G)
        HF.bump_weight_version(self.D)

        # A steady-state graph folds a module's LSTM weights only BEHIND the optimizer step that changed them and reads, at its
        # start, what the previous iteration left in the folded buffers.  True after the capture (nothing executed, the
        # eager iteration before it left every fold current) and after every replay; anything else that writes weights
        # in between -- an eager iteration, load_state_dict, a parameter broadcast -- shows in the fingerprint, and the
        # replay re-folds the module's buffers first.
        # (one fingerprint per trainer: the graphs of train()'s cache replay one after the other and each leaves the buffers
        #  as the next one expects them)
        fp = self._fold_fp = {"G": HF.weights_fingerprint(self.G), "D": HF.weights_fingerprint(self.D)}

        def replay(metrics=None, fetch=True):
            fp = self._fold_fp
            for tag, root in (("G", self.G), ("D", self.D)):
                if HF.weights_fingerprint(root) != fp[tag]:
                    HF.refold_root(root)
            run()
            HF.bump_weight_version(self.G)  # the optimizer steps inside the graph are invisible to the host
            HF.bump_weight_version(self.D)
            fp["G"], fp["D"] = HF.weights_fingerprint(self.G), HF.weights_fingerprint(self.D)
            self.dist.check()  # host read: a peer-mapped collective of an earlier replay has timed out -> stop here
            if metrics is not None and fetch:
                self._fetch([(metrics, items, snap) for _, items, snap in pending])

        replay.graph = graph
        replay.pending = pending  # [(metrics dict at capture, [(key, slot | (slot, slot))], snapshot buffer)]
        return replay

    def _capture(self, batch, warmup, pool, in_graph, HF):
        """The warm-up iterations and the capture itself.  -> (run, graph object, metrics dict of the captured iteration)"""
        scratch = defaultdict(list)
        side = HF.role_stream("capture")  # (one per process: torch's stream pool wraps around after 32 objects)
        side.wait_stream(HF._cur())
        with torch.cuda.stream(side):
            self._warming_up = True  # (the warm-up creates what the capture will use: streams, per-role tables -- same shape)
            try:
                for _ in range(warmup):
                    self.train_iteration(batch, scratch)
            finally:
                self._warming_up = False
        HF._cur().wait_stream(side)
        self.flush_metrics()
        if getattr(self, "_pipe", None) is not None and self._pipe["ctx"] is not None:
            # the context the last warm-up iteration issued is eager work on its held stream: the capture must find it done
            # (a captured stream cannot wait for work outside the capture)
            from mggan.model.train import _PIPE_BRANCH

            HF.join_branch(which=_PIPE_BRANCH, force=True)
            HF._BR["dirty"].discard(_PIPE_BRANCH)
        captured = defaultdict(list)
        it0 = self.total_iterations  # the captured iteration executes nothing: it does not count (the replays do)
        # no cyclic garbage collection while a stream is capturing: a collection that finalises an older trainer's graphs
        # or events in the middle of a capture calls HIP entry points that are not allowed there
        import gc

        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            return self._capture_graph(batch, pool, in_graph, HF, side, captured)
        finally:
            if gc_was:
                gc.enable()
            self.total_iterations = it0

    def _capture_graph(self, batch, pool, in_graph, HF, side, captured):
        if self.dist.enabled and not in_graph:
            # sharded iteration: the collectives cut the capture into graph segments and stay eager calls
            # between them (parallel.SegmentRecorder)
            from mggan.parallel import SegmentRecorder

            rec = SegmentRecorder()
            torch.cuda.synchronize()
            side.wait_stream(HF._cur())
            with torch.cuda.stream(side):
                self.dist.recorder = rec
                rec.begin()
                try:
                    self.train_iteration(batch, captured)
                finally:
                    rec.end()
                    self.dist.recorder = None
            HF._cur().wait_stream(side)
            run, graph = rec.replay, rec
            self.launch_mode = "{} hipGraph segments per iteration, collectives between them".format(rec.n_graphs)
        else:
            graph = torch.cuda.CUDAGraph()
            dot = os.environ.get("MGGAN_GRAPH_DOT")  # debugging aid: dump the captured graph's nodes and edges
            if dot:
                graph.enable_debug_mode()
            self._static_metrics = True
            try:
                # pool: graphs that are only ever replayed one after the other (train()'s graph cache) share their
                # scratch memory -- nothing but the metric buffer is read after a replay
                with torch.cuda.graph(graph, pool=pool):
                    self.train_iteration(batch, captured)
            finally:
                self._static_metrics = False
            if dot:
                graph.debug_dump(dot)
            run = graph.replay
            self.launch_mode = "hipGraph replay of the whole iteration" + (
                (", peer-mapped all-reduce kernels inside it" if self.dist.devcomm is not None else
                 ", RCCL all-reduce (ncclAllReduce on the capturing stream) inside it") if in_graph else "")
        return run, graph, captured

    def graph_mode(self):
        """Does train() replay captured iterations?  --graph on | off | auto (auto: whenever the configuration allows it).
        A captured iteration cannot contain a host synchronisation (so: the device RNG).  What changes from one iteration to
        the next lives in device memory (the learning rate of the cosine schedule, the 0.9 ** epoch of --weighting_target
        mgan) or in the graph cache's key (--num_gen_steps k / --keep_gen_steps: iterations with and without the
        discriminator step are two graphs per shape)."""
        cfg = self.config
        mode = getattr(cfg, "graph", "auto")
        ok = bool(getattr(self.rng, "on_device", False))
        if mode == "on" and not ok:
            raise ValueError("--graph on needs --rng device")
        if self.dist.enabled and not self.dist.equal_shards:
            # a sharded iteration is only capturable with equal shards (unequal ones read the global row count back to
            # the host): without them train() launches eagerly instead of trying -- and failing -- a capture per shape
            return False
        return ok and mode != "off"

    def train(self):
        cfg = self.config
        kw = dict(synthetic_scenes=getattr(cfg, "synthetic_scenes", 64), synthetic_peds=getattr(cfg, "synthetic_peds", 0))
        if getattr(cfg, "cache_device", 0):
            kw["cache_device"] = self.device
        workers = cfg.workers
        if cfg.dataset != "synthetic" and getattr(cfg, "crop_device", "auto") != "off":
            # scene crops on the GPU; the host half of a batch (trajectory transforms, the augmentation's geometry) runs where
            # --workers says -- in this process by default, like the reference's loader and with its numpy draw order; with
            # --workers N in N forked processes that hand their batches over through shared-memory slots
            # (mggan/data_utils/device_crops.py: DeviceCropLoader).  --rng device without --workers: two such processes
            # (MGGAN_LOADER_WORKERS=0 keeps the host half in this process)
            kw["crop_device"] = self.device
            if workers == 0 and getattr(cfg, "rng", "host") == "device" and (os.cpu_count() or 1) >= 8:
                workers = 2  # (--rng device is not seed-identical with the reference anyway: take the loader processes)
            workers = int(os.environ.get("MGGAN_LOADER_WORKERS", str(workers)))
        pad = getattr(cfg, "graph_pad", "auto")
        graphs = self.iteration_graphs = IterationGraphs(
            self, getattr(cfg, "graph_shapes", 8), pad=pad, bucket=getattr(cfg, "graph_bucket", "quarter"),
            capture=self.graph_mode(), bucket_limit=getattr(cfg, "graph_buckets", 64)) if (self.graph_mode() or pad == "on") else None
        self.epoch_seconds, self.epoch_iterations = [], []  # wall time of the training loop of every epoch (bench.py)
        train_loader = get_dataloader(dataset=cfg.dataset, phase="train", augment=cfg.augment,
                                      batch_size=cfg.batch_size, workers=workers, shuffle=True, **kw)
        val_loader = get_dataloader(dataset=cfg.dataset, phase="val", augment=False, batch_size=cfg.batch_size,
                                    workers=0 if "crop_device" in kw else workers, shuffle=False, **kw)
        track_metric = "val/ADE k=20"
        min_track_metric = math.inf
        self.total_iterations = 0  # a local of train() in the reference (abstract_train.py:104): every call starts at 0
        for epoch in range(cfg.epochs):
            self.epoch += 1
            if hasattr(self, "_pm_reg"):
                self._pm_reg()  # (the device word of 0.9 ** epoch: before the epoch's first replay)
            self.D.train()
            self.G.train()
            metrics = defaultdict(list)
            self.dist.host_barrier()  # loaders, validation and checkpoints are per-rank host phases
            torch.cuda.synchronize()
            t_epoch, n_it = time.perf_counter(), 0
            # the logged losses stay on the device until the end of the epoch (one read-back every 64 iterations at most):
            # an eager iteration is host-bound, and a read-back per step drains the launch queue three times per iteration
            keep_defer, self.defer_metrics = getattr(self, "defer_metrics", False), True
            for batch in train_loader:
                n_it += 1
                if graphs is not None and graphs.step(batch, metrics):
                    continue
                batch = self.to_device(batch)
                self.train_iteration(batch, metrics)
                if n_it % 64 == 0 and hasattr(self, "flush_metrics"):
                    self.flush_metrics()
            self.defer_metrics = keep_defer
            if hasattr(self, "flush_metrics"):
                self.flush_metrics()
            if graphs is not None:
                r0, e0 = (graphs.history[-1][:2] if graphs.history else (0, 0))
                graphs.flush(metrics)  # the epoch's replayed iterations: one D2H of their summed metric snapshots
                if graphs.history and self.dist.rank == 0:
                    r1, e1, n_g = graphs.history[-1]
                    # (stderr: a caller such as bench.py owns stdout -- its one JSON line must stay the only thing there)
                    print("[mggan] epoch {}: {} of {} iterations replayed from {} graph(s)".format(
                        self.epoch, r1 - r0, (r1 - r0) + (e1 - e0), n_g), file=sys.stderr)
            torch.cuda.synchronize()
            self.epoch_seconds.append(time.perf_counter() - t_epoch)
            self.epoch_iterations.append(n_it)
            self.dist.check(sync=True)

            if self.epoch % cfg.val_every == 0:
                self.D.eval()
                self.G.eval()
                with torch.no_grad():
                    m = self.check_accuracy(val_loader, vis=True, prefix="val/", num_k=cfg.top_k_test)
                    for k, v in m.items():
                        metrics["val/{}".format(k)].append(v)
                cur = mean(metrics[track_metric])
                if cur < min_track_metric:
                    print('Saving best model... "{}: Before: {}, After: {}'.format(track_metric, min_track_metric, cur))
                    min_track_metric = cur
                    self.save(checkpoint_name="checkpoint_best.pth")

            metrics = {k: float(np.mean(v)) for k, v in metrics.items()}
            self.writer.log(metrics, epoch)
            if self.epoch % cfg.save_every == 0:
                self.save()
            self.l2_weight *= cfg.l2_decay_rate
            if hasattr(self, "_set_l2_weight"):
                self._set_l2_weight(self.l2_weight)
            self.lr_schedulerD.step()
            self.lr_schedulerG.step()
            self.writer.save()
        return metrics

    def save(self, checkpoint_name=None):
        save_obj = {"generator": self.G.state_dict(), "discriminator": self.D.state_dict(),
                    "gen_opt": self.optimizerG.state_dict(), "disc_opt": self.optimizerD.state_dict()}
        if not checkpoint_name:
            checkpoint_name = "checkpoint_{}.pth".format(self.epoch)
        torch.save(save_obj, self.model_save_dir / checkpoint_name)

    @classmethod
    def load(cls, log_path, exp_name, version, checkpoint):
        version_dir = Path(log_path) / exp_name / "version_{}".format(version)
        checkpoint_dir = version_dir / "checkpoints"
        if checkpoint == "latest":
            epochs = [int(p.stem.split("_")[1]) for p in checkpoint_dir.iterdir() if p.stem.split("_")[1] != "best"]
            checkpoint = max(epochs)
        state_dicts = torch.load(checkpoint_dir / "checkpoint_{}.pth".format(checkpoint), map_location="cpu")
        config = read_meta_tags(version_dir / "meta_tags.csv")
        g, d = cls.construct_model(config)
        m = cls(g, d, config, Experiment(log_path, name=exp_name, version=version))
        m.G.load_state_dict(state_dicts["generator"], strict=False)
        m.D.load_state_dict(state_dicts["discriminator"], strict=False)
        try:
            m.optimizerD.load_state_dict(state_dicts["disc_opt"])
            m.optimizerG.load_state_dict(state_dicts["gen_opt"])
        except Exception as e:  # same tolerance as the reference (abstract_train.py:278-283)
            print("Could not restore optimizers.", str(e))
        return m, config

    @classmethod
    def load_from_path(cls, version_path, checkpoint="best"):
        version_path = Path(version_path)
        assert "version" in version_path.stem, "Input path should point to model version directory."
        return cls.load(version_path.parent.parent, version_path.parent.name, int(version_path.stem.split("_")[1]),
                        checkpoint)

    @abc.abstractmethod
    def generator_step(self, in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, train_metrics, loss_mask, img=None,
                       shared=None):
        pass

    @abc.abstractmethod
    def discriminator_step(self, in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, train_metrics, loss_mask, img=None,
                           shared=None):
        pass

    @abc.abstractmethod
    def net_chooser_step(self, in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, metrics, loss_mask, img):
        pass

    @abc.abstractmethod
    def check_accuracy(self, loader, vis=False, prefix="", num_k=20):
        pass


class _PadDesc(ctypes.Structure):  # mirrors csrc/crop.hip: mggan_pad_batch
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("inner", ctypes.c_long), ("outer", ctypes.c_int),
                ("position", ctypes.c_int)]


def bucket_size(n, mode="quarter"):
    """Smallest bucket >= n.  quarter: multiples of 8 up to 64, then four buckets per octave (64, 80, 96, 112, 128, 160,
    ...: at most a fifth of a bucket is padding); pow2: powers of two from 8 on.  -> (bucket, the bucket below it)"""
    n = max(int(n), 1)
    if mode == "pow2":
        p = 8
        while p < n:
            p *= 2
        return p, (p // 2 if p > 8 else 0)
    if n <= 64:
        p = -(-n // 8) * 8
        return p, p - 8
    step = 16
    while 8 * step < n:  # buckets (4 step, 8 step] in steps of `step`
        step *= 2
    p = -(-n // step) * step
    return p, p - step


class IterationGraphs:
    """train()'s graph cache (reference loop: abstract_train.py:114-168).

    Uniform batches (every scene the same size -- the synthetic benchmark shapes): key = the batch's exact shape.  A shape
    seen for the first time gets static input buffers and runs ONE eager iteration on them (which also builds, and pins,
    every per-shape table); at its second appearance the iteration is captured (nothing executes during a capture) and
    replayed; from then on a batch of that shape costs one copy into the static buffers and one graph launch.

    Ragged batches (the reference loader, /root/reference/mggan/data_utils/trajectories_scene.py:40-78: a new tuple of
    scene sizes almost every batch) are padded to a SHAPE BUCKET: (pedestrians rounded up to bucket_size, scene slots,
    largest scene rounded up to 16).  The padding consists of inert phantom pedestrians behind the real ones (constant
    trajectories, black crops, scenes of their own): every row-wise kernel computes them like any other row, the kernels
    that mix rows -- BatchNorm statistics of the scene CNNs, loss means, generator counts, per-scene L2 -- read the number of
    real pedestrians / scenes from device memory (HF.StaticSceneTables.dims), and the scene tables live at fixed device
    addresses that are re-filled for every batch (one host-to-device copy).  One captured graph per bucket then replays
    every batch that falls into it; the results equal the unpadded iteration up to the order of the floating-point sums
    (tests/test_train_loop.py) and are bit-identical to eager launches on the same padded batch.

    Batches with NaN ground truth (masked pedestrians), shapes / buckets beyond `limit` and a failed capture take the
    eager path.  The logged losses of replayed iterations are summed on the device and read back once per epoch."""

    class Entry:
        def __init__(self, static):
            self.static, self.replay, self.failed, self.tables = static, None, False, None

    def __init__(self, trainer, limit=8, pad="auto", bucket="quarter", capture=True, bucket_limit=64):
        self.tr, self.limit, self.entries = trainer, int(limit), {}
        # padded buckets have their own, larger limit: the reference loader's batches fall into a few dozen of them
        # (pedestrian count in quarter-octave steps x scene slots x largest-scene class), where a limit of 8 kept the first
        # eight buckets of a run and sent every other batch down the eager path for good
        self.bucket_limit = int(bucket_limit)
        self.history = []  # (replays, eager iterations, graphs) at every flush, i.e. once per epoch: the replay rate
        self.pad, self.bucket, self.capture = pad, bucket, bool(capture)
        self.pool = None  # the graphs' shared memory pool (created with the first capture)
        self._acc = {}  # id(snapshot buffer) -> [buffer, running sum, count, items]
        self.replays = self.eager = self.padded = 0

    @staticmethod
    def key_of(batch):
        sse = batch["seq_start_end"]
        return (tuple(int(e) - int(s) for s, e in sse),) + tuple(
            (k, tuple(v.shape)) for k, v in sorted(batch.items()) if torch.is_tensor(v))

    PED_AXIS = {"in_xy": 1, "in_dxdy": 1, "gt_xy": 1, "gt_dxdy": 1, "features": 0}

    def bucket_of(self, batch):
        """-> (key, b_pad, S_pad, max_n_pad) of the shape bucket a ragged batch is padded to, or None (uniform batch,
        padding switched off or not supported by this configuration, a scene of more than 64 pedestrians, tensors the
        padder does not know)."""
        tr = self.tr
        if self.pad == "off" or not (self.capture or self.pad == "on") or not tr.padding_ok():
            return None
        sse = [(int(s), int(e)) for s, e in batch["seq_start_end"]]
        sizes = [e - s for s, e in sse]
        if not sizes or len(set(sizes)) == 1 or max(sizes) > 64 or min(sizes) < 1:
            return None
        b = batch["in_xy"].shape[1]
        if sse[0][0] != 0 or sse[-1][1] != b or any(sse[i][1] != sse[i + 1][0] for i in range(len(sse) - 1)):
            return None
        if any(torch.is_tensor(v) and k not in self.PED_AXIS for k, v in batch.items()):
            return None
        b_pad, below = bucket_size(b, self.bucket)
        S_b = bucket_size(len(sse), "quarter")[0]
        from mggan.hip.functions import StaticSceneTables

        S_pad = StaticSceneTables.slots_for(S_b, b_pad, below + 1)
        max_n = -(-max(max(sizes), StaticSceneTables.PHANTOM_SCENE) // 16) * 16
        other = tuple((k, tuple(d for i, d in enumerate(v.shape) if i != self.PED_AXIS[k]))
                      for k, v in sorted(batch.items()) if torch.is_tensor(v))
        return ("pad", b_pad, S_pad, max_n) + other, b_pad, S_pad, max_n

    def _padded_entry(self, key, b_pad, S_pad, max_n, batch):
        """Static buffers of a bucket: the batch tensors at the padded size, pre-filled with the phantom pedestrians
        (constant positions x = slot within a phantom scene, zero steps, black crops), and the static scene tables."""
        from mggan.hip import functions as HF

        tr = self.tr
        tables = HF.StaticSceneTables(b_pad, S_pad, max_n, tr.device)
        tables.register(tr.D.__dict__.setdefault("_pair_scenes", HF.BoundedCache(16)))
        static = {}
        for k, v in batch.items():
            if not torch.is_tensor(v):
                continue
            if v.dtype != torch.float32:
                raise TypeError("padded batches hold float32 tensors ({}: {})".format(k, v.dtype))
            shape = list(v.shape)
            shape[self.PED_AXIS[k]] = b_pad
            static[k] = torch.zeros(shape, dtype=v.dtype, device=tr.device)  # (filled by mggan_pad_batch, every batch)
        ent = self.Entry(static)
        static["seq_start_end"], static["loss_mask"], static["pad"] = tables.seq_start_end, None, tables
        ent.tables = tables
        return ent

    def _load(self, ent, batch):
        """Copy a batch into an entry's static buffers (padded entries: the real pedestrians in front, the phantom
        pedestrians restored behind them where the previous batch was longer, the scene tables re-filled)."""
        from mggan.hip import functions as HF
        from mggan.hip.lib import lib

        b = batch["in_xy"].shape[1]
        if ent.tables is None:
            # an exact-shape entry: the float32 tensors with a pedestrian axis in ONE launch too (mggan_pad_batch with nothing
            # to pad: six copy launches of a device batch otherwise, in series in front of every replay)
            keys = []
            for k, v in batch.items():
                if torch.is_tensor(v):
                    if v.is_cuda and v.dtype == torch.float32 and k in self.PED_AXIS and v.numel():
                        keys.append(k)
                    else:
                        ent.static[k].copy_(v, non_blocking=True)
            if not keys:
                return
        else:
            keys = [k for k, v in batch.items() if torch.is_tensor(v)]
        descs = (_PadDesc * len(keys))()
        keep = []
        for d, k in zip(descs, keys):
            v = batch[k].to(self.tr.device, non_blocking=True).contiguous()  # (a no-op for device-resident batches)
            keep.append(v)
            ax, dst = self.PED_AXIS[k], ent.static[k]
            outer = 1
            for n in v.shape[:ax]:
                outer *= int(n)
            d.src, d.dst, d.inner, d.outer = v.data_ptr(), dst.data_ptr(), v.numel() // max(outer * b, 1), outer
            d.position = 1 if k in ("in_xy", "gt_xy") else 0
        # ONE launch: the real pedestrians in front, the phantom pedestrians (constant positions x = slot within a phantom
        # scene, zero steps, black crops) behind them
        lib.mggan_pad_batch(ctypes.addressof(descs), len(keys), b, ent.tables.b if ent.tables is not None else b,
                            HF.StaticSceneTables.PHANTOM_SCENE, HF._s())
        if ent.tables is not None:
            ent.tables.fill(batch["seq_start_end"])

    def step(self, batch, metrics):
        """-> True when the batch was consumed here (eagerly on its static buffers, or as a replay)."""
        from mggan.hip import functions as HF

        tr = self.tr
        if "loss_mask" in batch:
            valid = batch["loss_mask"] is None
        else:
            gt = batch["gt_xy"]
            valid = (not gt.is_cuda) and not bool(torch.isnan(gt).any())  # (a device batch would need a sync to tell)
        if not valid:
            return False
        if "_local_mask" in batch:  # (the loader's verdict has been used; the static copies are valid by construction)
            batch = {k: v for k, v in batch.items() if k != "_local_mask"}
        bucket = self.bucket_of(batch)
        if bucket is None and not self.capture:
            return False
        key = bucket[0] if bucket is not None else self.key_of(batch)
        cfg = tr.config
        if int(cfg.num_gen_steps) != 1:  # (abstract_train.py:136-150 of the reference: does this iteration hold a D step?)
            run_d = tr.total_iterations % max(int(cfg.num_gen_steps), 1) == 0 or tr.epoch >= cfg.keep_gen_steps
            key = (key, bool(run_d))
        ent = self.entries.get(key)
        if ent is None:
            n_pad = sum(1 for e in self.entries.values() if e.tables is not None)
            if (n_pad >= self.bucket_limit) if bucket is not None else (len(self.entries) - n_pad >= self.limit):
                return False
            if bucket is not None:
                ent = self.entries[key] = self._padded_entry(*bucket, batch)
                self._load(ent, batch)
            else:
                static = {k: (v.to(tr.device).clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
                static["seq_start_end"] = [[int(s), int(e)] for s, e in batch["seq_start_end"]]  # the tables are keyed by it
                static["loss_mask"] = None
                ent = self.entries[key] = self.Entry(static)
            with HF.pin_tables():
                tr.train_iteration(ent.static, metrics)
            self.eager += 1
            self.padded += ent.tables is not None
            return True
        self._load(ent, batch)
        self.padded += ent.tables is not None
        if self.capture and ent.replay is None and not ent.failed:
            try:
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                with HF.pin_tables():
                    ent.replay = tr.capture_iteration(ent.static, warmup=0, pool=self.pool)
            except Exception as exc:  # noqa: BLE001
                ent.failed = True
                print("[mggan] graph capture failed for batch shape {} ({}: {}); eager launches for this shape".format(
                    key[0][:8] if bucket is None else key[:4], type(exc).__name__, exc))
                torch.cuda.synchronize()
        if ent.replay is None:
            tr.train_iteration(ent.static, metrics)
            self.eager += 1
            return True
        ent.replay(None, False)
        tr.total_iterations += 1
        self.replays += 1
        seen, totals, snaps = set(), [], []
        for _, items, snap in ent.replay.pending:  # the replay refreshed its snapshot buffers: add them up on the device
            acc = self._acc.setdefault(id(snap), [snap, torch.zeros_like(snap), 0, set()])
            acc[3].add(tuple(items))
            if id(snap) not in seen:
                seen.add(id(snap))
                totals.append(acc[1])
                snaps.append(snap)
                acc[2] += 1
        if totals:
            torch._foreach_add_(totals, snaps)  # (one launch for the snapshots of all three steps)
        return True

    def flush(self, metrics):
        """The replayed iterations of the epoch enter `metrics` as their mean, once per iteration (the epoch's logged
        value is the mean over its iterations, abstract_train.py:194).  Sharded runs: ONE host-side exchange per flush,
        issued by every rank whatever it replayed (a rank whose batches all ran eagerly, or whose buckets were captured in
        another order, must not issue fewer or differently ordered collectives than its peers): the per-metric (sum, count)
        pairs travel as one object and are merged by metric name."""
        tr = self.tr
        local = {}  # metric key -> [sum over this rank's replayed iterations, their number]
        for acc in self._acc.values():
            snap, total, count, item_lists = acc
            if count == 0:
                continue
            v = total.double().cpu().numpy()
            for items in sorted(item_lists):
                for key, slot in items:
                    e = local.setdefault(key, [0.0, 0])
                    e[0] += float(v[slot] if isinstance(slot, int) else v[slot[0]] + v[slot[1]])
                    e[1] += count
            total.zero_()
            acc[2] = 0
        merged = local
        if tr.dist.enabled:
            import torch.distributed as dist

            parts = [None] * tr.dist.world_size
            dist.all_gather_object(parts, local, group=tr.dist.group)
            merged = {}
            for part in parts:
                for key, (s_, c_) in part.items():
                    e = merged.setdefault(key, [0.0, 0])
                    e[0] += s_
                    e[1] += c_
        for key in sorted(local):
            s_, c_ = merged[key]
            # (a logged loss of a sharded iteration is this rank's share of the global-batch mean: the ranks' values add up)
            val = s_ / max(c_, 1) * tr.dist.world_size if tr.dist.enabled else s_ / max(c_, 1)
            metrics[key].extend([val] * local[key][1])
        if self.replays + self.eager:
            self.history.append((self.replays, self.eager, len(self.entries)))


def read_meta_tags(path):
    """meta_tags.csv (test_tube's `key,value` rows, utils.py:97-131 of the reference) over the parser defaults ->
    Namespace.  `gpus` stays the device-id string it is on the command line ("0" must not turn into a falsy 0)."""
    import csv

    defaults = {a.dest: a.default for a in get_parser()._actions if not a.required and a.dest != "help"}
    with open(path) as f:
        for row in csv.DictReader(f):
            defaults[row["key"]] = row["value"] if row["key"] == "gpus" else _convert(row["value"])
    return Namespace(**defaults)


def _convert(val):
    if isinstance(val, str):
        if val.lower() == "true":
            return True
        if val.lower() == "false":
            return False
        if val == "None" or val == "":
            return None
    for c in (int, float, str):
        try:
            return c(val)
        except (ValueError, TypeError):
            pass
    return val
