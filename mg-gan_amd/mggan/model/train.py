"""PiNetMultiGeneratorGAN: the three optimisation steps of MG-GAN with the reference's surface
(/root/reference/mggan/model/train.py:18-289,578-662) on hand-written HIP kernels.

    python mg-gan_amd/mggan/model/train.py --name test --num_gens 4 --dataset synthetic --epochs 2
"""
import os
import random
import sys
from functools import partial
from math import ceil
from collections import defaultdict
from pathlib import Path

if __name__ == "__main__":  # allow running the file directly like the reference's README
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

import numpy as np
import torch

from mggan.abstract_train import MultiGeneratorGAN
from mggan.evaluation import evaluate_ade_fde
from mggan.hip import functions as HF
from mggan.hip.lib import lib
from mggan.logging import Experiment
from mggan.model.config import get_parser
from mggan.model.model_factory import construct_model
from mggan.utils import (expected_sample_idxs, get_selection_indices, thresholded_generators, to_numpy,
                         uniform_sample_idxs)

# slots of the device-side metric buffer (one D2H copy per step instead of one .item() per metric)
M_REAL, M_FAKE, M_CE_D, M_L2, M_ADV, M_CLF, M_PM, M_PROBS = 0, 1, 2, 3, 4, 5, 6, 8


_GRAM_LATE = os.environ.get("MGGAN_GRAM_LATE", "0") == "1"
# The generator step's forward pass (sampling, row bucketing, the K-sample rollout with its saved state) reads nothing the
# discriminator step's update writes -- G's weights, the shared trunk, the PM-network logits, this iteration's random numbers --
# so it is issued on a branch stream of its own BEFORE the discriminator step's backward pass and runs beside it (MGGAN_G_EARLY=0:
# inside the generator step, beside the discriminator's history context).  Same arithmetic, same order of random draws.
# Measured (one box, alternating; ms per iteration, off -> on): 1,280 pedestrians 1.362 -> 1.390 (the rollout takes CUs from
# two balanced latency-bound chains), 2,560: 2.027 -> 1.989, 4,096: 2.840 -> 2.779, 6,144: 3.755 -> 3.642, 8,192: 4.578 -> 4.500
# (it fills the tail where only the discriminator's half-empty C = 8 convolution adjoints run): on from MGGAN_G_EARLY_MIN_B.
# (At 1,280 pedestrians forking it BEHIND the main chain's adjoints instead -- beside the branch's tail and the optimizer -- is
#  worse still, 1.371 -> 1.453: the rollout's 1,600 tiles hold every CU while the tail's small launches wait for a slot.)
_G_EARLY = os.environ.get("MGGAN_G_EARLY", "1") == "1"
_G_EARLY_MIN_B = int(os.environ.get("MGGAN_G_EARLY_MIN_B", "2048"))
_EARLY_BRANCH = 5
_G_SCENE_JOIN_EARLY = os.environ.get("MGGAN_G_SCENE_JOIN", "late") == "early"
_G_SCENE_IN_PLACE = os.environ.get("MGGAN_G_SCENE_IN_PLACE", "1") != "0"
_PM_PREFOLD = os.environ.get("MGGAN_PM_PREFOLD", "0") == "1"  # measured (round 6): 1.3226 -> 1.3292 ms at 64 x 20 -- the fork and the
#   event cost more than the 6 us fold they take off the chain (the lesson of DESIGN section 4 again); off
# Cross-iteration pipelining (MGGAN_PIPELINE=1, capture_iteration(pipeline=True)): the NEXT iteration's discriminator context --
# history LSTM + scene CNN of D on the next batch, with the weights D holds since this iteration's discriminator update -- is
# issued on a held branch stream beside the PM-network step, whose tail (the generator's scene-CNN adjoint) runs alone on the
# chip; the next discriminator step finds it done.  Same arithmetic in the same order (BatchNorm running statistics included:
# nothing else touches D's between the two points), so the weights are bit-identical to the in-order schedule; an issued
# context that is never consumed is rolled back (drain_pipeline).
# MEASURED (round 6, one box, three alternating pairs, ms per iteration in-order -> pipelined): 64 x 20: 1.358 -> 1.435-1.48
# (issued at PM.begin), 1.46-1.50 (before the PM step's backward pass), 1.50-1.53 (before the generator step's), 1.53 (behind the
# PM step's main backward chain); 256 x 32: 4.48 -> 4.60-4.64 / 4.61-4.65 / 4.96 / 4.68.  It LOSES everywhere: in the
# discriminator step the context already runs beside the fake-trajectory branch and is exposed for ~40 us only (64 x 20), while
# beside the PM step its persistent CNN grids and the generator's share the CUs (a sum, not a maximum) -- and what does not fit
# extends the iteration.  Off by default; the mechanism (ReplayAlloc, roll-back of the running statistics) is kept and tested.
_PIPE_BRANCH = 6
_PIPE_AT = os.environ.get("MGGAN_PIPE_AT", "pm_begin")  # pm_begin | pm_bwd | g_bwd


class PiNetMultiGeneratorGAN(MultiGeneratorGAN):
    def __init__(self, generator, discriminator, config, writer):
        super().__init__(generator, discriminator, config, writer)
        assert self.gan_type in ("mgan", "gan"), self.gan_type
        if config.weighting_target not in ("ml", "l2", "endpoint", "mgan", "none"):
            raise ValueError("HIP path implements weighting_target 'ml' (default), 'l2', 'endpoint', 'mgan' and 'none' "
                             "('disc_scores' raises NotImplementedError in the reference too)")
        if config.l2_loss_type not in ("min_g_z", "min_z", "min_g_min_z", "none"):
            raise ValueError("HIP path implements the per-scene min-over-samples L2 (not 'mse')")
        dev = self.device
        self._m = torch.zeros(M_PROBS + 64, dtype=torch.float32, device=dev)
        self._one = torch.ones((), device=dev)
        HF.register_unit_grad(self._one)
        self._w = {k: (self._one if float(v) == 1.0 else torch.full((), float(v), device=dev)) for k, v in
                   (("l2", config.l2_loss_weight), ("clf", config.clf_loss_weight), ("pi", config.pi_net_loss_weight))}
        if float(getattr(config, "l2_decay_rate", 1)) != 1.0:  # decays per epoch (abstract_train.py:197): its own device word
            self._w["l2"] = torch.full((), float(config.l2_loss_weight), device=dev)
        self.defer_metrics = False  # True: steps only enqueue work; fetch with flush_metrics()
        # legitimate de-duplications (same results): D's history context once per D step, and the
        # generator trunk once for {no-grad G call of the D step, G step} (G's weights do not change in between)
        self.share_context = True
        # branch stream of D's history LSTM in the generator step: 0 = behind D's scene CNN on the same stream,
        # 2 = a third stream (measured slower: the rollout forward next to it loses more than the LSTM gains)
        self.g_step_lstm_branch = int(os.environ.get("MGGAN_G_LSTM_BRANCH", "0"))
        # ... and, on the same stream, in front of the CNN (1) or behind it (0)
        # (round 5, three alternating pairs on one box: LSTM chain first 1.380 vs 1.385 ms at 64 x 20, 4.66 vs 4.73 ms at
        #  256 x 32 -- three small launches get their CU slots before the CNN's persistent grids fill the register files)
        self.g_step_lstm_first = os.environ.get("MGGAN_G_LSTM_FIRST", "1") == "1"
        # the same order in the discriminator step (LSTM on the main stream, the CNN forked behind it): 1.408 vs 1.388 ms and
        # 4.77 vs 4.72 ms -- there the CNN is the longer chain and starts late
        self.d_step_lstm_first = os.environ.get("MGGAN_D_LSTM_FIRST", "0") == "1"
        # ... and D's scene CNN of the generator step forked behind the generator's sampling launches (beside the rollouts
        # only): 1.408 vs 1.395 ms, 4.75 vs 4.69 ms -- lost; kept as a knob
        self.g_step_cnn_after_sampling = os.environ.get("MGGAN_G_CNN_AFTER_SAMPLING", "0") == "1"
        self.share_trunk = True
        # weight-gradient GEMMs on a side stream during backward, joined in optimizer.step
        self.overlap_wgrad = os.environ.get("MGGAN_OVERLAP_WGRAD", "0") == "1"  # measured: no gain inside a hipGraph (5.5 vs 5.3 ms)
        # True: AdamW leaves the gradients it consumed at zero (p.grad reads 0 after a step instead of the clipped
        # gradient), which removes the memset at the start of the next step; same parameter updates
        self.zero_grads_in_step = False
        self._pending = []
        self._bwd_stream = None
        self._pipe = {"on": os.environ.get("MGGAN_PIPELINE", "0") == "1", "stores": {}, "ctx": None, "key": None, "snap": None,
                      "next": None}

    def padding_ok(self):
        """Can train() pad ragged batches to shape buckets (abstract_train.IterationGraphs)?  The kernels that mix rows
        know about phantom pedestrians on the default path only: sways pooling, the fused pair pass and shared contexts,
        'ml' PM-network target, per-scene min-over-samples L2, one GPU."""
        cfg = self.config
        return (cfg.pool_type == "sways" and cfg.experiment == "multi_generator" and cfg.weighting_target == "ml"
                and cfg.l2_loss_type in ("min_g_z", "min_z", "min_g_min_z", "none") and self.gan_type in ("mgan", "gan")
                and self.share_context and self.share_trunk and getattr(self, "pair_passes", True)
                and int(cfg.num_unrolling_steps) == 0 and not self.dist.enabled
                and getattr(self.rng, "on_device", False))

    def _pm_reg(self):
        """0.9 ** epoch (train.py:612 of the reference: the entropy term of the 'mgan' PM-network target) as a device word that
        the loss kernel reads at run time; refreshed here whenever the epoch has moved on -- never inside a capture (train()
        refreshes it at the start of every epoch, before the first replay)."""
        t = self.__dict__.get("_pm_reg_dev")
        if t is None:
            t = self._pm_reg_dev = torch.zeros((), dtype=torch.float32, device=self.device)
            self._pm_reg_epoch = None
        if self._pm_reg_epoch != self.epoch and not torch.cuda.is_current_stream_capturing():
            t.fill_(0.9 ** self.epoch)
            self._pm_reg_epoch = self.epoch
        return t

    def _set_l2_weight(self, value):
        """self.l2_weight decays per epoch (abstract_train.py:197); the backward pass reads it from device memory."""
        if self._w["l2"] is not self._one:
            self._w["l2"].fill_(float(value))

    # ---- metric plumbing ---------------------------------------------------------------
    def _emit(self, train_metrics, items):
        """items: list of (key, slot | (slot_a, slot_b) summed).  Every step owns its slots of the metric buffer, so one
        snapshot per ITERATION serves all three steps (train_iteration -> _close_iteration); a step called on its own
        is snapshotted right away."""
        if getattr(self, "_iter_open", False):
            self._iter_items.append((train_metrics, items))
            return
        self._pending.append((train_metrics, items, self._m.clone()))
        if not self.defer_metrics:
            self.flush_metrics()

    def _open_iteration(self):
        self._iter_open, self._iter_items = True, []

    def _close_iteration(self):
        self._iter_open = False
        items, self._iter_items = self._iter_items, []
        if not items:
            return
        # inside a graph capture the buffer itself is registered: a replay refreshes it and replay(fetch=True) reads it
        # before the next replay -- no copy kernel in the graph; eager deferred iterations need their own snapshot
        snap = self._m if getattr(self, "_static_metrics", False) else self._m.clone()
        for train_metrics, it in items:
            self._pending.append((train_metrics, it, snap))
        if not self.defer_metrics:
            self.flush_metrics()

    def _fetch(self, pending):
        """One D2H copy per pending snapshot; sharded runs sum the per-rank partial means first (the loss
        kernels already divide by the GLOBAL row count)."""
        cache = {}
        for train_metrics, items, snap in pending:
            v = cache.get(id(snap))
            if v is None:
                red = self.dist.all_reduce_(snap.clone()) if self.dist.enabled else snap
                v = cache[id(snap)] = red.cpu().numpy()
            for key, slot in items:
                train_metrics[key].append(float(v[slot] if isinstance(slot, int) else v[slot[0]] + v[slot[1]]))

    def flush_metrics(self):
        self._fetch(self._pending)
        self._pending = []

    def _global(self, n):
        """Row count over all ranks (loss normalisers are means over the GLOBAL batch)."""
        return self.dist.global_count(n) if self.dist.enabled else n

    def _early_rider_counts(self, logits, b):
        """Sharded training: the generator step weights its rows by 1 / count(generator) over the GLOBAL batch
        (train.py:94-96 of the reference) -- an exchange of its own per iteration.  Its picks are already determined when
        the PM-network logits of the shared trunk exist (the uniforms were drawn at the start of the iteration), so this
        rank's counts are computed right there -- on the discriminator step's fake-trajectory branch, which has slack --
        and travel as riders in the f64 tail of the discriminator's gradient exchange (the tail fold copies them in,
        HF.set_rider_src); _gen_weights finds the global counts there."""
        self._rider_counts, self._rider_early = None, None
        HF.set_rider_src(self.D, None)
        if not (self.dist.enabled and logits is not None and getattr(self.rng, "on_device", False)
                and hasattr(self.rng, "peek_uniforms") and getattr(self.config, "bn_sync", "global") == "global"
                and os.environ.get("MGGAN_COUNT_RIDER", "1") != "0"):
            return
        K, g = int(self.config.num_samples), self.G.n_gs
        u = self.rng.peek_uniforms(b * K, logits.device)
        if u is None or g > HF.TAIL_RIDERS or logits.shape != (b, g) or not logits.is_contiguous():
            return
        buf = self.__dict__.get("_rider_buf")
        if buf is None or buf.device != logits.device:
            buf = self._rider_buf = torch.zeros(HF.TAIL_RIDERS, dtype=torch.float64, device=logits.device)
            self._count_scratch = torch.zeros(17, dtype=torch.int32, device=logits.device)
        # (where the caller stands: the end of the fake-sample branch.  Measured against it, same box, 64 x 20 / 256 x 32: on
        #  a stream of its own forked from there 1.46 vs 1.44 ms; on the main stream behind an event on the logits 1.46 vs
        #  1.44 and 4.65 vs 4.61 ms)
        lib.mggan_sample_counts(b, K, g, logits.data_ptr(), u.data_ptr(), self._count_scratch.data_ptr(), buf.data_ptr(), HF._s())
        HF.set_rider_src(self.D, buf)
        self._rider_early = (u.data_ptr(), b * K)

    def _ride_generator_counts(self):
        """Behind the discriminator step's backward pass: did its gradient tail take the counts along?"""
        early, self._rider_early = getattr(self, "_rider_early", None), None
        HF.set_rider_src(self.D, None)
        tails = self.D.__dict__.get("_grad_tails") or []
        if early is None or len(tails) != 1 or not self.D.__dict__.get("_rider_in_tail", False):
            return
        tail, tf = tails[0][0], tails[0][2]
        self._rider_counts = (tail[tf:tf + HF.TAIL_RIDERS], early[0], early[1])

    def _gen_weights(self, gen_idxs):
        """Batch-global 1/count(generator) weights (train.py:94-96) + int32 row targets in (k*b+ped) order."""
        g = self.G.n_gs
        rows = getattr(self.G, "last_rows", None)
        if rows is not None and rows.R == gen_idxs.numel():
            row_gen = rows.row_gen_pos  # produced by the row-bucketing step of this very forward pass
        else:
            row_gen = gen_idxs.t().reshape(-1).to(torch.int32)
        counts = torch.empty(g, dtype=torch.int32, device=self.device)
        inv = torch.empty(g, dtype=torch.float32, device=self.device)
        st = HF._s()
        rider, self._rider_counts = getattr(self, "_rider_counts", None), None
        check = os.environ.get("MGGAN_CHECK_RIDERS", "0") == "1" and not (  # (tests: a host sync and an exchange of its own --
            self.device.type == "cuda" and torch.cuda.is_current_stream_capturing())  # not inside a capture)
        # (the riders are the counts of THIS step's picks only if its sampling read the uniforms they were counted on: a
        #  re-drawn pool or an unplanned call in between falls back to the counted exchange)
        if self.dist.enabled and rider is not None and rider[2] == row_gen.numel() \
                and getattr(self.rng, "last_sample_u", None) == (rider[1], rider[2]):
            # the global counts came with the discriminator step's gradient exchange (_ride_generator_counts)
            if check:
                lib.mggan_gen_counts(row_gen.data_ptr(), row_gen.numel(), g, counts.data_ptr(), inv.data_ptr(), 0, 0, st)
                self.dist.all_reduce_(counts, what="count (check)")
                assert torch.equal(counts.double().cpu(), rider[0][:g].cpu()), (counts.cpu(), rider[0][:g].cpu())
            lib.mggan_inv_counts_f64(rider[0].data_ptr(), g, inv.data_ptr(), st)
            return row_gen, inv
        # (padded batch: the rows of phantom pedestrians -- row % b_pad >= n_real -- are not counted)
        lib.mggan_gen_counts(row_gen.data_ptr(), row_gen.numel(), g, counts.data_ptr(), inv.data_ptr(), HF._pad_ptr(),
                             HF._PAD["b"], st)
        if self.dist.enabled:
            self.dist.all_reduce_(counts, what="count")
            lib.mggan_inv_counts(counts.data_ptr(), g, inv.data_ptr(), st)
        return row_gen, inv

    def _backward(self, losses, grads, at_main_end=None):
        # (autograd's device worker thread is switched off for the pass: every node here is a Python function, and handing
        #  each one to another thread through the GIL cost 35 % of an eager iteration's host time -- 7.9 vs 5.1 ms on the
        #  configs[0] shape; stream semantics are unchanged: a node still runs on the stream of its forward)
        HF.enable_side_stream(self.overlap_wgrad)
        HF.defer_grad_reduce(True)  # weight-grad kernels leave partial sums; one batched reduce below
        try:
            if HF._BR["on"]:
                # autograd ends a backward pass by making the CALLER's stream wait for every stream its nodes ran
                # on.  Called from a throw-away stream, that wait lands there and the main stream stays where its
                # own last backward node left it: the weight-gradient GEMMs below then run beside the tail of the
                # branch streams (the scene CNN's convolution adjoints) instead of behind it.
                main = HF._cur()
                if self._bwd_stream is None:
                    self._bwd_stream = HF.role_stream("backward")
                self._bwd_stream.wait_stream(main)
                with torch.cuda.stream(self._bwd_stream), torch.autograd.set_multithreading_enabled(False):
                    torch.autograd.backward(losses, grads)
            else:
                with torch.autograd.set_multithreading_enabled(False):
                    torch.autograd.backward(losses, grads)
        finally:
            HF.enable_side_stream(False)
            HF.defer_grad_reduce(False)
        HF.join_side_stream()
        HF.mark("bwd.main.end")
        if at_main_end is not None:
            at_main_end()  # (work forked here runs beside the tail only: the scene CNN's adjoint on its branch, the reductions)
        HF.flush_wgrad_gemms()  # beside what is left of the branch streams' backward (the scene CNN's convolutions)
        HF.mark("bwd.gemms.end")
        if HF._MARKS["on"] and 0 in HF._BR["streams"]:
            with torch.cuda.stream(HF._BR["streams"][0]):
                HF.mark("bwd.branch0.end")
        # the batched reduction of the partial sums (and the LSTM un-folding behind it) waits for its producers only -
        # events recorded where branch streams queued partials - not for the branches' last kernels
        HF.flush_grad_reduces()
        HF.mark("bwd.reduce.end")
        if HF._BR["on"]:
            HF._cur().wait_stream(self._bwd_stream)
        HF.join_branch(force=True)  # backward nodes ran on the streams of their forwards

    # ---- the next iteration's discriminator context, ahead of time ----------------------------
    def _d_bn_buffers(self):
        bufs = []
        for mod in self.D.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                bufs += [mod.running_mean, mod.running_var, mod.num_batches_tracked]
        return bufs

    def pipeline_ok(self, loss_mask=None):
        cfg = self.config
        return (self._pipe["on"] and loss_mask is None and self.share_context and int(cfg.num_unrolling_steps) == 0
                and int(cfg.num_gen_steps) == 1 and not self.dist.enabled and HF._BR["on"] and HF.pad_dims() is None)

    def _issue_d_context(self, where):
        """At point `where` of the iteration: issue D.history_context for the batch announced as next (self._pipe['next'])."""
        nxt = self._pipe["next"]
        if nxt is None or where != _PIPE_AT or self._pipe["ctx"] is not None:
            return
        in_dxdy, img = nxt
        self._pipe["next"] = None
        key = (in_dxdy.data_ptr(), tuple(in_dxdy.shape), img.data_ptr(), tuple(img.shape))
        store = self._pipe["stores"].setdefault((key[1], key[3]), {})
        bufs = self._d_bn_buffers()
        if self._pipe["snap"] is None:
            self._pipe["snap"] = [torch.empty_like(t) for t in bufs]
        HF.hold_branch(_PIPE_BRANCH)
        try:
            with HF.branch(_PIPE_BRANCH):
                HF.mark("pipe.ctx.begin")
                HF.copy_small(list(zip(bufs, self._pipe["snap"])))  # (what drain_pipeline restores if nobody takes the context)
                with HF.ReplayAlloc(store):
                    ctx = self.D.history_context(in_dxdy, img, passes=2, lstm_first=True)
                HF.mark("pipe.ctx.end")
        except RuntimeError as exc:
            # the pass allocated differently than the recorded one: not a pass a captured graph could have re-run in place
            HF.hold_branch(_PIPE_BRANCH, False)
            self._pipe["on"] = False
            raise RuntimeError("cross-iteration pipelining switched off: {}".format(exc)) from exc
        self._pipe["ctx"], self._pipe["key"] = ctx, key

    def _take_d_context(self, in_dxdy, img):
        """-> the context issued for exactly this batch by the previous iteration, or None."""
        ctx, key = self._pipe["ctx"], self._pipe["key"]
        if ctx is None:
            return None
        if key != (in_dxdy.data_ptr(), tuple(in_dxdy.shape), img.data_ptr(), tuple(img.shape)) or not self.D.training:
            self.drain_pipeline()
            return None
        self._pipe["ctx"] = self._pipe["key"] = None
        HF.hold_branch(_PIPE_BRANCH, False)
        HF.join_branch(ctx[0], ctx[1], which=_PIPE_BRANCH)
        return ctx

    def drain_pipeline(self):
        """Forget a context that was issued ahead of time and not consumed: D's BatchNorm running statistics go back to what
        they were before it (its two momentum updates belong to a discriminator step that is not going to happen).  Called
        before anything outside the steady loop looks at D (validation, a checkpoint, another batch shape)."""
        if self._pipe["ctx"] is None:
            return
        self._pipe["ctx"] = self._pipe["key"] = None
        HF.hold_branch(_PIPE_BRANCH, False)
        HF.join_branch(which=_PIPE_BRANCH, force=True)
        HF.copy_small(list(zip(self._pipe["snap"], self._d_bn_buffers())))

    # ---- the three steps -----------------------------------------------------------------
    def discriminator_step(self, in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, train_metrics, loss_mask, img=None,
                           shared=None):
        m = self._m
        g_trunk = None if shared is None else shared.get("g_trunk")
        # the fake trajectories (no-grad generator call: PM-network, sampling, row bucketing, rollout) depend on
        # nothing the real pass computes: they are produced on a branch stream beside it.  (The label scalars
        # come from the numpy generator, noise and sampling from torch's: reordering the two is seed-neutral.)
        HF.mark("D.begin")
        with HF.branch(1), torch.no_grad():
            HF.mark("D.fake.begin")
            noise = self.rng.noise(1, self.config.noise_dim, sub_batches, self.device)
            gen_out, g_logits, gen_labels_gt = self.G(in_xy, in_dxdy, sub_batches, noise=noise, all_gen_out=False, img=img,
                                                      num_samples=1, mask=loss_mask,
                                                      trunk=None if g_trunk is None else tuple(t.detach() for t in g_trunk))
            if shared is not None and g_trunk is not None:
                shared["g_logits"] = g_logits  # same trunk, same weights in the generator step: not recomputed there
                self._early_rider_counts(g_logits, in_xy.size(1))
            rows_d = getattr(self.G, "last_rows", None)
            HF.mark("D.fake.end")
        # history LSTM + scene CNN of D are identical in the real and the fake pass: run them once
        ctx = None
        if loss_mask is None and self.share_context:
            ctx = self._take_d_context(in_dxdy, img)  # issued by the previous iteration (cross-iteration pipelining)?
            if ctx is None:
                ctx = self.D.history_context(in_dxdy, img, passes=2, lstm_first=self.d_step_lstm_first)
        # the Gram matrix of the image crops starts behind the history LSTM on this stream (by then D's scene CNN on its
        # branch stream is nearly through as well): beside the latency-bound row pass.  (Forked from this stream only: a
        # side stream with two parents inside a capture makes hipStreamEndCapture crash.)
        if not _GRAM_LATE:
            HF.launch_images()
        pair = ctx is not None and getattr(self, "pair_passes", True)
        kind = 1 if self.config.gan_obj == "LS" else 0  # phi_1 / phi_2: squared error for 'LS', BCE for 'NS' and 'MM'
        join_fake = lambda: HF.join_branch(gen_out.abs, gen_out.rel, gen_labels_gt, g_logits,
                                           None if rows_d is None else rows_d.row_gen_pos, which=1)
        ce_target = lambda: rows_d.row_gen_pos if rows_d is not None and rows_d.R == gen_labels_gt.numel() \
            else gen_labels_gt.t().reshape(-1).to(torch.int32)
        items = []
        if pair:
            # real and fake pass batched into one 2b-row pass (same results: every operator is row-wise / per scene),
            # and the two or three loss terms as one launch / one autograd node
            HF.mark("D.ctx.end")
            join_fake()
            HF.mark("D.pair.begin")
            y, branch_out = self.D.forward_pair(in_xy, in_dxdy, gt_dxdy, gen_out.rel, sub_batches, ctx)
            HF.mark("D.pair.end")
            b = in_xy.size(1)
            label_real, _ = self.rng.labels()
            _, label_fake = self.rng.labels()
            mgan = self.gan_type == "mgan"
            n_b = self._global(b)
            losses = [HF.GanLossesFn.apply(y.reshape(-1), branch_out.reshape(b, -1) if mgan else None, dict(
                nA=b, nB=b, labels=(label_real, label_fake), norms=(n_b, n_b, n_b), kind=kind,
                target=ce_target() if mgan else None,
                outs=(m[M_REAL:M_REAL + 1], m[M_FAKE:M_FAKE + 1], m[M_CE_D:M_CE_D + 1] if mgan else None)))]
            if mgan:
                items.append(("train/info_mgan_disc_loss", M_CE_D))
        else:
            real_result = self.D(in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, img=img, mask=loss_mask, context=ctx)
            if isinstance(real_result, tuple):
                real_result = real_result[0]
            n_real = self._global(real_result.numel())
            label_real, _ = self.rng.labels()
            real_loss = HF.BceMeanFn.apply(real_result.t().reshape(-1), label_real, None, None, m[M_REAL:M_REAL + 1],
                                           n_real, kind)
            join_fake()
            disc_out = self.D(in_xy, in_dxdy, gen_out.abs, gen_out.rel, sub_batches, img=img, mask=loss_mask, context=ctx)
            losses = [real_loss]
            if self.gan_type == "mgan":
                disc_out, branch_out = disc_out
                rows = branch_out.transpose(0, 1).reshape(-1, branch_out.shape[-1])
                ce_loss = HF.CeMeanFn.apply(rows, ce_target(), None, m[M_CE_D:M_CE_D + 1], self._global(rows.shape[0]))
                losses.append(ce_loss)
                items.append(("train/info_mgan_disc_loss", M_CE_D))
            _, label_fake = self.rng.labels()
            fake_loss = HF.BceMeanFn.apply(disc_out.t().reshape(-1), label_fake, None, None, m[M_FAKE:M_FAKE + 1],
                                           self._global(disc_out.numel()), kind)
            losses.append(fake_loss)
        items.append(("train/discr_loss", (M_FAKE, M_REAL)))

        HF.mark("D.loss.end")
        self.optimizerD.zero_grad()
        if _GRAM_LATE:
            HF.launch_images()  # (A/B knob: the Gram matrix beside the backward pass instead of the row pass)
        if (_G_EARLY and in_xy.size(1) >= _G_EARLY_MIN_B and shared is not None and shared.get("g_trunk") is not None
                and shared.get("g_logits") is not None and HF._BR["on"] and not HF._on_branch() and loss_mask is None
                and (not self.dist.enabled or self.dist.stream_safe)):
            # (sharded training over an in-graph transport too: the generator counts that ride in the discriminator's
            #  gradient exchange were counted on the fake-trajectory branch, on uniforms peeked BEFORE this call consumes them
            #  -- _early_rider_counts; between graph segments nothing may stay in flight, so the segmented replay keeps the
            #  in-step order)
            self._early_generator_forward(in_xy, in_dxdy, sub_batches, img, shared)
        self._backward(losses, [self._one] * len(losses))
        HF.mark("D.bwd.end")
        self._ride_generator_counts()
        ex = self.dist.all_reduce_grads(self.D, defer=True)
        self.optimizerD.step(self.config.clipping_threshold_d, zero_grad=self.zero_grads_in_step, exchange=ex)
        HF.mark("D.opt.end")
        self._emit(train_metrics, items)

    def _early_generator_forward(self, in_xy, in_dxdy, sub_batches, img, shared):
        """The generator step's G(...) call, queued on its own (held) branch stream from the discriminator step."""
        cfg = self.config
        HF.hold_branch(_EARLY_BRANCH)
        with HF.branch(_EARLY_BRANCH):  # (the branches G.forward would fork are no-ops on a branch: one chain)
            noise = self.rng.noise(cfg.num_samples, cfg.noise_dim, sub_batches, self.device)
            gen_out, _, gen_idxs = self.G(in_xy, in_dxdy, sub_batches, noise=noise, all_gen_out=False, img=img, mask=None,
                                          num_samples=cfg.num_samples, trunk=shared.get("g_trunk"),
                                          logits=shared.get("g_logits"))
            HF.mark("G.gen.end")
        shared["g_early"] = (gen_out, gen_idxs, getattr(self.G, "last_rows", None))

    def generator_step(self, in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, train_metrics, loss_mask, img=None,
                       shared=None):
        m, cfg = self._m, self.config
        b = in_xy.size(1)
        early = None if shared is None else shared.pop("g_early", None)
        # adversarial pass: D's weights get no gradient here (the reference discards them: D.zero_grad()
        # precedes backward and the next discriminator step zeroes them again, train.py:128,207)
        d_params = self.__dict__.get("_d_params")
        if d_params is None or d_params[0] is not self.D._flat:
            d_params = self._d_params = (self.D._flat, list(self.D.parameters()))  # (walked once per flat buffer)
        d_params = d_params[1]
        flags = [p.requires_grad for p in d_params]
        for p in d_params:
            p.requires_grad_(False)
        try:
            ctx_d = None
            if loss_mask is None and self.share_context:
                # D's history LSTM + scene CNN depend on neither G nor (in this step) any gradient: they run on
                # branch streams (CNN on 0, LSTM on 2) next to the generator's forward pass
                with torch.no_grad():
                    defer = None
                    if self.g_step_cnn_after_sampling and getattr(self.rng, "on_device", False) and HF._BR["on"]:
                        defer = lambda fn: setattr(self.G, "_after_sampling", fn)
                    # (with the generator's forward pass already issued from the discriminator step -- large batches -- the
                    #  chip is empty here: history LSTM and scene CNN side by side, 4.494 -> 4.466 ms at 256 x 32, three
                    #  alternating pairs; beside the rollout forward -- small batches -- the third stream loses 20 us)
                    lstm_branch = 2 if (early is not None and "MGGAN_G_LSTM_BRANCH" not in os.environ) else self.g_step_lstm_branch
                    # the lean row pass reads the scene features from the scene columns of its (b, 192) classifier input: the
                    # attention head writes them there itself (no broadcast launch behind the branch join)
                    slot = None
                    if _G_SCENE_IN_PLACE and defer is None and cfg.pool_type == "sways" and self.gan_type == "mgan":
                        slot = HF.OutSlot(torch.empty(b, 192, dtype=torch.float32, device=self.device), 128, 192)
                    ctx_d = self.D.history_context(in_dxdy, img, passes=1, lstm_branch=lstm_branch,
                                                   lstm_first=self.g_step_lstm_first, defer_cnn=defer, scene_out=slot)
            if early is not None:
                # issued from the discriminator step: this stream takes its results over (and autograd will run their
                # adjoints on the branch stream they were computed on)
                gen_out, gen_idxs, rows_early = early
                HF.hold_branch(_EARLY_BRANCH, False)
                HF.join_branch(gen_out.abs, gen_out.rel, gen_idxs,
                               *([v for v in vars(rows_early).values() if torch.is_tensor(v)] if rows_early is not None else []),
                               which=_EARLY_BRANCH)
                self.G.last_rows = rows_early
                gen_out = type(gen_out)(*HF.HandoffFn.apply(*gen_out))
            else:
                noise = self.rng.noise(cfg.num_samples, cfg.noise_dim, sub_batches, self.device)
                gen_out, _, gen_idxs = self.G(in_xy, in_dxdy, sub_batches, noise=noise, all_gen_out=False, img=img,
                                              mask=loss_mask, num_samples=cfg.num_samples,
                                              trunk=None if shared is None else shared.get("g_trunk"),
                                              logits=None if shared is None else shared.get("g_logits"))
                HF.mark("G.gen.end")
            if isinstance(ctx_d, list):
                late, self.G._after_sampling = getattr(self.G, "_after_sampling", None), None
                if late is not None:  # (the generator took a path without the device sampler: the hook never fired)
                    late()
                ctx_d = tuple(ctx_d)
            losses, grads, items = [], [], []
            if cfg.l2_loss_type != "none":
                # min-over-samples L2 (three small launches) beside the discriminator pass on the predictions; autograd
                # runs its backward on the same branch stream
                with HF.branch(1):
                    bm = gen_out.abs.shape[2]
                    sse = sub_batches
                    if bm != b:
                        # masked batch: the reference slices the MASKED predictions with the UNMASKED scene bounds
                        # (train.py:67-68; python slices clip at the end) and still divides by the unmasked b (:73)
                        sse = [[min(int(s0), bm), min(int(e0), bm)] for s0, e0 in sub_batches]
                        sse = [se for se in sse if se[1] > se[0]]
                    tb = HF.scene_tables(sse, bm, self.device)
                    min_l2 = HF.L2MinSceneFn.apply(gen_out.abs, gt_xy, tb, self._global(b), m[M_L2:M_L2 + 1])
                    losses.append(min_l2)
                    grads.append(self._w["l2"])
                    items.append(("train/L2_loss", M_L2))
                    HF.mark("G.l2.end")

            if ctx_d is not None:
                # (only the history encoding is needed at once; the discriminator's row pass waits for the scene CNN's
                #  branch where it first reads the scene features -- MGGAN_G_SCENE_JOIN=early: before the pass, as until round 5)
                ev = getattr(ctx_d[0], "_mggan_ready", None)
                if _G_SCENE_JOIN_EARLY or ev is None:
                    HF.join_branch(ctx_d[1], which=0)
                    HF.join_branch(ctx_d[0], which=lstm_branch)
                else:
                    HF._cur().wait_event(ev)
                    ctx_d[0].record_stream(HF._cur())
                    ctx_d[1].record_stream(HF._cur())
            HF.mark("G.dpass.begin")
            disc_out = self.D(in_xy, in_dxdy, gen_out.abs, gen_out.rel, sub_batches, img=img, mask=loss_mask,
                              context=ctx_d)
        finally:
            for p, f in zip(d_params, flags):
                p.requires_grad_(f)
        HF.mark("G.dpass.end")
        branch_out = None
        if isinstance(disc_out, tuple):
            disc_out, branch_out = disc_out
        label_real, label_fake = self.rng.labels()
        n_rows = self._global(disc_out.numel())
        # phi_3 (abstract_train.py:62-75): 'NS' BCE(d, real), 'LS' (d - real)^2, 'MM' -BCE(d, fake); rows weighted by
        # 1/count(generator) (train.py:92-97); classifier CE with the same weights (train.py:105-111)
        obj = cfg.gan_obj
        mgan = self.gan_type == "mgan"
        rows_g = getattr(self.G, "last_rows", None)
        o = dict(nA=disc_out.numel(), labels=(label_fake if obj == "MM" else label_real, None), norms=(n_rows, 0, n_rows),
                 kind=1 if obj == "LS" else 0, sign_a=-1.0 if obj == "MM" else 1.0, grad_c=float(cfg.clf_loss_weight),
                 weighted_c=True, g=self.G.n_gs, outs=(m[M_ADV:M_ADV + 1], None, m[M_CLF:M_CLF + 1] if mgan else None))
        if rows_g is not None and rows_g.R == gen_idxs.numel() and not self.dist.enabled and HF.pad_dims() is None:
            o["row_gen"], o["seg"] = rows_g.row_gen_pos, rows_g.seg  # counts = segment lengths of the bucketed rows
        else:
            o["row_gen"], o["inv_count"] = self._gen_weights(gen_idxs)
        o["target"] = o["row_gen"]
        logits = branch_out.transpose(0, 1).reshape(-1, branch_out.shape[-1]) if mgan else None
        losses.append(HF.GanLossesFn.apply(disc_out.t().reshape(-1), logits, o))
        grads.append(self._one)
        items.append(("train/gen_loss", M_ADV))
        if mgan:
            items.append(("train/info_mgan_loss", M_CLF))

        HF.mark("G.loss.end")
        self.optimizerG.zero_grad()
        self._issue_d_context("g_bwd")
        self._backward(losses, grads)
        HF.mark("G.bwd.end")
        ex = self.dist.all_reduce_grads(self.G, defer=True)
        self.optimizerG.step(cfg.clipping_threshold_g, zero_grad=self.zero_grads_in_step, exchange=ex)
        HF.mark("G.opt.end")
        self._emit(train_metrics, items)

    def net_chooser_step(self, in_xy, in_dxdy, gt_xy, gt_dxdy, sub_batches, metrics, mask, img):
        cfg = self.config
        if cfg.weighting_target == "none":
            return
        m, g = self._m, self.G.n_gs
        HF.mark("PM.begin")
        self._issue_d_context("pm_begin")
        if _PM_PREFOLD and HF._BR["on"] and not HF._on_branch():
            # the generator step has just updated the decoders: their folded weights are due before this step's rollout, whose
            # first launch would otherwise find the fold (6 us and a queue hop) between the trunk and itself; here it runs on a
            # branch stream beside the trunk (the rollout waits for its event)
            with HF.branch(2):
                HF.prefold(self.G)
        gen_out, net_chooser_weights, _ = self.G(in_xy, in_dxdy, sub_batches, noise=None, all_gen_out=True, img=img,
                                                 num_samples=cfg.num_expectation_samples, mask=mask, need_samples=False)
        n_pm = self._global(net_chooser_weights.shape[0])
        if cfg.weighting_target == "ml":
            loss = HF.PmMlFn.apply(net_chooser_weights, gen_out.abs, gt_xy, cfg.sigma, m[M_PM:M_PM + 1],
                                   m[M_PROBS:M_PROBS + g], n_pm)
        elif cfg.weighting_target == "mgan":
            assert self.gan_type == "mgan"
            # the reference runs D on the real trajectories here (train.py:608-610); with its target softmax over a
            # singleton axis (see PmMganFn) the only trace D's forward leaves is one more update of the BatchNorm
            # running statistics of its scene encoder
            with torch.no_grad():
                self.D.scene_encoder(img[mask] if HF.is_masked(mask) else img)
            loss = HF.PmMganFn.apply(net_chooser_weights, self._pm_reg(), m[M_PM:M_PM + 1], m[M_PROBS:M_PROBS + g], n_pm)
        else:  # 'l2' / 'endpoint' (train.py:616-624,641-647): cross entropy against the closest generator
            T_, E_, _, b_, _ = gen_out.abs.shape
            target = torch.empty(b_, dtype=torch.int32, device=self.device)
            lib.mggan_pm_target(b_, T_, E_, g, 1 if cfg.weighting_target == "endpoint" else 0,
                                gen_out.abs.contiguous().data_ptr(), gt_xy.contiguous().data_ptr(), target.data_ptr(),
                                HF._s())
            probs = torch.softmax(net_chooser_weights.detach(), 1).contiguous()  # logged mean generator probabilities
            lib.mggan_colmean(probs.data_ptr(), b_, g, float(b_) / n_pm, m[M_PROBS:M_PROBS + g].data_ptr(),
                              HF._s())
            loss = HF.CeMeanFn.apply(net_chooser_weights, target, None, m[M_PM:M_PM + 1], n_pm)
        HF.mark("PM.loss.end")
        self.optimizerG.zero_grad()
        self._issue_d_context("pm_bwd")
        self._backward([loss], [self._w["pi"]], at_main_end=lambda: self._issue_d_context("pm_tail"))
        HF.mark("PM.bwd.end")
        ex = self.dist.all_reduce_grads(self.G, defer=True)
        self.optimizerG.step(0.0, zero_grad=self.zero_grads_in_step, exchange=ex)
        HF.mark("PM.opt.end")
        items = [("probs/Gen {} probability".format(i), M_PROBS + i) for i in range(g)]
        self._emit(metrics, items + [("train/net_chooser_loss", M_PM)])

    # ---- prediction / evaluation -----------------------------------------------------------
    def predict(self, in_dxdy, in_xy, sub_batches, img=None, num=20, noise=None, mask=None):
        """-> (abs (pred_len,num,b,2), rel, probs (b,g) numpy, gen_idxs (b,num) numpy)   (train.py:259-289)."""
        self.G.eval()
        with torch.no_grad():
            preds, net_chooser_out, gen_idxs = self.G(in_xy, in_dxdy, sub_batches, noise=noise, all_gen_out=False,
                                                      img=img, num_samples=num, mask=mask)
            probs = torch.softmax(net_chooser_out, 1)
        assert preds.abs.shape[1] == num
        return preds.abs, preds.rel, to_numpy(probs), to_numpy(gen_idxs)

    # The strategies below run every generator (`all_gen_out=True`, kernels of the hot path in eval mode) and
    # then pick `num` of the (noise sample, generator) pairs per pedestrian; the picking rules are host logic in
    # mggan/utils.py, the picking itself is one gather on the device.
    def _all_generators(self, in_dxdy, in_xy, sub_batches, img, n_samples, noise, mask):
        self.G.eval()
        with torch.no_grad():
            preds, logits, gen_idxs = self.G(in_xy, in_dxdy, sub_batches, noise=noise, all_gen_out=True, img=img,
                                             num_samples=n_samples, mask=mask)
            return preds, torch.softmax(logits, 1), gen_idxs

    @staticmethod
    def _pick(preds, gen, slot):
        """preds.abs/.rel (T, n, g, b, 2); gen/slot (b, num) -> two tensors (T, num, b, 2)."""
        dev = preds.abs.device
        gen, slot = gen.to(dev), slot.to(dev)
        ped = torch.arange(gen.shape[0], device=dev)[:, None]
        return (preds.abs[:, slot, gen, ped].transpose(1, 2).contiguous(),
                preds.rel[:, slot, gen, ped].transpose(1, 2).contiguous())

    def predict_expected(self, in_dxdy, in_xy, sub_batches, img=None, num=20, noise=None, mask=None):
        """Number of predictions per generator proportional to the PM-network probability (train.py:291-352)."""
        preds, probs, _ = self._all_generators(in_dxdy, in_xy, sub_batches, img, num, noise, mask)
        probs = to_numpy(probs)
        gen = torch.from_numpy(expected_sample_idxs(probs, num))
        out_xy, out_dxdy = self._pick(preds, gen, get_selection_indices(gen))
        return out_xy, out_dxdy, probs, to_numpy(gen)

    def predict_uniform(self, in_dxdy, in_xy, sub_batches, img=None, num=20, noise=None, eps=0.0, mask=None):
        """Generators over the probability threshold, best first, in turn (train.py:354-408)."""
        preds, probs, _ = self._all_generators(in_dxdy, in_xy, sub_batches, img, num * self.G.n_gs, noise, mask)
        gen, slot = uniform_sample_idxs(probs, eps, num)
        out_xy, out_dxdy = self._pick(preds, gen, slot)
        return out_xy, out_dxdy, to_numpy(probs), to_numpy(gen)

    def predict_smart_sampling(self, in_dxdy, in_xy, sub_batches, img=None, num=20, noise=None, eps=0.0, mask=None):
        """Uniform sampling among the generators over the threshold (train.py:410-461)."""
        preds, probs, _ = self._all_generators(in_dxdy, in_xy, sub_batches, img, num * self.G.n_gs, noise, mask)
        over = thresholded_generators(probs, eps).float()
        gen = self.rng.sample_generators(torch.log(over), num).cpu()  # Categorical(probs=over).sample((num,)).T
        out_xy, out_dxdy = self._pick(preds, gen, get_selection_indices(gen))
        return out_xy, out_dxdy, to_numpy(probs), to_numpy(gen)

    def predict_rejection(self, in_dxdy, in_xy, sub_batches, img=None, num=20, noise=None, sigma=1e-3, N=10,
                          truncation_ratio=0.7, debug=False, mask=None):
        """Keep the `num` samples with the smallest estimated Jacobian norm |dG/dz| (train.py:463-545,
        'Learning disconnected manifolds: no GAN's land'); single generator only."""
        assert self.config.num_gens == 1, "Only implemented for single generator"
        assert 0.0 < truncation_ratio <= 1.0
        b = in_xy.shape[1]
        total = num + ceil((1 - truncation_ratio) * num)
        if noise is None:
            noise = self.rng.noise(total, self.config.noise_dim, sub_batches, self.device)
        noise = noise.to(self.device)
        preds, probs, gen_idxs = self._all_generators(in_dxdy, in_xy, sub_batches, img, total, noise, mask)
        vec = preds.abs.permute(3, 1, 2, 0, 4).reshape(b, total, -1)
        jac = torch.zeros(b, total, device=self.device)
        for _ in range(N):
            eps_i = self.rng.randn(total, b, self.config.noise_dim).to(self.device) * sigma ** 2
            preds_eps, _, _ = self._all_generators(in_dxdy, in_xy, sub_batches, img, total, noise + eps_i, mask)
            vec_eps = preds_eps.abs.permute(3, 1, 2, 0, 4).reshape(b, total, -1)
            jac += ((vec_eps - vec) ** 2).sum(-1) / sigma ** 2
        keep = torch.sort(jac / N, dim=1)[1][:, :num]
        gen_idxs = gen_idxs.to(self.device)
        if debug:
            gen_idxs = torch.ones_like(gen_idxs)
            gen_idxs[torch.arange(b, device=self.device)[:, None], keep] = 0
            return preds.abs.squeeze(2), preds.rel.squeeze(2), to_numpy(probs), to_numpy(gen_idxs)
        out_xy, out_dxdy = self._pick(preds, torch.zeros_like(keep), keep)
        return out_xy, out_dxdy, to_numpy(probs), to_numpy(gen_idxs[torch.arange(b, device=self.device)[:, None], keep])

    def get_predict_func(self, strategy):
        assert strategy in ("uniform_expected", "sampling", "expected", "rejection", "smart_expected", "smart_sampling",
                            "uniform_sampling")
        if strategy == "expected":
            return self.predict_expected
        if strategy == "rejection":
            return self.predict_rejection
        if strategy == "uniform_expected":
            return self.predict_uniform
        if strategy == "smart_expected":
            return partial(self.predict_uniform, eps=1.0 / self.G.n_gs)
        if strategy == "smart_sampling":
            return partial(self.predict_smart_sampling, eps=1.0 / self.G.n_gs ** 2)
        if strategy == "uniform_sampling":
            return partial(self.predict_smart_sampling, eps=0.0)
        return self.predict

    def get_predictions(self, loader, num_preds=20, strategy="sampling"):
        self.D.eval()
        self.G.eval()
        pred_func = self.get_predict_func(strategy)
        all_preds = []
        for batch in loader:
            batch = self.to_device(batch)
            in_xy, in_dxdy = batch["in_xy"], batch["in_dxdy"]
            b = in_dxdy.size(1)
            sub_batches = batch["seq_start_end"] if "seq_start_end" in batch else list(zip(range(b), range(1, b + 1)))
            preds, _, _, _ = pred_func(in_dxdy, in_xy, sub_batches, img=batch.get("features"), num=num_preds)
            all_preds.append(to_numpy(preds))
        return np.concatenate(all_preds, 2)

    def check_accuracy(self, loader, vis=False, prefix="", num_k=20, predict_strategy="sampling", debug=False, **kw):
        preds = self.get_predictions(loader, num_preds=num_k, strategy=predict_strategy)
        ds = getattr(loader, "dataset", None)
        ds = getattr(ds, "ds", ds)  # DeviceCropDataset wraps the on-disk dataset
        if ds is not None and hasattr(ds, "pred_traj") and hasattr(ds, "seq_start_end"):
            # the reference's path (train.py:255-257): ground truth, scene bounds and pixel ratios from the dataset
            return evaluate_ade_fde(ds, preds, [num_k])
        gts, sse, off = [], [], 0  # datasets that yield ready-made batches (the built-in synthetic one)
        for batch in loader:
            gts.append(batch["gt_xy"].permute(1, 0, 2))
            sse += [(s + off, e + off) for s, e in batch["seq_start_end"]]
            off += batch["gt_xy"].shape[1]

        class _DS:
            pred_traj = torch.cat(gts, 0)
            seq_start_end = sse
            dataset_name = getattr(loader.dataset, "dataset_name", None)

        return evaluate_ade_fde(_DS, preds, [num_k])

    @staticmethod
    def construct_model(config):
        return construct_model(config)


if __name__ == "__main__":
    args = get_parser().parse_args()
    if args.checkpoint:
        output_dir = Path(args.checkpoint)
        assert output_dir.is_dir()
        model, config = PiNetMultiGeneratorGAN.load_from_path(output_dir)
        config.gpus = True
        config.val_every = 1
    else:
        output_dir = Path(args.log_dir) / args.experiment
        output_dir.mkdir(exist_ok=True, parents=True)
        print(str(output_dir.resolve()))
        logger = Experiment(output_dir.resolve(), name=args.name, debug=args.debug,
                            version=random.randint(10 ** 10, (10 ** 11) - 1))
        G, D = construct_model(config=args)
        logger.argparse(args)
        model = PiNetMultiGeneratorGAN(G, D, args, logger)
        logger.save()
    model.train()
