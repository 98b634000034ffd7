"""Scene-sharded data parallelism for the training steps: one process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

The per-scene work is independent (social attention, min-over-K L2 and noise are per scene),
so each rank takes a contiguous range of scenes; weights and optimizer state are replicated.
What couples the ranks:
  C1  parameter gradients    -> ONE all-reduce(sum) of the flat gradient buffer per optimizer step
  C2  BatchNorm statistics   -> all-reduce of the f64 per-channel sums (forward and backward)
  C3  batch-global counters  -> generator-id counts and loss normalisers
All messages are <= 360 KB: latency-bound, so they are kept to one collective each -- and, on one node, they are
plain kernels over peer-mapped memory (mggan/devcomm.py, csrc/comm.hip) so that the iteration stays one HIP graph.
"""
import os

import torch
import torch.distributed as dist


# the gradient exchange and the conv1 tail's finalize inside the optimizer's launch (csrc/loss_opt.hip: clip_adamw_kernel);
# MGGAN_FUSED_STEP=0: round 5's three launches (all-reduce, finalize, optimizer)
FUSED_STEP = os.environ.get("MGGAN_FUSED_STEP", "1") != "0"


def shard_scenes(seq_start_end, rank, world_size):
    """Contiguous scene range for `rank`, balanced by pedestrian count.
    Returns (scene_slice, ped_start, ped_end, local_seq_start_end)."""
    sse = [(int(s), int(e)) for s, e in seq_start_end]
    total = sum(e - s for s, e in sse)
    bounds, acc, nxt = [0], 0, 1
    for i, (s, e) in enumerate(sse):
        acc += e - s
        while nxt < world_size and acc >= total * nxt / world_size:
            bounds.append(i + 1)
            nxt += 1
    while len(bounds) < world_size:
        bounds.append(len(sse))
    bounds.append(len(sse))
    lo, hi = bounds[rank], max(bounds[rank], bounds[rank + 1])
    if lo >= hi:
        return slice(lo, lo), 0, 0, []
    p0, p1 = sse[lo][0], sse[hi - 1][1]
    return slice(lo, hi), p0, p1, [[s - p0, e - p0] for s, e in sse[lo:hi]]


def shard_batch(batch, rank, world_size):
    """Slice a collated batch (reference schema, trajectories_scene.py:68-78) to this rank's scenes."""
    _, p0, p1, local = shard_scenes(batch["seq_start_end"], rank, world_size)
    out = {"seq_start_end": local}
    for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy"):
        out[k] = batch[k][:, p0:p1].contiguous()
    out["features"] = batch["features"][p0:p1].contiguous()
    return out


class DistContext:
    """Collective hooks used by the modules / trainer.  world_size == 1 -> every hook is the identity."""

    def __init__(self, group=None):
        # MGGAN_FORCE_DIST=1: run the collective hooks with a single rank too (exercises the RCCL code path,
        # including its capture into a HIP graph, on a one-GPU box)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("MGGAN_FORCE_DIST", "0") == "1")
        self.group = group
        self.world_size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    recorder = None  # a SegmentRecorder while an iteration is being captured into HIP graphs
    devcomm = None   # mggan.devcomm.DeviceComm: the collectives as plain kernels over peer-mapped memory
    rccl = None      # mggan.devcomm.RcclComm: ncclAllReduce issued by this package's library on the caller's stream
    #                  (capturable: the sharded iteration stays one graph); used where the peer-mapped kernels are not

    @property
    def stream_safe(self):
        """True when the collectives are launches on the CALLER's stream (peer-mapped kernels or in-graph RCCL): they may
        be issued from branch streams and captured into one graph.  False: torch.distributed between graph segments."""
        return self.devcomm is not None or self.rccl is not None

    @property
    def transport(self):
        return "peer-mapped" if self.devcomm is not None else "rccl-graph" if self.rccl is not None else "rccl-segments"
    n_collectives = 0  # exchanges ISSUED by this rank since reset_count() (whatever the transport; tests assert the
    collective_log = None  # schedule: DESIGN section 6) -- and, when a list, their names in issue order

    def count_collective(self, what):
        self.n_collectives += 1
        if self.collective_log is not None:
            self.collective_log.append(what)

    def reset_count(self, log=False):
        self.n_collectives = 0
        self.collective_log = [] if log else None

    def _collective(self, t, tail=None, what="all_reduce"):
        """Sum `t` over the ranks in place -- and, in the SAME exchange, an optional f64 `tail` (the per-rank sums that ride
        with a gradient buffer).  Preferred: the peer-mapped kernel (mggan/devcomm.py) -- an ordinary launch on the
        current stream, capturable, so the sharded iteration stays ONE graph.  Otherwise torch.distributed: while an
        iteration is being captured that collective CUTS the graph (the kernels queued so far become one graph segment,
        the collective stays an eager call replayed between the segments); the tail is then a second call (counted)."""
        self.count_collective(what)
        for comm in (self.devcomm, self.rccl):
            if comm is not None and comm.supports(t, tail):
                comm.all_reduce_(t, tail)
                return
        if tail is not None:
            self.count_collective(what + " (tail: second call)")
        group = self.group

        def exchange():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            if tail is not None:
                # (a second call: packing vector and tail into one f64 buffer needs allocations between the replays of the
                #  graph segments, and those made the segment replay fault at 8,192 pedestrians -- "write access to a
                #  read-only page", every run; the calls below touch only memory that existed when the capture ended)
                dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=group)

        run = exchange

        if self.recorder is not None:
            self.recorder.cut(run)
        else:
            run()

    def all_reduce_(self, t, what="all_reduce"):
        if self.enabled:
            self._collective(t, what=what)
        return t

    def all_reduce_stats(self, sums, n_local):
        """BatchNorm: sums (2C,) f64 and the local image count -> global sums in place, global count returned."""
        if not self.enabled:
            return n_local
        if self.equal_shards:  # every rank holds n_local images: no count exchange, no host read-back
            self._collective(sums, what="bn.stats")
            return float(n_local) * self.world_size
        if self.recorder is not None or (sums.is_cuda and torch.cuda.is_current_stream_capturing()):
            raise RuntimeError("graph capture of a sharded iteration needs equal shards (the global image count of "
                               "unequal shards is read back to the host)")
        buf = torch.cat([sums, sums.new_tensor([float(n_local)])])
        self.count_collective("bn.stats")
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        sums.copy_(buf[:-1])
        return float(buf[-1].item())

    def global_count(self, n_local):
        if not self.enabled:
            return n_local
        if self.equal_shards:
            return n_local * self.world_size
        if self.recorder is not None:
            raise RuntimeError("graph capture of a sharded iteration needs equal shards")
        t = torch.tensor([float(n_local)], dtype=torch.float64, device=self._dev)
        self.count_collective("count")
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    equal_shards = False  # set True when every rank is known to hold the same number of rows (no sync needed)

    _dev = "cpu"

    def all_reduce_grads(self, root, defer=False):
        """C1: one collective over the whole flat gradient buffer -- and, riding in it, the f64 tail a sharded scene-CNN
        backward pass of this step left on the root (mggan/hip/functions.py, SceneAttentionFn.backward: this rank's raw
        conv1 weight-gradient and BatchNorm-1 adjoint sums); its finalize runs right behind the exchange, identically on
        every rank, and adds the global-batch dW1 / dgamma1 / dbeta1 into slots that held zeros during the exchange.
        defer=True (the trainer): -> (comm, tail descriptor, keep-alive) for FlatAdamW.step(exchange=...) when exchange and
        finalize can run INSIDE the optimizer's launch (peer-mapped arenas: both; RCCL inside the graph: the finalize) --
        one launch per optimizer step instead of three; None when everything has been done here."""
        if not self.enabled:
            return None
        from mggan.hip.functions import join_side_stream

        join_side_stream()
        tails = root.__dict__.pop("_grad_tails", [])
        fuse = defer and FUSED_STEP and len(tails) <= 1 and (not tails or len(tails[0]) >= 5)
        if fuse and self.devcomm is not None and self.devcomm.supports(root._flat_grad, tails[0][0] if tails else None) \
                and self.devcomm.chunks_fit(root._flat_grad.numel()):
            self.count_collective("gradients+conv1.tail" if tails else "gradients")
            return (self.devcomm.channel_dev(), tails[0][3] if tails else None, tails[0][4] if tails else None)
        if tails:
            self._collective(root._flat_grad, tails[0][0], what="gradients+conv1.tail")
            # a step with several scene-CNN backward passes on one root (the masked discriminator step runs the
            # encoder once for the real and once for the fake pass): the first tail rides with the gradients, every
            # further one is an exchange of its own; each finalize ADDS its global-batch dW1 / dgamma1 / dbeta1
            for extra in tails[1:]:
                self._collective(extra[0], what="conv1.tail (extra pass)")
            if fuse and self.stream_safe:  # (the sums are global now: the optimizer's launch finalizes)
                return (0, tails[0][3], tails[0][4])
            for t in tails:
                t[1]()
        else:
            self._collective(root._flat_grad, what="gradients")
        return None

    def close(self, rccl=True):
        """Unmap / free the peer-mapped arenas (a process that builds several trainers in a row); rccl: the in-graph RCCL
        communicators too."""
        if self.devcomm is not None:
            self.devcomm.close()
            self.devcomm = None
        if rccl and self.rccl is not None:
            self.rccl.close()
            self.rccl = None

    def attach(self, *roots, bn_sync="global"):
        """bn_sync 'global': the scene CNNs' BatchNorm statistics are all-reduced (14 of the ~18 collectives of an
        iteration; results equal a single process on the whole batch); 'local': per-rank statistics, no exchange."""
        for r in roots:
            for m in r.modules():
                if hasattr(m, "sync") and m.__class__.__name__ == "AttentionGlobal":
                    m.sync = self if (self.enabled and bn_sync == "global") else None
            if r._flat is not None:
                self._dev = r._flat.device
        if self.enabled and self.devcomm is None and torch.device(self._dev).type == "cuda":
            from mggan import devcomm

            # slot capacity from the largest vector this trainer reduces (the flat gradient buffers; f32 = half an
            # 8-byte element each), rounded up to a power of two; every rank builds the same models
            # (+ 1,024 doubles: the f64 tail that rides with a gradient buffer, 608 at C = 16, behind a 256-byte boundary)
            need = max([(r._flat_grad.numel() + 1) // 2 + 1024 for r in roots if getattr(r, "_flat_grad", None) is not None] + [1])
            cap = 1 << 16
            while cap < need:
                cap <<= 1
            self.devcomm = devcomm.create(self.group, self._dev, cap)
        if self.enabled and self.rccl is None and torch.device(self._dev).type == "cuda" and (
                self.devcomm is None or os.environ.get("MGGAN_RCCL_GRAPH", "1") == "2"):
            # no peer-mapped arenas (several nodes, IPC refused, MGGAN_DEVICE_COMM=0): RCCL, issued by this package's
            # library on the caller's stream -- still ONE graph per iteration.  (MGGAN_RCCL_GRAPH=2: build it beside the
            # peer-mapped kernels as well, as the fallback of a trainer whose gradient buffers outgrow an arena slot.)
            from mggan import devcomm

            self.rccl = devcomm.create_rccl(self.group, self._dev)

    def graph_safe(self, *roots):
        """Can every collective of an iteration run INSIDE one captured graph?  Needs the peer-mapped kernels, equal
        shards (unequal ones read the global row count back to the host) and slots that hold the flat gradient buffers;
        otherwise capture_iteration uses graph segments with the collectives between them."""
        if not self.stream_safe or not self.equal_shards:
            return False
        if self.devcomm is None:
            return True  # in-graph RCCL takes any length
        tail = torch.empty(608, dtype=torch.float64, device=self._dev)
        return all(r._flat_grad is None or self.devcomm.supports(r._flat_grad, tail) or
                   (self.rccl is not None and self.rccl.supports(r._flat_grad, tail)) for r in roots)

    def check(self, sync=False):
        """Raise if a peer-mapped collective has timed out (sync=False: a host read, free; the training loop calls it every
        iteration -- a lost peer leaves NaN in the reduced buffers and must stop the run, not train on)."""
        if self.devcomm is not None:
            self.devcomm.check(sync=sync)
        if self.rccl is not None:
            self.rccl.check(sync=sync)

    def host_barrier(self):
        """Line the ranks up on the host (after per-rank host phases -- data-loader construction, validation, checkpoint
        writes -- the ranks can be seconds apart; the in-kernel waits of the first collective afterwards are bounded)."""
        if self.enabled and self.world_size > 1:
            dist.barrier(group=self.group)


def replicas_in_sync(*roots, group=None):
    """Data-parallel invariant: after any number of optimizer steps every rank holds the same weights.
    (max - min over the ranks of each flat parameter buffer's checksum must be exactly zero.)"""
    sums = torch.stack([r._flat.double().sum() for r in roots] + [r._flat.double().abs().sum() for r in roots])
    hi, lo = sums.clone(), sums.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return bool(torch.isfinite(hi).all()) and bool((hi == lo).all())


class SegmentRecorder:
    """Captures one iteration as a chain  graph_0, call_0, graph_1, call_1, ..., graph_n  where the calls are
    the eager collectives that cut the capture (DistContext._collective).  All segments share one memory pool
    and are replayed strictly in capture order, so a tensor produced in one segment is valid in the next.
    Capture mode "relaxed": a cut can come from the autograd engine's worker thread (BatchNorm statistics in a
    backward pass), and HIP only lets another thread end a capture that was begun relaxed."""

    def __init__(self, error_mode="relaxed"):
        self.items = []  # ("graph", CUDAGraph) | ("call", fn)
        self.pool = torch.cuda.graph_pool_handle()
        self.error_mode = error_mode
        self._g = None
        self.origin = None  # the stream every segment is captured on; branch streams fork from / join into it

    def begin(self):
        if self.origin is None:
            self.origin = torch.cuda.current_stream()
        with torch.cuda.stream(self.origin):
            self._g = torch.cuda.CUDAGraph()
            self._g.capture_begin(pool=self.pool, capture_error_mode=self.error_mode)
        cur = torch.cuda.current_stream()
        if cur != self.origin:  # the cut came from inside a branch: the branch stream forks into the new segment
            cur.wait_stream(self.origin)

    def end(self):
        from mggan.hip.functions import branch_streams

        cur = torch.cuda.current_stream()
        for s in list(branch_streams()) + [cur]:  # a capture can only end with every fork joined
            if s != self.origin:
                self.origin.wait_stream(s)
        with torch.cuda.stream(self.origin):
            self._g.capture_end()
        self.items.append(("graph", self._g))
        self._g = None

    def cut(self, fn):
        self.end()
        fn()  # keeps the ranks' collective sequence in lock-step during capture (the data is not meaningful yet)
        self.items.append(("call", fn))
        self.begin()

    def replay(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()

    @property
    def n_graphs(self):
        return sum(1 for k, _ in self.items if k == "graph")
