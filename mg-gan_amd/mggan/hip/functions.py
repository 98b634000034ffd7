"""torch.autograd.Function wrappers over the C ABI (include/mggan_hip.h).

Each Function is one coarse stage of the reference forward pass; its forward and
backward are sequences of HIP kernel launches on the current stream.  PyTorch
tensors are storage only.  Parameter gradients are accumulated by the kernels
into the root module's flat gradient buffer (hip/flat.py), so backward returns
None for parameter inputs.
"""
import contextlib
import ctypes
import json
import os

import torch
from torch.autograd import Function

from .lib import lib, load as _load_lib
from .flat import root_of

ACT_NONE, ACT_LEAKY, ACT_SIGMOID, ACT_SIGMOID_EPS = 0, 1, 2, 3
F32 = torch.float32


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_STREAM_OBJS = {}


def _s():
    """Raw handle of torch's current stream (torch.cuda.current_stream() builds a Stream object per call: 10 us of host
    time, a hundred times per eager iteration)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(_GET_DEVICE() if _GET_DEVICE is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _cur():
    """torch.cuda.current_stream() without a new Stream object per call: the objects are kept by (device, raw handle)
    (torch's streams live in a static pool: a handle never changes its meaning).  60 calls per eager iteration."""
    if _RAW_STREAM is None or _GET_DEVICE is None:
        return torch.cuda.current_stream()
    dev = _GET_DEVICE()
    key = (dev, _RAW_STREAM(dev))
    obj = _STREAM_OBJS.get(key)
    if obj is None:
        obj = _STREAM_OBJS[key] = torch.cuda.current_stream()
    return obj


def _p(t):
    return 0 if t is None else t.data_ptr()


# Debugging aid (MGGAN_POISON=1 or poison_scratch(True)): scratch buffers start as NaN instead of whatever the allocator
# hands back, so a kernel that reads what no kernel wrote shows up as NaN instead of as a box-dependent flake.
_DEBUG = {"poison": os.environ.get("MGGAN_POISON", "0") == "1", "wgrad_dump": os.environ.get("MGGAN_WGRAD_DUMP", "0") == "1",
          "reduce_dump": os.environ.get("MGGAN_REDUCE_DUMP", "0") == "1"}


# Padded batches (train()'s shape buckets, abstract_train.IterationGraphs): the record {n_real, s_real, b_pad / n_real}
# in device memory that the row-mixing kernels read (csrc/common.h); None = the batch at hand is not padded.
_PAD = {"dims": None, "b": 0}


def set_pad_dims(dims, b_pad=0):
    """dims: int32 device tensor of 4 words (or None).  The trainer sets it for the duration of an iteration."""
    was = (_PAD["dims"], _PAD["b"])
    _PAD["dims"], _PAD["b"] = dims, int(b_pad)
    return was


def pad_dims():
    return _PAD["dims"]


def _pad_ptr(t=None):
    t = _PAD["dims"] if t is None else t
    return t.data_ptr() if t is not None else 0


# torch.cuda.Stream() hands out the streams of a pool of 32 per device round-robin: the 33rd object ALIASES the first.
# A trainer that made two streams of its own per capture (the capture stream, the backward stream) reached that after 16
# captures in one process, the capture stream then WAS one of the branch streams, and hipStreamEndCapture segfaulted
# (a stream forked from itself).  Every stream role is therefore created once per process and device, here.
_ROLE_STREAMS = {}


def role_stream(role):
    key = (role, torch.cuda.current_device())
    st = _ROLE_STREAMS.get(key)
    if st is None:
        taken = {s.cuda_stream for s in _ROLE_STREAMS.values()}
        for _ in range(64):  # (a pool slot another role already holds is skipped)
            st = torch.cuda.Stream()
            if st.cuda_stream not in taken:
                break
        else:
            raise RuntimeError("no free HIP stream for role {!r}".format(role))
        _ROLE_STREAMS[key] = st
    return st


class ReplayAlloc:
    """with ReplayAlloc(store): the FIRST time (empty store) every tensor torch.empty / zeros / full / ones / empty_like /
    zeros_like hands out inside the block is recorded; every later time the same tensors are handed out again, in call
    order.  A forward pass that runs under it therefore writes its outputs and everything it saves for its backward pass to
    the same device addresses every time -- what lets work that a captured graph reads at its START be produced at the END
    of the previous replay (the next iteration's discriminator context, mggan/model/train.py) while Python still builds a
    fresh autograd graph over those tensors in every eager iteration.  A sequence that diverges from the recorded one
    (another shape, dtype or count) raises: the caller falls back to the in-order schedule."""
    NAMES = ("empty", "zeros", "full", "ones", "empty_like", "zeros_like")

    def __init__(self, store):
        self.store = store  # {"tensors": [...], "specs": [...], "done": bool}

    def __enter__(self):
        st = self.store
        st.setdefault("tensors", [])
        st.setdefault("specs", [])
        self.pos, self.replay = 0, bool(st.get("done"))
        self.saved = {n: getattr(torch, n) for n in self.NAMES}

        def wrap(name, fn):
            def alloc(*a, **k):
                if self.replay:
                    if self.pos >= len(st["tensors"]):
                        raise RuntimeError("ReplayAlloc: more allocations than recorded ({})".format(len(st["tensors"])))
                    t, spec = st["tensors"][self.pos], st["specs"][self.pos]
                    self.pos += 1
                    now = (name,) + _alloc_spec(name, a, k)
                    if spec != now:
                        raise RuntimeError("ReplayAlloc: allocation {} is {} now, recorded {}".format(self.pos - 1, now, spec))
                    if name in ("zeros", "zeros_like"):
                        t.zero_()
                    elif name == "ones":
                        t.fill_(1)
                    elif name == "full":
                        t.fill_(a[1] if len(a) > 1 else k["fill_value"])
                    return t
                with _Unpatched(self):
                    t = fn(*a, **k)
                st["tensors"].append(t)
                st["specs"].append((name, tuple(t.shape), t.dtype))
                return t

            return alloc

        for n, fn in self.saved.items():
            setattr(torch, n, wrap(n, fn))
        return self

    def __exit__(self, et, ev, tb):
        for n, fn in self.saved.items():
            setattr(torch, n, fn)
        if et is None:
            if self.replay and self.pos != len(self.store["tensors"]):
                raise RuntimeError("ReplayAlloc: {} allocations now, {} recorded".format(self.pos, len(self.store["tensors"])))
            self.store["done"] = True
        return False


def _alloc_spec(name, a, k):
    """(shape, dtype) a torch allocation call asks for, from its arguments."""
    if name in ("empty_like", "zeros_like"):
        return tuple(a[0].shape), k.get("dtype", a[0].dtype)
    sh = a[0] if name == "full" else (a[0] if (len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size))) else a)
    return tuple(int(x) for x in sh), k.get("dtype", torch.float32)


class _Unpatched:
    def __init__(self, ra):
        self.ra = ra

    def __enter__(self):
        self.cur = {n: getattr(torch, n) for n in self.ra.NAMES}
        for n, fn in self.ra.saved.items():
            setattr(torch, n, fn)

    def __exit__(self, *exc):
        for n, fn in self.cur.items():
            setattr(torch, n, fn)
        return False


class _CopyDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_long)]


def copy_small(pairs):
    """[(src tensor, dst tensor)] (<= 8, contiguous, equal byte sizes) copied in ONE launch on the current stream."""
    arr = (_CopyDesc * len(pairs))(*[_CopyDesc(a.data_ptr(), b.data_ptr(), a.numel() * a.element_size()) for a, b in pairs])
    lib.mggan_copy_small(ctypes.addressof(arr), len(pairs), _s())


def poison_scratch(on=True):
    _DEBUG["poison"] = bool(on)


def _empty(*shape, like=None, dtype=F32):
    if _DEBUG["poison"] and dtype == F32:
        return torch.full(shape if not (len(shape) == 1 and isinstance(shape[0], (tuple, list))) else shape[0],
                          float("nan"), dtype=dtype, device=like.device)
    return torch.empty(*shape, dtype=dtype, device=like.device)


def _rows2d(t):
    """(rows, cols) view with unit inner stride -> (tensor, ld)."""
    assert t.dim() == 2 and t.dtype == F32 and t.is_cuda, (t.shape, t.dtype, t.device)
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1])


# Time marks (MGGAN_MARKS=1): a one-lane kernel stores the device clock at the point of the call, on whatever
# stream is current -- also inside a captured graph, where each replay refreshes the slots.
_MARKS = {"on": os.environ.get("MGGAN_MARKS", "0") == "1", "buf": None, "names": []}


def mark(name):
    if not _MARKS["on"]:
        return
    if _MARKS["buf"] is None:
        _MARKS["buf"] = torch.zeros(512, dtype=torch.int64, device="cuda")
    i = len(_MARKS["names"])
    _MARKS["names"].append(name)
    lib.mggan_timestamp(_MARKS["buf"].data_ptr() + 8 * i, _s())


def read_marks():
    """-> [(name, microseconds since the first mark)] in time order."""
    if _MARKS["buf"] is None:
        return []
    t = _MARKS["buf"][:len(_MARKS["names"])].cpu().tolist()
    t0 = min(t)
    return sorted(((n, (v - t0) / 100.0) for n, v in zip(_MARKS["names"], t)), key=lambda x: x[1])


def want_grad(*tensors):
    """Evaluated at the call site (autograd.Function.forward always runs with grad mode off)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# Weight-gradient GEMMs are off the critical path of a backward pass (nothing consumes them before the
# optimizer step).  When enabled (the trainer does), they are issued on a side stream so that they overlap
# the latency-bound input-gradient chain; `join_side_stream()` (called before clip+AdamW / gradient
# all-reduce) makes the main stream wait for them.  Inside a HIP-graph capture this becomes a parallel branch.
_SIDE = {"on": False, "stream": None, "keep": [], "dirty": False}


def enable_side_stream(on=True):
    _SIDE["on"] = bool(on)


def _side_stream():
    if _SIDE["stream"] is None:
        _SIDE["stream"] = role_stream("wgrad-side")
    return _SIDE["stream"]


class side_stream:
    """Context: kernels launched inside run on the side stream, ordered after everything already queued on
    the main stream.  Tensors passed in `keep` stay referenced until join_side_stream()."""

    def __init__(self, *keep):
        self.keep = keep

    def __enter__(self):
        if not _SIDE["on"]:
            self.ctx = None
            return self
        side = _side_stream()
        side.wait_stream(_cur())
        _SIDE["keep"].extend(self.keep)
        _SIDE["dirty"] = True
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def join_side_stream():
    if _SIDE["dirty"]:
        _cur().wait_stream(_side_stream())
        _SIDE["keep"].clear()
        _SIDE["dirty"] = False


# Independent sub-graphs of a step (the scene CNN next to the trajectory LSTM / social attention chain; the
# discriminator's history context next to the generator rollouts) are issued on a second stream: the small
# latency-bound kernels of one branch fill the gaps of the other.  Autograd replays every backward node on
# the stream of its forward, so the backward pass inherits the same two-branch shape; inside a HIP-graph
# capture the fork/join events become graph edges.  A branch ALWAYS starts by waiting for the main stream, and
# every step ends joined, so memory freed by one stream is never re-used by the other before it is ordered.
# MGGAN_BRANCH = 1 | 0 | auto (default): auto = the trainer turns them on for batches of up to MGGAN_BRANCH_MAX_B
# pedestrians.  Small batches are chains of latency-bound launches that leave most of the chip idle (64 x 20 pedestrians:
# 1.67 ms with branches, 1.94 ms without); at 256 x 32 every kernel fills the chip by itself and concurrent kernels only
# take each other's CUs and cache (6.66 ms with branches, 6.45 ms without; eight hardware queues instead of four: 7.76 ms).
# Measured in between (branches on / off): 2,560 pedestrians 2.59 / 2.74 ms, 4,096: 3.73 / 3.81, 6,144: 5.07 / 5.18.
# Re-measured after the round-3 kernels (each of which leaves more of the chip free): 8,192 pedestrians 5.29 / 5.37 ms,
# 12,288: 7.70 / 7.59, 16,384: 9.92 / 9.64 -- the crossover moved up, the default threshold with it (6,144 -> 8,192).
_BR = {"on": os.environ.get("MGGAN_BRANCH", "auto") != "0", "streams": {}, "dirty": set(), "raw": {}, "hold": set(),
       "auto": os.environ.get("MGGAN_BRANCH", "auto") == "auto", "max_b": int(os.environ.get("MGGAN_BRANCH_MAX_B", "8192"))}


# Eager iterations are bound by the HOST (3.5-5 ms of Python per iteration against 0.8-4.5 ms of kernels), and every fork / join
# of a branch stream is host work (an event record and a wait, ~13 us each, ~70 per iteration): eager launches on ONE stream
# are faster at every size (round 6, ms per iteration with branches -> without: 32 ragged scenes 4.50 -> 3.54, --rng host 7.09
# -> 6.21; 64 x 20: 4.76 -> 3.85; 256 x 32: 5.60 -> 4.99).  The branch streams pay inside a captured graph, where they are edges
# and cost nothing to issue: auto = on for captures (and the warm-up iterations in front of one), off for eager iterations.
# Same kernels, fixed-order reductions: the results do not depend on it (tests/test_hip_graph.py).  MGGAN_BRANCH_EAGER=1: as
# until round 5.
_BRANCH_EAGER = os.environ.get("MGGAN_BRANCH_EAGER", "0") == "1"


def auto_branches(b, for_graph=True):
    """Called by the trainer at the start of an iteration over b pedestrians (every branch is joined there); for_graph:
    the iteration is being captured, warms a capture up, or is a sharded one (whose channels are per stream role)."""
    if _BR["auto"]:
        want = b <= _BR["max_b"] and (for_graph or _BRANCH_EAGER)
        if want != _BR["on"]:
            enable_branches(want)


def _on_branch():
    """Is torch's current stream one of the branch streams?  (raw-handle lookup: no Stream object per call)"""
    return bool(_BR["raw"]) and _s() in _BR["raw"]


def enable_branches(on=True):
    join_branch()
    _BR["on"] = bool(on)


def branch_streams():
    return _BR["streams"].values()


class branch:
    """Context: launches inside go to branch stream `which` (no-op when disabled or already on a branch)."""

    def __init__(self, which=0):
        self.which = which

    def __enter__(self):
        self.ctx = None
        if not _BR["on"]:
            return self
        side = _BR["streams"].get(self.which)
        if side is None:
            side = _BR["streams"][self.which] = role_stream(("branch", self.which))
            _BR["raw"][side.cuda_stream] = self.which
        if _on_branch():
            return self
        cur = _cur()
        side.wait_stream(cur)
        _BR["dirty"].add(self.which)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class HandoffFn(Function):
    """Identity on tensors computed on another stream: a node of the CONSUMING stream.  Autograd runs a node's adjoint on the
    stream of its forward; gradients that reach a value from several streams (the min-L2 branch and the discriminator pass on
    the generator's predictions) then meet HERE, on the consumer's stream, and travel on to the producer's stream as one --
    a captured stream that waits for two foreign streams at one point crashes hipStreamEndCapture (ROCm 7.0)."""

    @staticmethod
    def forward(ctx, *xs):
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        return gs


def hold_branch(which, on=True):
    """A held branch is skipped by the join-everything calls at the end of a backward pass / a step: work forked in one step
    and consumed in the next (the generator step's forward pass beside the discriminator step's backward pass) stays in
    flight until join_branch(..., which=<it>) names it."""
    (_BR["hold"].add if on else _BR["hold"].discard)(which)


def join_branch(*tensors, which=None, force=False):
    """The current stream waits for branch stream `which` (None: all of them); `tensors` (branch results consumed
    from here on) are registered with the consuming stream so the allocator keeps them until it is done."""
    cur = None
    for w, side in list(_BR["streams"].items()):
        if which is not None and w != which:
            continue
        if which is None and w in _BR["hold"]:
            continue  # (a branch that outlives the step it was forked in: joined by name only, hold_branch)
        if not (w in _BR["dirty"] or force):
            continue
        if cur is None:
            cur = _cur()
        if cur == side:
            continue
        cur.wait_stream(side)
        _BR["dirty"].discard(w)
    if tensors:
        if cur is None:
            cur = _cur()
        for t in tensors:
            if t is not None and torch.is_tensor(t):
                t.record_stream(cur)


class _ReduceDesc(ctypes.Structure):
    _fields_ = [("P", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("w_stride", ctypes.c_long),
                ("b_stride", ctypes.c_long), ("M", ctypes.c_int), ("Naug", ctypes.c_int), ("has_bias", ctypes.c_int),
                ("lddw", ctypes.c_int), ("splits", ctypes.c_int), ("groups", ctypes.c_int), ("p_stride", ctypes.c_int),
                ("block0", ctypes.c_int)]


# Deferred gradient reduction: inside the trainer's backward every weight-gradient kernel only leaves its
# partial sums behind; ONE batched launch per optimizer step folds them all into the flat gradient buffer.
class _WgradDesc(ctypes.Structure):
    _fields_ = [("dZ", ctypes.c_void_p), ("X", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("seg", ctypes.c_void_p),
                ("rows", ctypes.c_int), ("K", ctypes.c_int), ("N", ctypes.c_int), ("lddz", ctypes.c_int),
                ("ldx", ctypes.c_int), ("seg_scale", ctypes.c_int), ("n_groups", ctypes.c_int),
                ("feature_major", ctypes.c_int)]


_DEFER = {"on": False, "descs": [], "keep": [], "gemms": [], "events": {}, "revents": {}, "after": []}
TRACE_NOTES = {"wgrad_multi_flops": [], "wgrad_multi_bytes": [], "mlp_chain_flops": [], "social_rows_fwd_pairs": [],
               "social_rows_bwd_pairs": []}


def defer_grad_reduce(on=True):
    _DEFER["on"] = bool(on)


def reset_deferred():
    """Forget everything queued for later launches (a capture failed half-way through a backward pass: the queued
    reductions / weight-gradient products point at operands of the aborted iteration and must never be folded into the
    gradients of a later one)."""
    _DEFER.update(descs=[], keep=[], gemms=[], events={}, revents={}, after=[])
    _SIDE["keep"].clear()
    _SIDE["dirty"] = False


def _note_branch_partials():
    """Partial sums queued for the batched reduction were (or are about to be) written by a kernel on the CURRENT stream.
    On a branch stream that point is remembered as an event, so that the reduction can go out as soon as every producer
    is done - beside what the branch still has to do (the scene CNN's conv1 adjoint) - instead of behind a full join."""
    if _on_branch():
        cur = _cur()
        ev = torch.cuda.Event()
        ev.record(cur)
        _DEFER["revents"][cur.cuda_stream] = ev


def _queue_reduce(P, dW, db, M, Naug, has_bias, lddw, splits, groups, p_stride, w_stride=0, b_stride=0, keep=()):
    _note_branch_partials()
    _DEFER["descs"].append(_ReduceDesc(P, dW, db or None, w_stride, b_stride, M, Naug, has_bias, lddw, splits, groups,
                                       p_stride, 0))
    _DEFER["keep"].extend(keep)


def flush_wgrad_gemms():
    """Launch the queued weight-gradient GEMMs (partial sums; one launch per operand layout and 16 problems) on
    the current stream.  Called BEFORE the branch streams are joined: the GEMMs need the branch streams only up to
    the points where those queued their operands (events recorded then), so they run beside the rest of a branch's
    backward (the scene CNN's convolution adjoints) instead of after it.  (Launching them in earlier batches on a stream
    of their own, beside the backward pass that is still queueing operands, was measured and is worse: the GEMM grids
    take the CUs the latency-bound main chain needs - 2.17-2.34 ms per configs[1] iteration against 2.08.)"""
    gm = _DEFER["gemms"]
    if not gm:
        return
    cur = _cur()
    for ev in _DEFER["events"].values():
        cur.wait_event(ev)
    _DEFER["events"] = {}
    arr = (_WgradDesc * len(gm))(*gm)
    if _DEBUG.get("wgrad_dump"):  # MGGAN_WGRAD_DUMP=1: what a step's weight-gradient batch is made of
        print("[wgrad batch] " + "; ".join("{}x({}x{}){}".format(g.rows, g.N, g.K, " fm" if g.feature_major else "")
                                           for g in gm) + " | MB {:.1f}".format(sum(4e-6 * g.rows * (g.K + g.N) for g in gm)))
    if _load_lib().trace is not None:  # bench.py's per-entry trace: algorithmic FLOPs of this batch
        TRACE_NOTES["wgrad_multi_flops"].append(sum(2.0 * g.rows * g.K * g.N for g in gm))
        TRACE_NOTES["wgrad_multi_bytes"].append(sum(4.0 * g.rows * (g.K + g.N) for g in gm))  # each operand once
    lib.mggan_wgrad_multi(ctypes.addressof(arr), len(gm), _s())
    _DEFER["gemms"] = []


def _note_gemm_operands():
    """The operands of a GEMM queued from a branch stream are complete on that stream from here on."""
    if _on_branch():
        cur = _cur()
        ev = torch.cuda.Event()
        ev.record(cur)
        _DEFER["events"][cur.cuda_stream] = ev


def flush_grad_reduces():
    """Batched reduce.  Two partial buffers that accumulate into the SAME gradient tensor (e.g. the real and
    the fake pass of a discriminator step) must not share a launch: batches are cut at such conflicts."""
    d = _DEFER["descs"]
    if not d:
        return
    flush_wgrad_gemms()
    cur = _cur()
    for ev in _DEFER["revents"].values():
        cur.wait_event(ev)
    _DEFER["revents"] = {}
    batch, seen = [], set()

    def launch():
        if batch:
            arr = (_ReduceDesc * len(batch))(*batch)
            if _DEBUG.get("reduce_dump"):  # MGGAN_REDUCE_DUMP=1: (M, Naug, splits, groups, p_stride) of every partial buffer
                print("[reduce batch] " + json.dumps([[q.M, q.Naug, q.splits, q.groups, q.p_stride] for q in batch])
                      + " | MB {:.1f}".format(sum(4e-6 * q.M * q.Naug * q.splits * q.groups for q in batch)))
            lib.mggan_grad_reduce_multi(ctypes.addressof(arr), len(batch), _s())

    for desc in d:
        keys = {desc.dW} | ({desc.db} if desc.db else set())
        if (keys & seen) or len(batch) == 48:
            launch()
            batch, seen = [], set()
        batch.append(desc)
        seen |= keys
    launch()
    after, _DEFER["after"] = _DEFER["after"], []
    for fn in after:  # consumers of reduced scratch (the LSTM gradient un-folding): behind the batched reductions
        fn()
    _DEFER["descs"], _DEFER["keep"] = [], []


_WGRAD_GEO = {}  # (rows, K, N, groups) -> (workspace bytes, splits): two C calls per weight gradient otherwise


def wgrad(dz, lddz, x, ldx, dW_ptr, lddw, db_ptr, rows, K, N, seg=None, seg_scale=1, n_groups=0, w_stride=0,
          b_stride=0, fm=0, now=False, yact=None, ld_yact=0, act=0, slope=0.0, overwrite=False):
    """dW += dz^T x, db += colsum(dz)  (deterministic split reduction).  Call inside `with side_stream(...)`
    to take it off the critical path."""
    if rows == 0:
        return
    geo = _WGRAD_GEO.get((rows, K, N, n_groups))
    if geo is None:
        geo = _WGRAD_GEO[(rows, K, N, n_groups)] = (lib.mggan_wgrad_workspace_bytes(rows, K, N, n_groups),
                                                    lib.mggan_wgrad_splits(rows, K, N, n_groups))
    nbytes, splits = geo
    ws = _empty(nbytes // 4, like=dz if torch.is_tensor(dz) else x)
    if _SIDE["dirty"]:
        _SIDE["keep"].append(ws)
    defer = _DEFER["on"] and not _SIDE["dirty"] and not now  # now=True: the result is consumed right away
    pz, px = (_p(dz) if torch.is_tensor(dz) else dz), (_p(x) if torch.is_tensor(x) else x)
    if defer and yact is None and overwrite:
        # queued like the rest, but its reduction STORES into dW/db (scratch that a queued consumer reads afterwards)
        _DEFER["gemms"].append(_WgradDesc(pz, px, ws.data_ptr(), _p(seg) or None, rows, K, N, lddz, ldx, seg_scale, n_groups,
                                          fm))
        _note_gemm_operands()
        _DEFER["descs"].append(_ReduceDesc(ws.data_ptr(), dW_ptr, db_ptr or None, w_stride, b_stride, N, K + 1, 3, lddw,
                                           splits, max(n_groups, 1), N * (K + 1), 0))
        _DEFER["keep"].extend((ws, dz, x, seg))
        return
    if defer and yact is None:
        # neither the GEMM nor its reduction runs now: both are queued (operands kept alive) and go out in the
        # batched launches of flush_grad_reduces() at the end of the backward pass
        _DEFER["gemms"].append(_WgradDesc(pz, px, ws.data_ptr(), _p(seg) or None, rows, K, N, lddz, ldx, seg_scale, n_groups,
                                          fm))
        keep = (ws, dz, x, seg)
        _note_gemm_operands()
    elif overwrite:  # partial sums now, then a storing (not accumulating) reduction: the destination is scratch
        lib.mggan_wgrad(pz, lddz, px, ldx, 0, lddw, db_ptr, rows, K, N, _p(seg), seg_scale, n_groups, w_stride, b_stride, fm,
                        _p(yact), ld_yact, act, float(slope), ws.data_ptr(), nbytes, _s())
        one = (_ReduceDesc * 1)(_ReduceDesc(ws.data_ptr(), dW_ptr, db_ptr or None, w_stride, b_stride, N, K + 1, 3, lddw,
                                            splits, max(n_groups, 1), N * (K + 1), 0))
        lib.mggan_grad_reduce_multi(ctypes.addressof(one), 1, _s())
        return
    else:
        lib.mggan_wgrad(pz, lddz, px, ldx, 0 if defer else dW_ptr, lddw, db_ptr, rows, K, N, _p(seg), seg_scale, n_groups,
                        w_stride, b_stride, fm, _p(yact), ld_yact, act, float(slope), ws.data_ptr(), nbytes, _s())
        keep = (ws,)
    if defer:
        ng = max(n_groups, 1)
        _queue_reduce(ws.data_ptr(), dW_ptr, db_ptr, N, K + 1, 1, lddw, splits, ng,
                      N * (K + 1), w_stride, b_stride, keep=keep)


# ------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = act(x W^T + b)   (nn.Linear + activation; reference utils.py:134-149)."""

    @staticmethod
    def forward(ctx, x, W, b, act, slope, owner):
        x, ldx = _rows2d(x)
        rows, K = x.shape
        N = W.shape[0]
        y = _empty(rows, N, like=x)
        lib.mggan_linear_fwd(_p(x), ldx, _p(W), _p(b), _p(y), N, rows, K, N, act, float(slope), _s())
        ctx.act, ctx.slope, ctx.owner, ctx.ldx = act, slope, owner, ldx
        # whether the weights train is decided when the forward pass runs (the generator step freezes D around its
        # forward: the reference discards those gradients, so they are never computed)
        ctx.train_w = W.requires_grad
        ctx.save_for_backward(x, W, b, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, b, y = ctx.saved_tensors
        rows, K = x.shape
        N = W.shape[0]
        dy, lddy = _rows2d(dy)
        # (the C ABI can fuse dY*act'(Y) into the GEMM operand loads, but measured on MI355X the extra operand
        # stream costs ~11 us per GEMM against ~5 us for this elementwise launch, so it stays separate)
        if ctx.act != ACT_NONE:
            dz = _empty(rows, N, like=x)
            lib.mggan_act_bwd(_p(dy), lddy, _p(y), N, _p(dz), N, rows, N, ctx.act, float(ctx.slope), _s())
            lddz = N
        else:
            dz, lddz = dy, lddy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty(rows, K, like=x)
            lib.mggan_linear_bwd_data(_p(dz), lddz, _p(W), K, _p(dx), K, rows, K, N, 0, 0, 0, 0, 0.0, _s())
        if ctx.train_w:
            root = root_of(ctx.owner)
            with side_stream(dz, x):
                wgrad(dz, lddz, x, ctx.ldx, root.grad_ptr(W), K, root.grad_ptr(b) if b is not None else 0, rows, K, N)
        return dx, None, None, None, None, None


def linear(x, layer, act=ACT_NONE, slope=0.0):
    lead = x.shape[:-1]
    y = LinearFn.apply(x.reshape(-1, x.shape[-1]), layer.weight, layer.bias, act, slope, layer)
    return y.reshape(*lead, -1)


class _McStage(ctypes.Structure):  # mirrors csrc/mlp.hip:McStage
    _fields_ = [("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("mul_src", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("K", ctypes.c_int), ("N", ctypes.c_int), ("ldw", ctypes.c_int), ("trans", ctypes.c_int),
                ("act", ctypes.c_int), ("mul_act", ctypes.c_int), ("ld_mul", ctypes.c_int), ("ld_out", ctypes.c_int),
                ("accumulate", ctypes.c_int), ("slope", ctypes.c_float), ("mul_slope", ctypes.c_float)]


class _McArgs(ctypes.Structure):
    _fields_ = [("X", ctypes.c_void_p), ("in_mul", ctypes.c_void_p), ("in_store", ctypes.c_void_p),
                ("ldx", ctypes.c_int), ("rows", ctypes.c_int), ("K0", ctypes.c_int), ("ld_in_mul", ctypes.c_int),
                ("in_mul_act", ctypes.c_int), ("ld_in_store", ctypes.c_int), ("n", ctypes.c_int),
                ("in_mul_slope", ctypes.c_float), ("s", _McStage * 3)]


MLP_MAX_WIDTH = 192


def _chain_fwd(x, ldx, rows, spec, Ws, bs, save, out_into=None):
    """One launch: act_n(... act_1(x W_1^T + b_1) ...) -> per-layer outputs (hidden ones only when `save`).
    x: tensor or raw device pointer.  out_into = (tensor view, row stride): where the LAST layer's output goes
    (a column block of a wider buffer) instead of a fresh tensor."""
    n = len(spec)
    K0 = Ws[0].shape[1]
    a = _McArgs()
    a.X, a.ldx, a.rows, a.K0, a.n = (_p(x) if torch.is_tensor(x) else x), ldx, rows, K0, n
    outs = []
    for i, ((act, slope), W, b) in enumerate(zip(spec, Ws, bs)):
        N, K = W.shape
        keep = i == n - 1 or save  # hidden activations only leave the chip when a backward pass needs them
        ld_y = N
        if i == n - 1 and out_into is not None:
            y, ld_y = out_into
        else:
            y = _empty(rows, N, like=W) if keep else None
        outs.append(y)
        st = a.s[i]
        st.W, st.bias, st.out = _p(W), _p(b), _p(y)
        st.K, st.N, st.ldw, st.trans, st.act, st.slope, st.ld_out = K, N, K, 0, act, float(slope), ld_y
    if _load_lib().trace is not None:
        TRACE_NOTES["mlp_chain_flops"].append(2.0 * rows * sum(W.shape[0] * W.shape[1] for W in Ws))
    lib.mggan_mlp_chain(ctypes.addressof(a), _s())
    return outs


def _chain_bwd(dy, lddy, x, ldx, rows, outs, spec, Ws, bs, need_dx, train_w, owner, dx_into=None, ld_last=None):
    """Adjoint of _chain_fwd in one launch (+ the weight-gradient GEMMs, queued or on the side stream).
    dx_into = (pointer, row stride, accumulate): where the input gradient goes (default: a fresh tensor);
    ld_last: row stride of outs[-1] when it is a column block of a wider buffer."""
    n = len(spec)
    # gate gradients dz_l (l = n-1 .. 0): dz_{n-1} = dy * act'(y); dz_{l-1} = (dz_l W_l) * act'(h_{l-1})
    a = _McArgs()
    a.X, a.ldx, a.rows, a.K0 = _p(dy), lddy, rows, Ws[-1].shape[0]
    dz = [None] * n
    act_last = spec[-1][0]
    if act_last != ACT_NONE:
        a.in_mul, a.ld_in_mul, a.in_mul_act, a.in_mul_slope = (_p(outs[-1]), ld_last or outs[-1].shape[1], act_last,
                                                               float(spec[-1][1]))
        if train_w:
            dz[-1] = _empty(rows, Ws[-1].shape[0], like=Ws[-1])
            a.in_store, a.ld_in_store = _p(dz[-1]), Ws[-1].shape[0]
    elif train_w:
        dz[-1] = dy
    k = 0
    dx = None
    for l in range(n - 1, -1, -1):  # stage k: dz_l (rows, N_l) -> (rows, K_l) through W_l
        if l == 0 and not need_dx:
            break
        N, K = Ws[l].shape
        st = a.s[k]
        st.W, st.K, st.N, st.ldw, st.trans, st.act = _p(Ws[l]), N, K, K, 1, ACT_NONE
        if l > 0:
            st.mul_src, st.ld_mul, st.mul_act, st.mul_slope = _p(outs[l - 1]), K, spec[l - 1][0], float(spec[l - 1][1])
            if train_w:
                dz[l - 1] = _empty(rows, K, like=Ws[l])
                st.out, st.ld_out = _p(dz[l - 1]), K
        elif dx_into is not None:
            st.out, st.ld_out, st.accumulate = dx_into[0], dx_into[1], int(dx_into[2])
        else:
            dx = _empty(rows, K, like=Ws[l])
            st.out, st.ld_out = _p(dx), K
        k += 1
    a.n = k
    if k > 0:
        if _load_lib().trace is not None:
            TRACE_NOTES["mlp_chain_flops"].append(2.0 * rows * sum(a.s[i].K * a.s[i].N for i in range(k)))
        lib.mggan_mlp_chain(ctypes.addressof(a), _s())
    if train_w:
        root = root_of(owner)
        lddz_last = lddy if dz[-1] is dy else Ws[-1].shape[0]
        with side_stream(*[t for t in dz if t is not None], x, *outs):
            for l in range(n):
                N, K = Ws[l].shape
                inp, ldi = (x, ldx) if l == 0 else (outs[l - 1], K)
                wgrad(dz[l], lddz_last if l == n - 1 else N, inp, ldi, root.grad_ptr(Ws[l]), K,
                      root.grad_ptr(bs[l]) if bs[l] is not None else 0, rows, K, N)
    return dx


class MlpChainFn(Function):
    """act_n(... act_1(x W_1^T + b_1) ...) for 2-3 Linear layers in ONE launch per direction
    (nn.Sequential stacks of utils.make_mlp, utils.py:134-149; csrc/mlp.hip)."""

    @staticmethod
    def forward(ctx, x, spec, owner, save, *wb):
        # spec: tuple of (act, slope) per layer; wb = (W_1, b_1, W_2, b_2, ...); save: a backward pass will follow
        x, ldx = _rows2d(x)
        rows = x.shape[0]
        n = len(spec)
        Ws, bs = wb[0::2], wb[1::2]
        outs = _chain_fwd(x, ldx, rows, spec, Ws, bs, save)
        if save:
            ctx.spec, ctx.owner, ctx.ldx, ctx.n, ctx.train_w = spec, owner, ldx, n, Ws[0].requires_grad
            ctx.save_for_backward(x, *outs, *wb)
        return outs[-1]

    @staticmethod
    def backward(ctx, dy):
        n, spec = ctx.n, ctx.spec
        sv = ctx.saved_tensors
        x, outs, wb = sv[0], sv[1:1 + n], sv[1 + n:]
        Ws, bs = wb[0::2], wb[1::2]
        dy, lddy = _rows2d(dy)
        dx = _chain_bwd(dy, lddy, x, ctx.ldx, x.shape[0], outs, spec, Ws, bs, ctx.needs_input_grad[0], ctx.train_w,
                        ctx.owner)
        return (dx, None, None, None) + (None,) * len(wb)


class TwoHeadsFn(Function):
    """Two MLP heads on one input: head A over all rows, head B over rows [row0, rows) (the discriminator's
    real/fake score and its generator-id classifier on the fake half of a pair pass, discriminators.py:186-199).
    As two autograd nodes the heads cost a slice backward (zero-fill + copy) and a gradient add; here head B's
    input gradient accumulates into head A's inside its own launch."""

    @staticmethod
    def forward(ctx, x, row0, spec_a, spec_b, owner_a, owner_b, save, *wb):
        x, ldx = _rows2d(x)
        rows, K = x.shape
        na, nb = len(spec_a), len(spec_b)
        wa, wbb = wb[:2 * na], wb[2 * na:]
        outs_a = _chain_fwd(x, ldx, rows, spec_a, wa[0::2], wa[1::2], save)
        xb = x[row0:]
        outs_b = _chain_fwd(xb, ldx, rows - row0, spec_b, wbb[0::2], wbb[1::2], save)
        if save:
            ctx.cfg = (row0, spec_a, spec_b, owner_a, owner_b, ldx, wa[0].requires_grad, wbb[0].requires_grad)
            ctx.save_for_backward(x, *outs_a, *outs_b, *wb)
        return outs_a[-1], outs_b[-1]

    @staticmethod
    def backward(ctx, dya, dyb):
        row0, spec_a, spec_b, owner_a, owner_b, ldx, train_a, train_b = ctx.cfg
        na, nb = len(spec_a), len(spec_b)
        sv = ctx.saved_tensors
        x, outs_a, outs_b, wb = sv[0], sv[1:1 + na], sv[1 + na:1 + na + nb], sv[1 + na + nb:]
        wa, wbb = wb[:2 * na], wb[2 * na:]
        rows, K = x.shape
        need_dx = ctx.needs_input_grad[0]
        dx = None
        if dya is None:
            dx = torch.zeros(rows, K, dtype=F32, device=x.device) if need_dx else None
        else:
            dya, ld = _rows2d(dya)
            dx = _chain_bwd(dya, ld, x, ldx, rows, outs_a, spec_a, wa[0::2], wa[1::2], need_dx, train_a, owner_a)
        if dyb is not None:
            dyb, ld = _rows2d(dyb)
            into = (dx.data_ptr() + 4 * row0 * K, K, 1) if need_dx else None
            _chain_bwd(dyb, ld, x[row0:], ldx, rows - row0, outs_b, spec_b, wbb[0::2], wbb[1::2], need_dx, train_b,
                       owner_b, dx_into=into)
        return (dx,) + (None,) * (6 + len(wb))


# Above this many rows the stack runs as one GEMM launch per layer.  Kernel for kernel the chain only ties the
# GEMMs at 25,600 rows (35.7 vs 35.2 us forward), but inside the iteration graph every launch it removes is a
# dependent hop less on the critical chain: 2.38 ms/iter with the 25,600-row discriminator pass fused, 2.40 without.
MLP_FUSE_MAX_ROWS = int(os.environ.get("MGGAN_MLP_FUSE_MAX_ROWS", "32768"))


def mlp(x, layers, owner=None):
    """layers: [(nn.Linear, act, slope), ...] applied in order as one fused chain (2-3 layers, widths <= 192)."""
    lead = x.shape[:-1]
    if x.numel() // max(x.shape[-1], 1) > MLP_FUSE_MAX_ROWS:
        for lin, act, slope in layers:
            x = linear(x, lin, act, slope)
        return x
    spec = tuple((act, slope) for _, act, slope in layers)
    wb = []
    for lin, _, _ in layers:
        wb += [lin.weight, lin.bias]
    y = MlpChainFn.apply(x.reshape(-1, x.shape[-1]), spec, owner if owner is not None else layers[0][0],
                         want_grad(x, wb[0]), *wb)
    return y.reshape(*lead, -1)


def steps_to_rows(a, b=None):
    """(T, n, 2) time-major steps [and a second set] -> (n [+ n], 2T) rows, one launch; inputs carry no gradient."""
    T = a.shape[0]
    a = a.reshape(T, -1, 2).contiguous()
    n = a.shape[1]
    out = _empty(2 * n if b is not None else n, 2 * T, like=a)
    lib.mggan_steps_to_rows(_p(a), _p(b.reshape(T, n, 2).contiguous()) if b is not None else 0, T, n, _p(out), _s())
    return out


def two_heads(x, layers_a, layers_b, row0=0):
    """-> (head_a(x), head_b(x[row0:])): one autograd node (TwoHeadsFn) when the fused chain applies."""
    if x.shape[0] > MLP_FUSE_MAX_ROWS:
        return mlp(x, layers_a), mlp(x[row0:] if row0 else x, layers_b)
    spec_a = tuple((act, slope) for _, act, slope in layers_a)
    spec_b = tuple((act, slope) for _, act, slope in layers_b)
    wb = []
    for lin, _, _ in list(layers_a) + list(layers_b):
        wb += [lin.weight, lin.bias]
    return TwoHeadsFn.apply(x, row0, spec_a, spec_b, layers_a[0][0], layers_b[0][0], want_grad(x, *wb[0::2]), *wb)


# ------------------------------------------------------------------------------------------
# Folded LSTM weights (`prep`: W_ih.emb folded into two input coefficients per gate row, transposed W_hh, ...) only change
# when the weights do.  Inside the trainer's iteration (`prep_cache(True)`) a fold is launched once per weight VERSION and
# kept: per iteration the generator's encoder is folded twice (after the generator's and after the PM network's AdamW
# step), its decoders once, the discriminator's encoder once -- four launches instead of seven, none of them in front of
# the discriminator step's rollouts.  A version = the data pointers and torch version counters of the folded tensors
# (load_state_dict, in-place torch ops) + a counter the kernels that write weights through raw pointers bump
# (FlatAdamW.step; a graph replay bumps it for both models: the launches inside a graph are invisible to the host).
# A capture needs no special case: the folds it records are those of an iteration that follows a complete iteration, and
# a replay is only ever preceded by a complete iteration (the warm-up iteration of capture_iteration, or another replay);
# a capture that follows a replay finds every entry stale and records more folds than needed, never fewer.
# Outside the trainer (module calls in tests, predict) every forward folds, as before.
# An entry lives ON its owner module (attribute `_mggan_prep` = [key, buffer, fold]); `owners` only remembers, weakly,
# which modules carry one: a trainer that goes away takes its folded-weight buffers with it.
import weakref  # noqa: E402

_PREP = {"on": False, "owners": weakref.WeakSet(), "enabled": os.environ.get("MGGAN_PREP_CACHE", "1") != "0"}


def prep_cache(on):
    """-> the previous setting.  The entries survive switching it off (the trainer switches it on per iteration)."""
    was, _PREP["on"] = _PREP["on"], bool(on)
    return was


def clear_prep_cache():
    for owner in list(_PREP["owners"]):
        if hasattr(owner, "_mggan_prep"):
            object.__delattr__(owner, "_mggan_prep")
    _PREP["owners"].clear()


def weights_fingerprint(root):
    """What the host can see of writes to the weights of `root`: the kernel-side version counter (optimizer steps, graph
    replays: bump_weight_version) and torch's own version counters (load_state_dict, a parameter broadcast, any in-place
    torch op)."""
    ps = root.__dict__.get("_fp_params")
    if ps is None or root.__dict__.get("_fp_flat") is not getattr(root, "_flat", None):
        # (the parameter list is walked once per flat buffer: a replay takes this fingerprint twice per root)
        ps = root.__dict__["_fp_params"] = tuple(root.parameters())
        root.__dict__["_fp_flat"] = getattr(root, "_flat", None)
    return (getattr(root, "_kernel_version", 0),) + tuple(p._version for p in ps)


def refold_root(root):
    """Re-fold, in place and on the current stream, every cached folded-weight buffer of the modules under `root`.  A graph
    replay reads those buffers as the PREVIOUS iteration left them (a steady-state graph folds a module's weights only behind
    the optimizer step that changed them); anything else that wrote the weights since -- an eager iteration, a loaded
    checkpoint, a broadcast -- is caught by the trainer (weights_fingerprint) and answered with this before the replay."""
    n = 0
    for owner in list(_PREP["owners"]):
        ent = getattr(owner, "_mggan_prep", None)
        if ent is None or getattr(owner, "_flat_root", owner) is not root:
            continue
        ent[2](ent[1])
        ent[0] = None  # (the key no longer describes the contents for the host-side cache: the next eager use folds again)
        n += 1
    return n


def bump_weight_version(root, touched=None):
    """The weights of `root` (a FlatModule) were written by a kernel: an optimizer step (`touched` = ids of the parameters
    it updated -- AdamW skips parameters without a gradient, so the PM network's step leaves the decoders alone) or a graph
    replay (touched=None: everything)."""
    v = getattr(root, "_kernel_version", 0) + 1
    object.__setattr__(root, "_kernel_version", v)
    if touched is None:
        object.__setattr__(root, "_all_written_at", v)
    else:
        at = getattr(root, "_written_at", None)
        if at is None:
            at = {}
            object.__setattr__(root, "_written_at", at)
        for i in touched:
            at[i] = v


def _prep_for(owner, tensors, psz_total, like, fold):
    """-> the folded-weight buffer for `tensors` (the module's raw weights); `fold(prep)` launches the fold.
    Inside the trainer's iteration ONE buffer per module, re-folded in place when its weights have a new version: every
    captured graph and every eager iteration reads and writes the same buffer (a graph's discriminator step reads what
    the previous iteration's generator step folded).  In place is safe: a fold only follows an optimizer step, which is
    ordered behind every reader of the old contents (the backward pass of its own step)."""
    if not (_PREP["on"] and _PREP["enabled"]) or owner is None:
        prep = _empty(psz_total, like=like)
        fold(prep)
        return prep
    root = getattr(owner, "_flat_root", owner)
    at = getattr(root, "_written_at", None) or {}
    written = max([getattr(root, "_all_written_at", 0)] + [at.get(id(t), 0) for t in tensors])
    key = (written,) + tuple((t.data_ptr(), t._version) for t in tensors)
    ent = getattr(owner, "_mggan_prep", None)
    if ent is not None and ent[1].numel() == psz_total and ent[1].device == like.device:
        if ent[0] != key:
            fold(ent[1])
            ent[0] = key
        elif ent[4] is not None:  # folded ahead of time on another stream (prefold): this stream waits for that launch
            _cur().wait_event(ent[4])
        ent[4] = None
        ent[2], ent[3] = fold, tensors  # (the closure of the latest use: same weights, same buffer, the current stream at call time)
        return ent[1]
    prep = torch.empty(psz_total, dtype=F32, device=like.device)
    fold(prep)
    object.__setattr__(owner, "_mggan_prep", [key, prep, fold, tensors, None])
    _PREP["owners"].add(owner)
    return prep


def prefold(owner):
    """Fold `owner`'s weights NOW, on the current stream, if an optimizer step has written them since its buffer was folded
    -- ahead of the module's first use in a step, off that step's critical chain (the trainer calls it on a branch stream:
    the rollout of the PM-network step found its fold, 6 us + a queue hop, between the trunk and the rollout).  The
    module's next use waits for this launch through an event.  No-op for a module that has never been used."""
    ent = getattr(owner, "_mggan_prep", None)
    if ent is None or not (_PREP["on"] and _PREP["enabled"]) or ent[3] is None:
        return False
    root = getattr(owner, "_flat_root", owner)
    at = getattr(root, "_written_at", None) or {}
    written = max([getattr(root, "_all_written_at", 0)] + [at.get(id(t), 0) for t in ent[3]])
    key = (written,) + tuple((t.data_ptr(), t._version) for t in ent[3])
    if ent[0] == key:
        return False
    ent[2](ent[1])
    ent[0] = key
    ev = torch.cuda.Event()
    ev.record(_cur())
    ent[4] = ev
    return True


class LstmEncoderFn(Function):
    """Linear(2,E) + nn.LSTM over T steps -> h_T   (common_modules.py:48-66)."""

    @staticmethod
    def forward(ctx, x, emb_w, emb_b, w_ih, w_hh, b_ih, b_hh, owner, save, out=None):
        T, b, _ = x.shape
        H, E = w_hh.shape[1], emb_w.shape[0]
        x = x.contiguous()
        psz = lib.mggan_lstm_prep_size(H, 0, 0)
        prep = _prep_for(owner, (emb_w, emb_b, w_ih, b_ih, b_hh, w_hh), psz, x,
                         lambda prep: lib.mggan_lstm_fold(_p(emb_w), _p(emb_b), _p(w_ih), _p(b_ih), _p(b_hh), _p(w_hh), 0, 0, 0,
                                                          0, 0, 1, H, E, 0, 0, _p(prep), psz, _s()))
        Gt = _empty(b, T, 4 * H, like=x) if save else None
        Cs = _empty(b, T, H, like=x) if save else None
        Hp = _empty(b, T, H, like=x) if save else None
        Din = _empty(b, T, 2, like=x) if save else None
        hout, ld_out = _out(out, b, H, x)
        lib.mggan_lstm_encoder_fwd(_p(x), T, b, H, _p(prep), _p(hout), ld_out, _p(Gt), _p(Cs), _p(Hp), _p(Din), _s())
        if save:
            ctx.dims, ctx.owner = (T, b, H, E), owner
            ctx.save_for_backward(emb_w, emb_b, w_ih, w_hh, b_ih, b_hh, prep, Gt, Cs, Hp, Din)
        return hout

    @staticmethod
    def backward(ctx, dh):
        emb_w, emb_b, w_ih, w_hh, b_ih, b_hh, prep, Gt, Cs, Hp, Din = ctx.saved_tensors
        T, b, H, E = ctx.dims
        root = root_of(ctx.owner)
        dh, ld = _rows2d(dh)
        dPre = _empty(b, T, 4 * H, like=dh)
        lib.mggan_lstm_encoder_bwd(_p(dh), ld, T, b, H, _p(w_hh), _p(prep), _p(Gt), _p(Cs), _p(dPre), _s())
        rows = b * T
        gp = [root.grad_ptr(t) for t in (w_hh, emb_w, emb_b, w_ih, b_ih, b_hh)]
        dprep = _empty(12 * H, like=dh)  # [dA (4H,2) | dbias (4H)], written (not accumulated) by the reduction
        with side_stream(dPre, Hp, Din, dprep):
            wgrad(dPre, 4 * H, Hp, H, gp[0], H, 0, rows, H, 4 * H)
            later = _DEFER["on"] and not _SIDE["dirty"]  # inside the trainer's backward: GEMM and un-folding are queued
            wgrad(dPre, 4 * H, Din, 2, dprep.data_ptr(), 2, dprep.data_ptr() + 4 * 8 * H, rows, 2, 4 * H, now=not later,
                  overwrite=True)
            unfold = lambda: lib.mggan_lstm_unfold_grads(_p(emb_w), _p(emb_b), _p(w_ih), gp[1], gp[2], gp[3], gp[4], gp[5],
                                                         0, 1, H, E, _p(dprep), 12 * H, _s())
            if later:
                _DEFER["after"].append(unfold)
                _DEFER["keep"].append(dprep)
            else:
                unfold()
        return (None,) * 10


# ------------------------------------------------------------------------------------------
class SceneTables:
    """Device-side index tables of one batch's scene structure (built once per batch)."""

    def __init__(self, seq_start_end, b, device):
        import numpy as np

        sse = [(int(s), int(e)) for s, e in seq_start_end]
        # a list repeated K times (the discriminator's masked path, discriminators.py:183) makes the reference write
        # the same rows K times with identical values: one copy of every scene carries the whole result (A.1)
        sse = list(dict.fromkeys(sse))
        ped_s0 = np.zeros(b, np.int32)
        ped_n = np.ones(b, np.int32)
        ped_prow = np.zeros(b, np.int32)
        ped_scene = np.zeros(b, np.int32)
        pi, pj, P = [], [], 0
        for si, (s, e) in enumerate(sse):
            n = e - s
            ped_scene[s:e] = si
            ped_s0[s:e] = s
            ped_n[s:e] = n
            if n > 1:
                ii, jj = np.meshgrid(np.arange(s, e), np.arange(s, e), indexing="ij")
                pi.append(ii.reshape(-1))
                pj.append(jj.reshape(-1))
                ped_prow[s:e] = P + np.arange(n) * n
                P += n * n
        self.P, self.b, self.S = P, b, len(sse)
        # the row-structured social kernels walk whole scenes of up to 64 pedestrians and write only rows that belong
        # to a scene: they need the scenes to tile [0, b) (every collated batch does)
        self.max_n = int(ped_n.max()) if b else 0
        self.cover = sum(e - s for s, e in sse) == b and all(e > s for s, e in sse)
        self.rows_ok = self.cover and self.max_n <= 64
        cat = (lambda l: np.concatenate(l).astype(np.int32)) if pi else (lambda l: np.zeros(0, np.int32))
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.pair_i, self.pair_j = to(cat(pi)), to(cat(pj))
        self.ped_s0, self.ped_n, self.ped_prow, self.ped_scene = to(ped_s0), to(ped_n), to(ped_prow), to(ped_scene)
        self.scenes = to(np.asarray(sse, np.int32).reshape(-1, 2))
        self.seq_start_end = sse


class StaticSceneTables:
    """The scene tables of a SHAPE BUCKET (train()'s padded batches): fixed capacity (b_pad pedestrians, S_pad scene slots),
    fixed device addresses, contents re-filled in place for every batch -- a captured graph reads them afresh at every replay.
    Slots [0, s_real) are the batch's scenes, the next ones hold the phantom pedestrians [n_real, b_pad) in scenes of at most
    PHANTOM_SCENE of them, the rest are empty (start == end == b_pad).  Only the row-structured kernels are served (scenes of
    at most 64 pedestrians): no pair tables.  `doubled`: also the tables of the discriminator's 2b-row pair pass (rows
    [0, b_pad) real, [b_pad, 2 b_pad) fake: the same scenes twice, discriminators.py forward_pair).
    One packed int32 buffer: [dims 4 | scenes 2S | ped_scene b | ped_s0 b | ped_n b | scenes2 4S | ped_scene2 2b | ped_s02 2b |
    ped_n2 2b] -- ONE host-to-device copy per batch."""
    PHANTOM_SCENE = 16
    static = True

    class View:
        static = True

    def __init__(self, b_pad, S_pad, max_n, device):
        import numpy as np

        self.b, self.S, self.max_n, self.device = int(b_pad), int(S_pad), int(max_n), torch.device(device)
        b, S = self.b, self.S
        sizes = [4, 2 * S, b, b, b, 4 * S, 2 * b, 2 * b, 2 * b]
        offs = np.concatenate([[0], np.cumsum([-(-n // 4) * 4 for n in sizes])])  # 16-byte aligned fields
        self._off = [int(o) for o in offs]
        total = self._off[-1]
        # host staging: a ring of pinned buffers, each guarded by the event of its last copy (the host runs several replays
        # ahead of the GPU: the copy of batch i may not have happened yet when batch i + 1 is laid out)
        cuda = self.device.type == "cuda"  # (on the CPU -- the host-logic tests -- plain buffers, no events)
        self._ring = [[torch.zeros(total, dtype=torch.int32).pin_memory() if cuda else torch.zeros(total, dtype=torch.int32),
                       None] for _ in range(8)]
        self._turn = 0
        self.dev = torch.zeros(total, dtype=torch.int32, device=self.device)
        f = lambda i, n: self.dev[self._off[i]:self._off[i] + n]
        self.dims = f(0, 4)
        self.seq_start_end = [[0, 0] for _ in range(S)]      # the list object the trainer passes around (mutated in place)
        self.seq_start_end2 = [[0, 0] for _ in range(2 * S)]  # ... and the pair pass's
        one, two = self, StaticSceneTables.View()
        one.scenes, one.ped_scene, one.ped_s0, one.ped_n = f(1, 2 * S).view(S, 2), f(2, b), f(3, b), f(4, b)
        two.scenes, two.ped_scene, two.ped_s0, two.ped_n = f(5, 4 * S).view(2 * S, 2), f(6, 2 * b), f(7, 2 * b), f(8, 2 * b)
        empty = torch.zeros(0, dtype=torch.int32, device=self.device)
        for t, bb, SS, lst in ((one, b, S, self.seq_start_end), (two, 2 * b, 2 * S, self.seq_start_end2)):
            t.P, t.b, t.S, t.max_n, t.cover, t.rows_ok = 0, bb, SS, self.max_n, True, True
            t.pair_i = t.pair_j = empty
            t.ped_prow = torch.zeros(bb, dtype=torch.int32, device=self.device)
            t.seq_start_end = lst
        self.pair = two
        self.n_real = self.s_real = 0

    @staticmethod
    def slots_for(S_real_max, b_pad, b_min):
        """Scene slots a bucket needs: its real scenes + the phantom scenes of its emptiest batch."""
        ph = max(b_pad - b_min, 0)
        return int(S_real_max) + -(-ph // StaticSceneTables.PHANTOM_SCENE)

    def fill(self, seq_start_end, stream=None):
        """Write the tables of a batch with these (real) scenes; phantom pedestrians follow the last real one."""
        import numpy as np

        sse = [(int(s), int(e)) for s, e in seq_start_end]
        b, S, PH = self.b, self.S, self.PHANTOM_SCENE
        n_real = sse[-1][1] if sse else 0
        scenes = list(sse)
        p = n_real
        while p < b:
            scenes.append((p, min(p + PH, b)))
            p += PH
        if len(scenes) > S or n_real > b or max((e - s_ for s_, e in sse), default=0) > self.max_n:
            raise ValueError("batch does not fit its bucket ({} scenes incl. phantoms > {}, or {} pedestrians > {})".format(
                len(scenes), S, n_real, b))
        s_real = len(sse)
        scenes += [(b, b)] * (S - len(scenes))
        sc = np.asarray(scenes, np.int32).reshape(S, 2)
        n = sc[:, 1] - sc[:, 0]
        ped_scene = np.repeat(np.arange(S, dtype=np.int32), n)
        ped_s0 = np.repeat(sc[:, 0], n)
        ped_n = np.repeat(n, n).astype(np.int32)
        slot = self._ring[self._turn]
        self._turn = (self._turn + 1) % len(self._ring)
        if slot[1] is not None:
            slot[1].synchronize()
        h, o = slot[0].numpy(), self._off
        h[0], h[1] = n_real, s_real
        h[2:3].view(np.float32)[0] = np.float32(b) / np.float32(max(n_real, 1))
        h[o[1]:o[1] + 2 * S] = sc.reshape(-1)
        h[o[2]:o[2] + b], h[o[3]:o[3] + b], h[o[4]:o[4] + b] = ped_scene, ped_s0, ped_n
        h[o[5]:o[5] + 4 * S] = np.concatenate([sc, sc + b]).reshape(-1)
        h[o[6]:o[6] + 2 * b] = np.concatenate([ped_scene, ped_scene + S])
        h[o[7]:o[7] + 2 * b] = np.concatenate([ped_s0, ped_s0 + b])
        h[o[8]:o[8] + 2 * b] = np.concatenate([ped_n, ped_n])
        self.dev.copy_(slot[0], non_blocking=True)
        if self.device.type == "cuda":
            slot[1] = torch.cuda.Event()
            slot[1].record()
        for lst, arr in ((self.seq_start_end, sc), (self.seq_start_end2, np.concatenate([sc, sc + b]))):
            for item, (s_, e) in zip(lst, arr.tolist()):
                item[0], item[1] = s_, e
        self.n_real, self.s_real = n_real, s_real

    def register(self, pair_cache=None):
        """Make scene_tables() (and the discriminator's pair-pass cache) hand these tables out for the static lists."""
        key = (id(self.seq_start_end), self.b, str(self.device))
        _TABLE_CACHE[key] = (self.seq_start_end, self, None)
        _TABLE_CACHE.pinned.add(key)
        key2 = (id(self.seq_start_end2), 2 * self.b, str(self.device))
        _TABLE_CACHE[key2] = (self.seq_start_end2, self.pair, None)
        _TABLE_CACHE.pinned.add(key2)
        if pair_cache is not None:
            pair_cache[id(self.seq_start_end)] = (self.seq_start_end, self.seq_start_end2, None, True)
            pair_cache.pinned.add(id(self.seq_start_end))


# Per-batch index tables, row tables and random-number pools are built lazily (with host copies) and cached by shape or
# by the identity of the caller's scene list.  A captured HIP graph holds raw pointers into them, so whatever an iteration
# touched while `pin_tables()` was active (the trainer's graph cache: the first eager iteration on a static batch and
# its capture) is never evicted; everything else is bounded.
_PIN = {"on": 0}


class pin_tables:
    def __enter__(self):
        _PIN["on"] += 1
        return self

    def __exit__(self, *exc):
        _PIN["on"] -= 1
        return False


class BoundedCache(dict):
    def __init__(self, limit):
        super().__init__()
        self.limit, self.pinned = limit, set()

    def get(self, key, default=None):
        hit = super().get(key, default)
        if hit is not default and _PIN["on"]:
            self.pinned.add(key)
        return hit

    def put(self, key, value):
        if len(self) - len(self.pinned) > self.limit:
            for k in [k for k in self if k not in self.pinned]:
                del self[k]
        self[key] = value
        if _PIN["on"]:
            self.pinned.add(key)
        return value


_TABLE_CACHE = BoundedCache(64)


def scene_fingerprint(seq_start_end):
    """Cheap content check for caches keyed by id(list): a caller that mutates its scene list in place (same id,
    other bounds) must not get the tables of the old contents back."""
    n = len(seq_start_end)
    if n == 0:
        return (0,)
    return (n, int(seq_start_end[0][0]), int(seq_start_end[-1][1]), sum(int(e) * (i + 1) for i, (_, e) in enumerate(seq_start_end)))


def scene_tables(seq_start_end, b, device):
    key = (id(seq_start_end), b, str(device))
    hit = _TABLE_CACHE.get(key)
    fp = scene_fingerprint(seq_start_end)
    if hit is not None and hit[0] is seq_start_end and (hit[2] == fp or getattr(hit[1], "static", False)):
        return hit[1]  # (static: the tables of a shape bucket, re-filled in place for every batch)
    t = SceneTables(seq_start_end, b, device)
    _TABLE_CACHE.put(key, (seq_start_end, t, fp))
    return t


def alias_cols(buf, c0, c1):
    """Columns [c0, c1) of a 2-d buffer as a tensor that SHARES the storage without being an autograd view of it
    (Tensor.set_ on the storage: metadata only, no kernel).  A real view would tie the version counters together and
    autograd refuses custom-Function outputs whose base is written again -- here several stages fill disjoint
    column blocks of one buffer."""
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset() + c0 * buf.stride(1),
                                                                   (buf.shape[0], c1 - c0), (buf.stride(0), buf.stride(1)))


class OutSlot:
    """Where a stage should leave its (rows, c1-c0) result: a column block of a wider preallocated buffer.  It is
    handed to the autograd Functions as a plain Python object (not a tensor input): the Function creates the view inside
    its forward and returns it as its output, so no torch.cat is needed to assemble [lstm | scene | social] and friends."""

    def __init__(self, buf, c0, c1):
        assert buf.dim() == 2 and buf.stride(1) == 1 and 0 <= c0 < c1 <= buf.shape[1]
        self.buf, self.c0, self.c1 = buf, c0, c1

    def view(self):
        return alias_cols(self.buf, self.c0, self.c1)

    @property
    def ld(self):
        return self.buf.stride(0)


def _out(slot, rows, cols, like):
    """-> (tensor to return, row stride) for a stage output: the slot's view or a fresh tensor."""
    if slot is None:
        return _empty(rows, cols, like=like), cols
    v = slot.view()
    assert tuple(v.shape) == (rows, cols), (tuple(v.shape), rows, cols)
    return v, slot.ld


def _social_fwd(xy_last, dxdy_last, h_ptr, ld_h, b, Hh, tb, w1, b1, w2, b2, w3, b3, wat, bat, S_ptr, ld_s, save, xy_mod,
                like):
    """SocialFeatures -> EmbedSocialFeatures -> AttentionPooling for rows [0,b) of a hidden-state matrix given by pointer
    + row stride; S (b,Hh) is written through S_ptr / ld_s.  -> tuple of saved tensors for _social_bwd."""
    Fd = wat.shape[0]
    st = _s()
    if tb.rows_ok and Fd <= 64 and Fd % 16 == 0:
        # ONE launch: Wh = h W_at^T + b_at, [v | c] = Wh [W3 | b3], pair MLP (MFMA), scores, softmax, pooling; nothing is
        # stored for the backward pass, which recomputes all of it from h and the positions
        if _load_lib().trace is not None:
            TRACE_NOTES["social_rows_fwd_pairs"].append(tb.P)
        lib.mggan_social_rows_fwd(tb.S, _p(tb.scenes), Hh, Fd, tb.max_n, _p(xy_last), _p(dxdy_last), int(xy_mod), _p(w1),
                                  _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(wat), _p(bat), h_ptr, ld_h, S_ptr, ld_s, st)
        return (None,) * 7
    # a scene of more than 64 pedestrians (or scenes that do not tile the batch): per-stage kernels over the pair list
    W3b = _empty(Fd, 65, like=like)
    lib.mggan_social_w3b(_p(w3), _p(b3), _p(W3b), Fd, st)
    # Wh = h W_at^T + b_at and vc = Wh [W3 | b3] as one two-stage chain launch
    Wh, vc = _empty(b, Fd, like=like), _empty(b, 65, like=like)
    a = _McArgs()
    a.X, a.ldx, a.rows, a.K0, a.n = h_ptr, ld_h, b, Hh, 2
    s0, s1 = a.s[0], a.s[1]
    s0.W, s0.bias, s0.out, s0.K, s0.N, s0.ldw, s0.trans, s0.act, s0.ld_out = _p(wat), _p(bat), _p(Wh), Hh, Fd, Hh, 0, ACT_NONE, Fd
    s1.W, s1.out, s1.K, s1.N, s1.ldw, s1.trans, s1.act, s1.ld_out = _p(W3b), _p(vc), Fd, 65, 65, 1, ACT_NONE, 65
    if _load_lib().trace is not None:
        TRACE_NOTES["mlp_chain_flops"].append(2.0 * b * (Hh * Fd + Fd * 65))
    lib.mggan_mlp_chain(ctypes.addressof(a), st)
    P = tb.P
    feat = _empty(3, max(P, 1), like=like) if save else None      # feature-major [feature][pair]
    l1 = _empty(32, max(P, 1), like=like) if save else None
    l2 = _empty(64, max(P, 1), like=like) if save else None
    att = _empty(max(P, 1), like=like)
    if xy_mod:
        rep_n = b // xy_mod
        xy_last, dxdy_last = xy_last.repeat(rep_n, 1), dxdy_last.repeat(rep_n, 1)
    sigma = _empty(max(P, 1), like=like)
    lib.mggan_social_pairs_fwd(P, _p(tb.pair_i), _p(tb.pair_j), _p(xy_last), _p(dxdy_last), _p(w1), _p(b1),
                               _p(w2), _p(b2), _p(vc), _p(feat), _p(l1), _p(l2), _p(sigma), st)
    lib.mggan_social_softmax_fwd(b, Hh, _p(tb.ped_prow), _p(tb.ped_s0), _p(tb.ped_n), _p(sigma), h_ptr, ld_h,
                                 _p(att), S_ptr, ld_s, st)
    return (W3b, Wh, vc, feat, l1, l2, att)


def _social_bwd(saved, xy_last, dxdy_last, xy_mod, h_ptr, ld_h, h_keep, b, Hh, tb, w1, b1, w2, b2, w3, b3, wat, bat, dS_ptr,
                ld_ds, dh_ptr, ld_dh, accumulate_dh, train_w1, train_w3, owner, like):
    """Adjoint of _social_fwd.  dh (b,Hh) goes to dh_ptr / ld_dh (accumulate_dh: added to what is there); the weight
    gradients are queued / launched like everywhere else.  h_keep: the tensor that owns h (kept alive)."""
    W3b, Wh, vc, feat, l1, l2, att = saved
    root = root_of(owner)
    Fd, P = wat.shape[0], tb.P
    st = _s()
    if tb.rows_ok and Fd <= 64 and Fd % 16 == 0:
        ldv = 68
        dvc, Wh, dWh = _empty(b, ldv, like=like), _empty(b, Fd, like=like), _empty(b, Fd, like=like)
        part, grid, pf = None, 0, 0
        if train_w1:  # one partial block [dW2 | db2 ; dW1 | db1] per workgroup, folded by the batched reduction
            grid, pf = lib.mggan_social_rows_grid(tb.S, tb.max_n), lib.mggan_social_rows_partial_floats()
            part = _empty(grid * pf, like=like)
        rs, scr, tick = lib.mggan_social_rows_splits(tb.S, tb.max_n), None, None
        if rs > 1:  # few scenes: a scene's rows are dealt to `rs` workgroups whose neighbour sums meet in scratch
            scr = _empty(rs * b * (65 + Hh), like=like)
            tick = tb.__dict__.setdefault("soc_tickets", {}).get(Hh)
            if tick is None:  # zero once; the kernel leaves the words at zero (one launch per (tables, width) at a time)
                tick = tb.soc_tickets[Hh] = torch.zeros(tb.S, dtype=torch.int32, device=like.device)
        if _load_lib().trace is not None:
            TRACE_NOTES["social_rows_bwd_pairs"].append(tb.P)
        lib.mggan_social_rows_bwd(tb.S, _p(tb.scenes), Hh, Fd, tb.max_n, _p(xy_last), _p(dxdy_last), int(xy_mod), _p(w1),
                                  _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(wat), _p(bat), h_ptr, ld_h, dS_ptr, ld_ds,
                                  _p(dvc), ldv, b, _p(Wh), _p(dWh), dh_ptr, ld_dh, int(accumulate_dh), _p(part), _p(scr),
                                  _p(tick), st)
        if scr is not None and (_DEFER["on"] or _SIDE["dirty"]):
            _DEFER["keep"].append(scr)
        if train_w1:
            p0 = part.data_ptr()
            descs = ((p0, root.grad_ptr(w2), root.grad_ptr(b2), 64, 33, 32), (p0 + 4 * 64 * 33, root.grad_ptr(w1), root.grad_ptr(b1), 32, 4, 3))
            if _DEFER["on"] and not _SIDE["dirty"]:
                for k, (pp_, dw, db, M, Naug, lddw) in enumerate(descs):
                    _queue_reduce(pp_, dw, db, M, Naug, 1, lddw, grid, 1, pf, keep=(part,) if k == 0 else ())
            else:
                arr = (_ReduceDesc * 2)(*[_ReduceDesc(pp_, dw, db or None, 0, 0, M, Naug, 1, lddw, grid, 1, pf, 0)
                                          for pp_, dw, db, M, Naug, lddw in descs])
                lib.mggan_grad_reduce_multi(ctypes.addressof(arr), 2, st)
                if _SIDE["dirty"]:
                    _SIDE["keep"].append(part)
    else:
        ldv = 65
        dvc = _empty(b, ldv, like=like)
        dsigma = _empty(max(P, 1), like=like)
        dz2 = _empty(64, max(P, 1), like=like)
        dz1 = _empty(32, max(P, 1), like=like)
        lib.mggan_social_softmax_bwd(b, Hh, _p(tb.ped_prow), _p(tb.ped_s0), _p(tb.ped_n), _p(att), h_ptr, ld_h,
                                     dS_ptr, ld_ds, _p(dsigma), dh_ptr, ld_dh, int(accumulate_dh), st)
        lib.mggan_social_pairs_bwd(P, b, _p(tb.pair_j), _p(tb.ped_prow), _p(tb.ped_s0), _p(tb.ped_n), _p(dsigma),
                                   _p(vc), _p(l1), _p(l2), _p(w2), _p(dz2), _p(dz1), _p(dvc), st)
        if train_w1:
            with side_stream(dz2, dz1, l1, feat):
                wgrad(dz2, P, l1, P, root.grad_ptr(w2), 32, root.grad_ptr(b2), P, 32, 64, fm=1)
                wgrad(dz1, P, feat, P, root.grad_ptr(w1), 3, root.grad_ptr(b1), P, 3, 32, fm=1)
        # dWh = dvc [W3|b3]^T and dh += dWh W_at as one two-stage chain launch; d[W3|b3] = Wh^T dvc
        dWh = _empty(b, Fd, like=like)
        a = _McArgs()
        a.X, a.ldx, a.rows, a.K0, a.n = _p(dvc), ldv, b, 65, 2
        s0, s1 = a.s[0], a.s[1]
        s0.W, s0.out, s0.K, s0.N, s0.ldw, s0.trans, s0.act, s0.ld_out = _p(W3b), _p(dWh), 65, Fd, 65, 0, ACT_NONE, Fd
        s1.W, s1.out, s1.K, s1.N, s1.ldw, s1.trans, s1.act, s1.ld_out, s1.accumulate = _p(wat), dh_ptr, Fd, Hh, Hh, 1, ACT_NONE, ld_dh, 1
        if _load_lib().trace is not None:
            TRACE_NOTES["mlp_chain_flops"].append(2.0 * b * (65 * Fd + Fd * Hh))
        lib.mggan_mlp_chain(ctypes.addressof(a), st)
    if train_w3:
        with side_stream(Wh, dvc, dWh, h_keep):
            wgrad(Wh, Fd, dvc, ldv, root.grad_ptr(w3), 64, 0, b, 64, Fd)
            wgrad(Wh, Fd, dvc.data_ptr() + 4 * 64, ldv, root.grad_ptr(b3), 1, 0, b, 1, Fd)
            wgrad(dWh, Fd, h_ptr, ld_h, root.grad_ptr(wat), Hh, root.grad_ptr(bat), b, Hh, Fd)
        if _DEFER["on"]:
            _DEFER["keep"].append(h_keep)


class SocialAttentionFn(Function):
    """SocialFeatures -> EmbedSocialFeatures -> AttentionPooling over in-scene pairs (social.py:7-123)."""

    @staticmethod
    def forward(ctx, xy_last, dxdy_last, h, tb, w1, b1, w2, b2, w3, b3, wat, bat, owner, save, xy_mod=0):
        h, ld_h = _rows2d(h)
        b, Hh = h.shape
        xy_last, dxdy_last = xy_last.contiguous(), dxdy_last.contiguous()
        S = _empty(b, Hh, like=h)
        saved = _social_fwd(xy_last, dxdy_last, _p(h), ld_h, b, Hh, tb, w1, b1, w2, b2, w3, b3, wat, bat, _p(S), Hh, save,
                            xy_mod, h)
        if save:
            ctx.tb, ctx.owner, ctx.ld_h, ctx.xy_mod = tb, owner, ld_h, xy_mod
            ctx.train_w1, ctx.train_w3 = w1.requires_grad, w3.requires_grad
            ctx.save_for_backward(h, w1, b1, w2, b2, w3, b3, wat, bat, xy_last, dxdy_last, *saved)
        return S

    @staticmethod
    def backward(ctx, dS):
        h, w1, b1, w2, b2, w3, b3, wat, bat, xy_last, dxdy_last = ctx.saved_tensors[:11]
        saved = ctx.saved_tensors[11:]
        tb, ld_h = ctx.tb, ctx.ld_h
        b, Hh = h.shape
        dS, ld_ds = _rows2d(dS)
        if dS.data_ptr() % 16:
            dS = dS.clone()
        dh = _empty(b, Hh, like=h)
        _social_bwd(saved, xy_last, dxdy_last, ctx.xy_mod, _p(h), ld_h, h, b, Hh, tb, w1, b1, w2, b2, w3, b3, wat, bat,
                    _p(dS), ld_ds, _p(dh), Hh, 0, ctx.train_w1, ctx.train_w3, ctx.owner, h)
        return (None, None, dh if ctx.needs_input_grad[2] else None) + (None,) * 12


class TrunkJoinFn(Function):
    """The generator trunk's last stage (standard.py:144-155): social attention over the encoder state, and
    enc_h = [lstm | scene | social] WITHOUT a concatenation: `enc` (and, through SceneTapFn below, `scene`) already sit
    in their column blocks of the (b, 128) buffer (their producers were handed OutSlots), the social features are
    written into the third block.  The gradient w.r.t. the encoder state (its column block of d enc_h plus the
    attention's dh) is formed in place: no gradient add, no slice copies.
    -> (enc_h (b,128), social (b,32) = its last column block)."""

    @staticmethod
    def forward(ctx, enc, xy_last, dxdy_last, tb, buf, w1, b1, w2, b2, w3, b3, wat, bat, owner, save, Sc):
        b, W = buf.shape
        Hh = enc.shape[1]
        Fd = wat.shape[0]
        assert W == Hh + Sc + Fd and buf.stride(1) == 1
        base = buf.data_ptr()
        if enc.data_ptr() != base or (b > 1 and enc.stride(0) != W):   # producer did not use its slot: copy in
            alias_cols(buf, 0, Hh).copy_(enc)
        xy_last, dxdy_last = xy_last.contiguous(), dxdy_last.contiguous()
        saved = _social_fwd(xy_last, dxdy_last, base, W, b, Hh, tb, w1, b1, w2, b2, w3, b3, wat, bat,
                            base + 4 * (Hh + Sc), W, save, 0, buf)
        if save:
            ctx.tb, ctx.owner, ctx.dims = tb, owner, (b, W, Hh, Sc, Fd)
            ctx.train_w1, ctx.train_w3 = w1.requires_grad, w3.requires_grad
            ctx.save_for_backward(buf, w1, b1, w2, b2, w3, b3, wat, bat, xy_last, dxdy_last, *saved)
        ctx.set_materialize_grads(False)
        return alias_cols(buf, 0, W), alias_cols(buf, Hh + Sc, W)

    @staticmethod
    def backward(ctx, d_enc_h, d_soc):
        buf, w1, b1, w2, b2, w3, b3, wat, bat, xy_last, dxdy_last = ctx.saved_tensors[:11]
        saved = ctx.saved_tensors[11:]
        b, W, Hh, Sc, Fd = ctx.dims
        if d_enc_h is None:
            d_enc_h = torch.zeros(b, W, dtype=F32, device=buf.device)
        elif d_enc_h.stride(1) != 1 or (b > 1 and d_enc_h.stride(0) != W):
            d_enc_h = d_enc_h.contiguous()
        elif d_enc_h.data_ptr() % 16:  # contiguous() is a no-op on a contiguous view at an odd storage offset
            d_enc_h = d_enc_h.clone()
        if d_soc is not None:  # a consumer of the separate `social` output that did not fold its gradient into d enc_h
            d_enc_h = d_enc_h.clone()
            d_enc_h[:, Hh + Sc:] += d_soc
        g = d_enc_h.data_ptr()
        # dS = d_enc_h[:, social block]; dh is ADDED to d_enc_h[:, lstm block] in place (this node is the only reader)
        _social_bwd(saved, xy_last, dxdy_last, 0, buf.data_ptr(), W, buf, b, Hh, ctx.tb, w1, b1, w2, b2, w3, b3, wat, bat,
                    g + 4 * (Hh + Sc), W, g, W, 1, ctx.train_w1, ctx.train_w3, ctx.owner, buf)
        return (d_enc_h[:, :Hh],) + (None,) * 15


class ColsTapFn(Function):
    """`part` already sits in columns [c0, c1) of `whole` (its producer wrote there through an OutSlot): declare that
    to autograd.  Forward: nothing to compute (copies `part` in only if the producer did not use the slot).  Backward:
    the gradient of `part` is that column block of d whole -- available at once, BEFORE the adjoint of whatever
    produced the rest of `whole` runs, so that a long branch (the scene CNN's adjoint) starts as early as with
    torch.cat's slice backward, without its copies."""

    @staticmethod
    def forward(ctx, whole, part, c0, c1):
        base = whole.data_ptr() + 4 * c0 * whole.stride(1)
        if part.data_ptr() != base or (whole.shape[0] > 1 and part.stride(0) != whole.stride(0)):
            alias_cols(whole, c0, c1).copy_(part)
        ctx.cols = (c0, c1)
        ctx.set_materialize_grads(False)
        return alias_cols(whole, 0, whole.shape[1])

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        c0, c1 = ctx.cols
        return g, g[:, c0:c1], None, None


# ------------------------------------------------------------------------------------------
class PoolTables:
    """Pair lists of the Social-GAN pooling for one scene LIST (entries may repeat: the discriminator passes the list
    K times, social_gan.py:212-228 walks it entry by entry): output rows in list order."""

    def __init__(self, seq_start_end, device):
        import numpy as np

        po, pi, pj, prow, pn = [], [], [], [], []
        P = rows = 0
        hmax = 0
        for s, e in seq_start_end:
            s, e = int(s), int(e)
            n = e - s
            if n <= 0:
                continue
            ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
            po.append(rows + ii.reshape(-1))
            pi.append(s + ii.reshape(-1))
            pj.append(s + jj.reshape(-1))
            prow.append(P + np.arange(n) * n)
            pn.append(np.full(n, n))
            P += n * n
            rows += n
            hmax = max(hmax, e)
        cat = lambda l: np.concatenate(l).astype(np.int32) if l else np.zeros(0, np.int32)
        pair_j = cat(pj)
        order = np.argsort(pair_j, kind="stable").astype(np.int32)  # pairs grouped by the hidden row they read
        hid_ptr = np.zeros(hmax + 1, np.int32)
        np.add.at(hid_ptr, pair_j + 1, 1)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.P, self.rows, self.h_rows = P, rows, hmax
        self.pair_o, self.pair_i, self.pair_j = to(cat(po)), to(cat(pi)), to(pair_j)
        self.ped_prow, self.ped_n = to(cat(prow)), to(cat(pn))
        self.hid_ptr, self.hid_pairs = to(np.cumsum(hid_ptr).astype(np.int32)), to(order)
        self.key = seq_start_end


_POOL_CACHE = BoundedCache(64)


def pool_tables(seq_start_end, device):
    key = (id(seq_start_end), len(seq_start_end), str(device))
    hit = _POOL_CACHE.get(key)
    fp = scene_fingerprint(seq_start_end)
    if hit is not None and hit.key is seq_start_end and hit.fp == fp:
        return hit
    t = PoolTables(seq_start_end, device)
    t.fp = fp
    return _POOL_CACHE.put(key, t)


class PoolHiddenFn(Function):
    """PoolHiddenNet.forward (social_gan.py:199-229): [Linear(2,E)(p_j - p_i) | h_j] -> Linear -> ReLU -> Linear ->
    max over the scene.  Pair rows by one launch, the MLP as one fused chain, the maximum by one launch."""

    @staticmethod
    def forward(ctx, xy_last, h, tb, we, be, w1, b1, w2, b2, owner, save, xy_mod=0):
        h, ld_h = _rows2d(h)
        H, E, B = h.shape[1], we.shape[0], w2.shape[0]
        xy_last = xy_last.contiguous()
        P = tb.P
        st = _s()
        X, rel = _empty(max(P, 1), E + H, like=h), _empty(max(P, 1), 2, like=h)
        lib.mggan_pool_pairs_fwd(P, _p(tb.pair_i), _p(tb.pair_j), _p(xy_last), int(xy_mod), _p(we), _p(be), E, _p(h), ld_h,
                                 H, _p(X), _p(rel), st)
        spec = ((ACT_LEAKY, 0.0), (ACT_NONE, 0.0))  # ReLU = leaky slope 0
        outs = _chain_fwd(X, E + H, P, spec, (w1, w2), (b1, b2), save) if P else [None, _empty(0, B, like=h)]
        out = _empty(tb.rows, B, like=h)
        arg = torch.empty(tb.rows, B, dtype=torch.int32, device=h.device)
        lib.mggan_segment_max_fwd(tb.rows, B, _p(tb.ped_prow), _p(tb.ped_n), _p(outs[-1]), _p(out), _p(arg), st)
        if save:
            ctx.tb, ctx.owner, ctx.spec, ctx.dims = tb, owner, spec, (H, E, B, h.shape[0])
            ctx.train_w = w1.requires_grad
            ctx.save_for_backward(X, rel, outs[0], outs[1], arg, we, be, w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        X, rel, y0, y1, arg, we, be, w1, b1, w2, b2 = ctx.saved_tensors
        tb, (H, E, B, hr) = ctx.tb, ctx.dims
        P = tb.P
        st = _s()
        dout, ld = _rows2d(dout)
        dY = _empty(max(P, 1), B, like=X)
        lib.mggan_segment_max_bwd(P, B, _p(tb.pair_o), _p(tb.ped_prow), _p(dout), ld, _p(arg), _p(dY), st)
        dX = _chain_bwd(dY, B, X, E + H, P, (y0, y1), ctx.spec, (w1, w2), (b1, b2), True, ctx.train_w, ctx.owner)
        dh = None
        if ctx.needs_input_grad[1]:
            dh = torch.zeros(hr, H, dtype=F32, device=X.device) if tb.h_rows < hr else _empty(hr, H, like=X)
            lib.mggan_pool_gather_bwd(tb.h_rows, H, E, _p(tb.hid_ptr), _p(tb.hid_pairs), _p(dX), _p(dh), H, st)
        if ctx.train_w:
            root = root_of(ctx.owner)
            with side_stream(dX, rel):
                wgrad(dX, E + H, rel, 2, root.grad_ptr(we), 2, root.grad_ptr(be), P, 2, E)
        return (None, dh) + (None,) * 10


# Sharded training: the masked steps issue other collectives than the unmasked ones (no shared discriminator context, a
# scene-CNN pass on img[mask] per call), so once ANY rank holds a pedestrian without ground truth every rank walks the
# masked path -- also a rank whose own mask is all True (mggan/abstract_train.py sets this around the iteration).
_MASK = {"force": False}


def force_masked(on=None):
    was = _MASK["force"]
    if on is not None:
        _MASK["force"] = bool(on)
    return was


def is_masked(mask):
    """Does `mask` select the masked code path?  (None == every pedestrian valid: no device sync.)"""
    return mask is not None and (_MASK["force"] or not bool(mask.all()))


# Gram matrices of the image patches (csrc/cnn2.hip: image_gram_kernel): the image-only part of every conv1 weight
# gradient of a batch.  The trainer announces the batch's images at the start of an iteration (begin_images): ONE launch
# on a side stream serves the backward passes of both scene CNNs; a backward pass that finds no announced Gram matrix
# (a stand-alone call) computes it on the spot.
_GRAM = {"reg": {}, "stream": None, "pending": None}


def _gram_launch(img, dims=None):
    B = img.shape[0]
    dims = _PAD["dims"] if dims is None else dims
    gram = torch.empty(37 * 37, dtype=torch.float64, device=img.device)
    nb = lib.mggan_image_gram_workspace(B)
    ws = torch.empty(nb // 8, dtype=torch.float64, device=img.device)
    lib.mggan_image_gram(_p(img), B, _p(gram), _p(ws), nb, _pad_ptr(dims) if dims is not None else 0, _s())
    return gram, ws


def begin_images(img, side=True, defer=False, sync=None):
    """Announce this batch's image crops (B,4,33,33); their Gram matrix is started on a side stream now, or -- defer=True
    -- at the point of the iteration where launch_images() is called (the trainer picks a latency-bound stretch: next to
    the discriminator's convolutions and LSTM it only stretches all three).
    sync (a DistContext, sharded training with global-batch BatchNorm): the Gram matrix is computed NOW on the caller's
    stream and all-reduced -- the ONE exchange that gives every conv1 forward pass of the iteration its global statistics
    and every conv1 weight gradient its image-only part (global_gram)."""
    _GRAM["reg"].clear()
    _GRAM["pending"] = None
    _GRAM["global"] = None
    _GRAM["pending_global"] = None
    if img is None or not img.is_cuda or (img.shape[0] == 0 and sync is None):
        return
    img = img.contiguous()
    if sync is not None:
        choose_gram_schedule(img.shape[0], sync)
        # (a rank whose shard holds no pedestrian announces its empty batch like any other: the Gram kernel copes with
        #  B = 0, and every rank must issue the same collectives on the same channels)
        if GRAM_SCHEDULE == "late" and side and _BR["on"] and getattr(sync, "stream_safe", False):
            _GRAM["pending_global"] = (img, sync)  # started by launch_images()
        else:
            global_gram(img, sync, side=side and GRAM_SCHEDULE != "first")
        return
    if defer and side and _BR["on"]:
        _GRAM["pending"] = img
        return
    _start_gram(img, side)


def global_gram(img, sync, side=False):
    """-> the 37 x 37 Gram matrix of the image patches of the GLOBAL batch (this rank's, all-reduced; the element
    [36][36] is the global number of conv1 output positions).  One collective per batch, cached per iteration.
    side=True (the trainer, peer-mapped collectives, branch streams on): launch and exchange go to the Gram side stream --
    on a channel of their own, csrc/comm.hip arenas are per stream role -- and readers wait for its event."""
    hit = _GRAM.get("global")
    if hit is not None and hit[1] == img.data_ptr() and hit[2] == tuple(img.shape):
        if hit[4] is not None:
            _cur().wait_event(hit[4])
            hit[0].record_stream(_cur())
        return hit[0]
    _GRAM["pending_global"] = None  # (announced for later, needed now: e.g. an iteration without a discriminator step)
    ev = None
    if side and _BR["on"] and getattr(sync, "stream_safe", False):
        if _GRAM["stream"] is None:
            _GRAM["stream"] = role_stream("gram")
        st = _GRAM["stream"]
        st.wait_stream(_cur())
        with torch.cuda.stream(st):
            gram, ws = _gram_launch(img)
            sync.all_reduce_(gram, what="gram")
            ev = torch.cuda.Event()
            ev.record(st)
    else:
        gram, ws = _gram_launch(img)
        sync.all_reduce_(gram, what="gram")
    _GRAM["global"] = (gram, img.data_ptr(), tuple(img.shape), ws, ev)
    _GRAM["passes"] = 0
    # (end_images joins the side stream through this entry; nobody looks the matrix up here in sharded mode)
    _GRAM["reg"][img.data_ptr()] = (gram, ws, ev, img)
    return gram


# WHEN the global Gram matrix is computed and exchanged (MGGAN_GRAM_SCHEDULE; measured on one rank with the collective hooks
# forced on):
#   late  (default) -- where the single-GPU path computes it: on the side stream beside the discriminator step's row pass.
#                      The scene-CNN passes issued before that point (the shared trunk, the discriminator's context of the
#                      discriminator step) exchange their own 2C sums: 13 exchanges per iteration.
#   side            -- on the side stream from the start of the iteration; only the FIRST pass exchanges its own sums (12),
#                      but the launch runs beside the trunk's conv1 (+270 us at 256 x 32: both fill the chip).
#   first           -- on the caller's stream before anything else: every pass uses it (11), the iteration waits for it.
#   auto  (default) -- decided per batch from the shard's image count: `first` from GRAM_FIRST_MIN_B images per rank on (the
#                      Gram launch is a few per cent of such an iteration and saves two exchanges), `late` below.  A host-side
#                      decision every rank must take alike: only shards known to be equal (DistContext.equal_shards, what a
#                      captured iteration needs anyway) go by their size, unequal ones stay `late`.
_GRAM_SCHEDULE_ENV = os.environ.get("MGGAN_GRAM_SCHEDULE", "auto")
GRAM_FIRST_MIN_B = int(os.environ.get("MGGAN_GRAM_FIRST_MIN_B", "4096"))
GRAM_SCHEDULE = "late" if _GRAM_SCHEDULE_ENV == "auto" else _GRAM_SCHEDULE_ENV


def choose_gram_schedule(n_images, sync):
    """Set the schedule of this iteration (begin_images calls it before anything is announced)."""
    global GRAM_SCHEDULE
    if _GRAM_SCHEDULE_ENV != "auto":
        return GRAM_SCHEDULE
    GRAM_SCHEDULE = "first" if (getattr(sync, "equal_shards", False) and n_images >= GRAM_FIRST_MIN_B) else "late"
    return GRAM_SCHEDULE


def gram_for_pass(img, sync):
    """The global Gram matrix for the BatchNorm-1 statistics of a scene-CNN forward pass, or None = this pass exchanges its
    own sums (the matrix has not been started yet, or -- schedule `side` -- it is the first pass of the iteration and would
    wait for it).  A host-side decision: every rank runs the same program, they all decide alike."""
    hit = _GRAM.get("global")
    if hit is None or hit[1] != img.data_ptr() or hit[2] != tuple(img.shape):
        pend = _GRAM.get("pending_global")
        if pend is not None and pend[0].data_ptr() == img.data_ptr():
            return None  # announced, not started yet
        return global_gram(img, sync)  # a stand-alone module call: computed (and exchanged) on the spot
    n = _GRAM.get("passes", 0)
    _GRAM["passes"] = n + 1
    if n == 0 and hit[4] is not None and GRAM_SCHEDULE == "side":
        return None
    return global_gram(img, sync)


def _start_gram(img, side=True, after_branches=()):
    if side and _BR["on"]:
        if _GRAM["stream"] is None:
            _GRAM["stream"] = role_stream("gram")
        st = _GRAM["stream"]
        st.wait_stream(_cur())
        for w in after_branches:  # ... and behind what those branch streams have queued (without joining them)
            if w in _BR["streams"]:
                st.wait_stream(_BR["streams"][w])
        with torch.cuda.stream(st):
            gram, ws = _gram_launch(img)
            ev = torch.cuda.Event()
            ev.record(st)
        _GRAM["reg"][img.data_ptr()] = (gram, ws, ev, img)
    else:
        gram, ws = _gram_launch(img)
        _GRAM["reg"][img.data_ptr()] = (gram, ws, None, img)


def launch_images(after_branches=()):
    """Start the announced Gram matrix behind what the current stream (and the named branch streams) have queued so far
    (no-op when it is running)."""
    img, _GRAM["pending"] = _GRAM.get("pending"), None
    if img is not None:
        _start_gram(img, True, after_branches)
    pend, _GRAM["pending_global"] = _GRAM.get("pending_global"), None
    if pend is not None:
        global_gram(pend[0], pend[1], side=True)


def end_images():
    """The side stream of the Gram launch joins the current stream (every fork has to be joined before a capture ends)."""
    st = _GRAM["stream"]
    if st is not None and any(e[2] is not None for e in _GRAM["reg"].values()):
        _cur().wait_stream(st)
    _GRAM["reg"].clear()
    _GRAM["pending"] = None
    _GRAM["global"] = None
    _GRAM["pending_global"] = None


def _image_gram(img, dims=None):
    launch_images()  # (announced but never started: a step order the trainer did not foresee)
    hit = _GRAM["reg"].get(img.data_ptr())
    if hit is not None and hit[3].shape == img.shape:
        if hit[2] is not None:
            _cur().wait_event(hit[2])
            hit[0].record_stream(_cur())
        return hit[0]
    return _gram_launch(img, dims)[0]


# conv1's weight-gradient finalize INSIDE its launch (the last C workgroups to finish fold the partial rows; csrc/cnn2.hip):
# measured and not used -- 1.400 vs 1.393 ms at 64 x 20 (three alternating pairs on one box), 4.77 = 4.76 ms at 256 x 32: the
# write-through stores of 576 doubles per workgroup and the wait cost more than the 8 us launch they replace.
C1_TICKET = os.environ.get("MGGAN_CONV1_TICKET", "0") == "1"
BN_EXCHANGE_IN_LAUNCH = os.environ.get("MGGAN_BN_EXCHANGE", "fused") != "launch"
TAIL_RIDERS = 16  # spare doubles at the end of a gradient tail (mggan/parallel.py: DistContext.all_reduce_grads)
_RIDER_SRC = {}   # id(root) -> f64 tensor (TAIL_RIDERS) whose contents the root's next gradient tail carries as riders


def set_rider_src(root, src):
    """The next scene-CNN backward pass of `root` (sharded) copies `src` (TAIL_RIDERS doubles, written on a stream that
    backward pass is ordered behind) into the rider slots of its gradient tail."""
    if src is None:
        _RIDER_SRC.pop(id(root), None)
    else:
        _RIDER_SRC[id(root)] = src


def _cnn_tickets(owner, dev):
    t = owner.__dict__.get("_cnn_tickets")
    if t is None or t.device != dev:
        t = owner.__dict__["_cnn_tickets"] = torch.zeros(8, dtype=torch.int32, device=dev)
    return t


class GradTailDesc(ctypes.Structure):
    """include/mggan_hip.h: mggan_grad_tail_t"""
    _fields_ = [("tail", ctypes.c_void_p), ("n2", ctypes.c_long), ("gram", ctypes.c_void_p), ("W", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("stat", ctypes.c_void_p), ("dW", ctypes.c_void_p),
                ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("C", ctypes.c_int)]


class SceneAttentionFn(Function):
    """CNN (2 x Conv-BN-ReLU-MaxPool) + channel-softmax attention -> (B,64)  (cnn.py:109-282).
    Forward: conv1 + pooling decision -> conv2 -> attention (three launches, BatchNorm finalized by the producing
    kernel's last workgroup).  Backward: attention adjoint -> conv2 adjoint -> conv1 weight gradient + its f64 finalize.
    Saved for backward per image: the raw window extreme and its position (C x 1.25 KB), raw conv2 output (C KB)."""

    @staticmethod
    def forward(ctx, img, c1w, c1b, g1, be1, c2w, c2b, g2, be2, wa, ba, wb, bb, bn1, bn2, training, owner, sync, save,
                stat_updates=1, out_slot=None):
        img = img.contiguous()
        B, C = img.shape[0], c1w.shape[0]
        st = _s()
        dev = img.device
        tk = _cnn_tickets(owner, dev)
        grid = max(lib.mggan_cnn_grid(B), 1)
        xsel = _empty(B, C, 16, 16, like=img)  # raw conv1 output at the pooling position of every window
        code = torch.empty(B, C, 16, 16, dtype=torch.uint8, device=dev)
        part = torch.empty(grid, 2 * C, dtype=torch.float64, device=dev)
        fused = training and sync is None  # single GPU: the kernel's last workgroup finalizes BatchNorm itself
        mk = lambda: (_empty(C, like=img), _empty(C, like=img), _empty(2 * C, like=img))

        # sharded over peer-mapped arenas: the exchange of a BatchNorm point rides in the producing launch's last
        # workgroup (csrc/cnn2.hip: bn_finalize_block) -- no launch of its own (MGGAN_BN_EXCHANGE=launch: round 5's
        # fold + exchange + finalize kernel behind the producer)
        dc = getattr(sync, "devcomm", None) if (training and sync is not None) else None
        inl = dc is not None and BN_EXCHANGE_IN_LAUNCH
        comm_dev = dc.channel_dev() if inl else 0

        def bn_args(bn, gamma, beta, hw, k, out3, in_launch=False):
            return (tk.data_ptr() + 4 * k if (fused or in_launch) else 0, float(B) * hw, _p(gamma), _p(beta),
                    _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), float(bn.momentum), float(bn.eps),
                    stat_updates, _p(out3[0]), _p(out3[1]), _p(out3[2]))

        def finalize_unfused(bn, gamma, beta, hw, out3):
            """eval mode (running statistics) or sharded training (sums exchanged between the ranks first)"""
            dc = getattr(sync, "devcomm", None) if training else None
            if dc is not None:  # fold + peer-mapped exchange + finalize: ONE launch per exchange point
                sync.count_collective("bn2.forward" if hw == 16 * 16 else "bn1.forward")
                lib.mggan_bn_sync_finalize(*dc.channel_args(), _p(part), grid if B else 0, float(B) * hw, C, _p(gamma),
                                           _p(beta), _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked),
                                           float(bn.momentum), float(bn.eps), stat_updates, _p(out3[0]), _p(out3[1]),
                                           _p(out3[2]), st)
                return float("nan")  # the global count travels with the sums; the backward pass re-exchanges its own
            sums, n = None, float(B)
            if training:
                sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
                lib.mggan_bn_reduce_rows(_p(part), grid if B else 0, 2 * C, _p(sums), st)
                n = sync.all_reduce_stats(sums, float(B))
            lib.mggan_bn_finalize(_p(sums), n * hw, C, stat_updates if training else 0, _p(gamma), _p(beta),
                                  _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), float(bn.momentum),
                                  float(bn.eps), _p(out3[0]), _p(out3[1]), _p(out3[2]), st)
            return n * hw

        b1 = mk()
        dims = _PAD["dims"]  # padded batch: only the real images enter the convolutions and the statistics
        if dims is not None and not fused:
            raise RuntimeError("padded batches need the fused BatchNorm finalize (train mode, one GPU)")
        pd = _pad_ptr(dims)
        # sharded, global-batch statistics: layer 1's follow from the batch's GLOBAL Gram matrix and these weights -- no
        # exchange of their own (csrc/cnn2.hip: bn1_from_gram_block, run by conv1_pool's workgroup 0 on its way out).  The
        # Gram matrix is computed and all-reduced on a side stream from the start of the iteration; the FIRST scene-CNN pass
        # of the iteration would wait ~a conv1_pool for it and exchanges its own 2C sums instead (gram_for_pass).
        gram_g = gram_for_pass(img, sync) if (training and sync is not None) else None
        inl1 = inl and gram_g is None
        lib.mggan_conv1_pool(_p(img), B, C, _p(c1w), _p(c1b), _p(xsel), _p(code), _p(part),
                             *bn_args(bn1, g1, be1, 33 * 33, 0, b1, inl1), _p(gram_g), comm_dev if inl1 else 0, pd, st)
        if gram_g is not None:
            cnt1 = float("nan")
        elif inl1:
            sync.count_collective("bn1.forward")
            cnt1 = float("nan")
        else:
            cnt1 = float(B) * 33 * 33 if fused else finalize_unfused(bn1, g1, be1, 33 * 33, b1)
        sc1, sh1, stat1 = b1
        y2 = _empty(B, C, 16, 16, like=img)
        b2 = mk()
        lib.mggan_conv2_fwd2(_p(xsel), B, C, _p(sc1), _p(sh1), _p(c2w), _p(c2b), _p(y2), _p(part),
                             *bn_args(bn2, g2, be2, 16 * 16, 1, b2, inl), comm_dev, pd, st)
        if inl:
            sync.count_collective("bn2.forward")
            cnt2 = float("nan")
        else:
            cnt2 = float(B) * 16 * 16 if fused else finalize_unfused(bn2, g2, be2, 16 * 16, b2)
        sc2, sh2, stat2 = b2
        out, ld_o = _out(out_slot, B, 64, img)
        # what the adjoint needs of the conv2 output per pooled cell: the raw value that won the 2x2 window and its position
        # (cell-major, channel innermost: (B, 64, C))
        ysel = _empty(B, 64, C, like=img) if save else None
        ycode = torch.empty(B, 64, C, dtype=torch.uint8, device=dev) if save else None
        lib.mggan_scene_attention_fwd(_p(y2), B, C, _p(sc2), _p(sh2), _p(wa), _p(ba), _p(wb), _p(bb), _p(out), ld_o,
                                      _p(ysel), _p(ycode), pd, st)
        if save:
            if not training:
                raise RuntimeError("scene attention backward is only implemented for train-mode BatchNorm "
                                   "(the reference never differentiates in eval mode)")
            ctx.owner, ctx.sync, ctx.counts, ctx.dims = owner, sync, (cnt1, cnt2), dims
            ctx.save_for_backward(img, xsel, code, y2, sc1, sh1, stat1, sc2, sh2, stat2, c1w, c1b, g1, be1, c2w, c2b,
                                  g2, be2, wa, ba, wb, bb, ysel, ycode)
        return out

    @staticmethod
    def backward(ctx, dout):
        (img, xsel, code, y2, sc1, sh1, stat1, sc2, sh2, stat2, c1w, c1b, g1, be1, c2w, c2b, g2, be2, wa, ba, wb,
         bb, ysel, ycode) = ctx.saved_tensors
        root, sync = root_of(ctx.owner), ctx.sync
        cnt1, cnt2 = ctx.counts
        B, C = img.shape[0], c1w.shape[0]
        st = _s()
        dev = img.device
        tk = _cnn_tickets(ctx.owner, dev)
        fused = sync is None
        pd = _pad_ptr(ctx.dims) if ctx.dims is not None else 0
        dout, ld = _rows2d(dout)
        G2 = _empty(B, 64, C, like=img)  # one value per pooled cell and channel (its window position: ycode)
        rows2 = max(lib.mggan_scene_attention_grid(B), 1)
        part2 = torch.empty(rows2, 2 * C, dtype=torch.float64, device=dev)
        coef2 = _empty(3 * C, like=img)
        # the attention head's weight gradients come out of the same launch: one partial block per workgroup
        pf = lib.mggan_scene_attention_partial_floats(C)
        wpart = _empty(rows2 * pf, like=img)
        dc = getattr(sync, "devcomm", None)
        inl = dc is not None and BN_EXCHANGE_IN_LAUNCH  # the BatchNorm-2 adjoint exchange inside this launch's last workgroup
        if inl:
            sync.count_collective("bn2.backward")
        lib.mggan_scene_attention_bwd(_p(ysel), _p(ycode), B, C, _p(sc2), _p(sh2), _p(stat2), _p(wa), _p(ba), _p(wb), _p(bb), _p(dout),
                                      ld, _p(G2), _p(wpart), _p(part2), tk.data_ptr() + 8 if (fused or inl) else 0,
                                      float(B) * 16 * 16 if inl else cnt2, _p(g2), _p(coef2), root.grad_ptr(g2),
                                      root.grad_ptr(be2), dc.channel_dev() if inl else 0, pd, st)
        if B:
            p0 = wpart.data_ptr()
            adescs = ((p0, root.grad_ptr(wa), root.grad_ptr(ba), 32, C + 1, C),
                      (p0 + 4 * 32 * (C + 1), root.grad_ptr(wb), root.grad_ptr(bb), C, 33, 32))
            if _DEFER["on"] and not _SIDE["dirty"]:
                for k, (pp_, dw, db, M, Naug, lddw) in enumerate(adescs):
                    _queue_reduce(pp_, dw, db, M, Naug, 1, lddw, rows2, 1, pf, keep=(wpart,) if k == 0 else ())
            else:
                arr = (_ReduceDesc * 2)(*[_ReduceDesc(pp_, dw, db or None, 0, 0, M, Naug, 1, lddw, rows2, 1, pf, 0)
                                          for pp_, dw, db, M, Naug, lddw in adescs])
                lib.mggan_grad_reduce_multi(ctypes.addressof(arr), 2, st)
                if _SIDE["dirty"]:
                    _SIDE["keep"].append(wpart)

        def bn_bwd_sharded(part, nrows, gamma, beta, stat, cnt, coef, coefd, hw):
            dc = getattr(sync, "devcomm", None)
            if dc is not None:
                sync.count_collective("bn2.backward" if hw == 16 * 16 else "bn1.backward")
                lib.mggan_bn_bwd_sync_finalize(*dc.channel_args(), _p(part), nrows if B else 0, float(B) * hw, C, _p(gamma),
                                               _p(stat), _p(coef), _p(coefd), root.grad_ptr(gamma), root.grad_ptr(beta), st)
                return
            sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
            lib.mggan_bn_reduce_rows(_p(part), nrows if B else 0, 2 * C, _p(sums), st)
            local = sums.clone()
            sync.all_reduce_(sums)
            lib.mggan_bn_bwd_coef(_p(sums), _p(local), cnt, C, _p(gamma), _p(stat), _p(coef), _p(coefd),
                                  root.grad_ptr(gamma), root.grad_ptr(beta), st)

        if not fused and not inl:
            bn_bwd_sharded(part2, rows2, g2, be2, stat2, cnt2, coef2, None, 16 * 16)
        G1c = _empty(B, C, 16, 16, like=img)
        grid = lib.mggan_cnn_bwd_grid(B)
        nb = grid * (256 // (C * C)) * (C * C * 9 + C) * 4
        ws = _empty(nb // 4, like=img)
        part1 = torch.empty(max(grid, 1), 2 * C, dtype=torch.float64, device=dev)
        coef1 = _empty(3 * C, like=img)
        coefd1 = torch.empty(5 * C + 1, dtype=torch.float64, device=dev)
        defer = _DEFER["on"]
        pw, pb = root.grad_ptr(c2w), root.grad_ptr(c2b)
        lib.mggan_conv2_bwd(_p(xsel), B, C, _p(sc1), _p(sh1), _p(stat1), _p(y2), _p(G2), _p(ycode), _p(stat2),
                            _p(coef2), _p(c2w), _p(G1c), _p(part1), 0 if defer else pw, 0 if defer else pb,
                            _p(ws), nb, tk.data_ptr() + 12 if fused else 0, cnt1, _p(g1), _p(coef1), _p(coefd1),
                            root.grad_ptr(g1), root.grad_ptr(be1), pd, st)
        if defer:
            wl = C * C * 9 + C
            _queue_reduce(ws.data_ptr(), pw, 0, 1, C * C * 9, 0, C * C * 9, grid, 1, wl, keep=(ws,))
            _queue_reduce(ws.data_ptr() + 4 * C * C * 9, pb, 0, 1, C, 0, C, grid, 1, wl)
        # conv1: the image needs no gradient; dW1 from the sparse routed gradients and the batch's Gram matrix (f64);
        # db1 is identically zero in front of a train-mode BatchNorm (the slot is attached: the reference's set of
        # touched parameters includes it)
        root.grad_ptr(c1b)
        nbw = max(lib.mggan_cnn_grid(B), 1) * C * 36 * 8
        wsw = torch.empty(nbw // 8, dtype=torch.float64, device=dev)
        if fused:
            gram = _image_gram(img, ctx.dims)
            lib.mggan_conv1_wgrad(_p(img), B, C, _p(G1c), _p(code), _p(gram), _p(c1w), _p(c1b), _p(coefd1),
                                  root.grad_ptr(c1w), _p(wsw), nbw, tk.data_ptr() + 16 if C1_TICKET else 0, pd, st)
            return (None,) * 21
        # sharded: layer 1 has no exchange of its own.  This rank's raw sums -- conv1 weight gradient A, BatchNorm-1 adjoint
        # S1 / S2 -- are folded into one f64 tail that travels with the step's gradient all-reduce
        # (DistContext.all_reduce_grads); the finalize runs behind it with the global sums and the global Gram matrix
        gram = global_gram(img, sync)
        lib.mggan_conv1_wgrad(_p(img), B, C, _p(G1c), _p(code), 0, 0, 0, 0, 0, _p(wsw), nbw, 0, pd, st)
        # (+ TAIL_RIDERS spare doubles behind the CNN's sums: small per-rank sums of the trainer that ride along, e.g. the
        #  generator counts of the next step -- zero unless somebody writes them)
        tf = lib.mggan_conv1_tail_floats(C)
        tail = torch.empty(tf + TAIL_RIDERS, dtype=torch.float64, device=dev)
        # riders prepared ahead of the backward pass (the trainer's generator counts, set_rider_src) are copied in by the
        # fold itself; the root remembers that its tail carries them
        src = _RIDER_SRC.pop(id(root), None)
        root.__dict__["_rider_in_tail"] = src is not None
        lib.mggan_conv1_tail_fold(_p(wsw), max(lib.mggan_cnn_grid(B), 1) if B else 0, _p(part1), grid if B else 0, C,
                                  _p(tail), TAIL_RIDERS, _p(src), st)
        ptrs = (root.grad_ptr(c1w), root.grad_ptr(g1), root.grad_ptr(be1))

        def finalize(tail=tail, gram=gram, keep=(wsw, part1)):
            lib.mggan_conv1_tail_finalize(_p(tail), _p(gram), C, _p(c1w), _p(c1b), _p(g1), _p(stat1), ptrs[0], ptrs[1],
                                          ptrs[2], _s())

        # (4th entry: what the optimizer launch needs to run exchange + finalize itself, include/mggan_hip.h:
        #  mggan_grad_tail_t -- and the tensors that have to outlive it)
        desc = GradTailDesc(_p(tail), tail.numel(), _p(gram), _p(c1w), _p(c1b), _p(g1), _p(stat1), ptrs[0], ptrs[1], ptrs[2], C)
        root.__dict__.setdefault("_grad_tails", []).append((tail, finalize, tf, desc, (tail, gram, wsw, part1, stat1)))
        return (None,) * 21


# ------------------------------------------------------------------------------------------
class RolloutRows:
    """Row tables of one decoder launch: rows sorted by generator (host-side bookkeeping that
    replaces get_selection_indices + the index gather of standard.py:190-214)."""

    def __init__(self, gen, ped, slot, n_gens, b, device):
        import numpy as np

        gen = np.asarray(gen, np.int64).reshape(-1)
        R = gen.shape[0]
        order = np.argsort(gen, kind="stable").astype(np.int32)
        inv = np.empty(R, np.int32)
        inv[order] = np.arange(R, dtype=np.int32)
        seg = np.zeros(n_gens + 1, np.int32)
        seg[1:] = np.cumsum(np.bincount(gen, minlength=n_gens))
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(device)
        self.R, self.b, self.K = R, b, R // max(b, 1)
        self.row_gen, self.row_ped, self.row_slot = to(gen[order]), to(np.asarray(ped)[order]), to(np.asarray(slot)[order])
        self.row_pos, self.inv, self.seg = to(order), to(inv), to(seg)
        self.row_gen_pos = to(gen)


def empty_rollout_rows(b, K, n_gens, dev):
    """-> (RolloutRows with unfilled device tables, the block-count scratch of the bucketing launches)."""
    R = b * K
    rows = RolloutRows.__new__(RolloutRows)
    mk = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
    rows.R, rows.b, rows.K = R, b, K
    rows.row_gen, rows.row_ped, rows.row_slot, rows.row_pos, rows.inv = mk(R), mk(R), mk(R), mk(R), mk(R)
    rows.seg, rows.row_gen_pos = mk(n_gens + 1), mk(R)
    return rows, mk(16 * ((R + 1023) // 1024))


def device_rollout_rows(idx, n_gens):
    """RolloutRows built on the GPU from sampled generator ids (b, K) int64 -- no host round trip."""
    b, K = idx.shape
    rows, blk = empty_rollout_rows(b, K, n_gens, idx.device)
    idx = idx.contiguous()
    lib.mggan_bucket_rows(_p(idx), b, K, n_gens, _p(rows.row_gen), _p(rows.row_ped), _p(rows.row_slot),
                          _p(rows.row_pos), _p(rows.inv), _p(rows.seg), _p(rows.row_gen_pos), _p(blk), _s())
    return rows


_FUSED_LAYOUT = {}


def _fused_layout():
    """Offsets (in floats) inside one partial block of mggan_decoder_rollout_bwd_fused."""
    if not _FUSED_LAYOUT:
        v = [ctypes.c_int() for _ in range(8)]
        lib.mggan_decoder_bwd_fused_layout(*[ctypes.byref(x) for x in v])
        _FUSED_LAYOUT.update(zip(("wlen", "A", "bias", "W1", "b1", "W2", "b2", "W1s"), (x.value for x in v)))
    return _FUSED_LAYOUT


_SAVE_PADS = []


def _save_pads():
    if not _SAVE_PADS:
        a, b = ctypes.c_int(), ctypes.c_int()
        lib.mggan_decoder_save_pads(ctypes.byref(a), ctypes.byref(b))
        _SAVE_PADS.extend((a.value, b.value))
    return _SAVE_PADS


E2D_SHARED_MIN_ROWS = int(os.environ.get("MGGAN_E2D_SHARED_MIN_ROWS", "4096"))


class DecoderRolloutFn(Function):
    """enc_h_to_dec_h + per-generator RelativeDecoder rollouts for the selected rows
    (standard.py:227-265, common_modules.py:97-131)."""

    @staticmethod
    def forward(ctx, enc_h, soc, noise, xy0, dxdy0, rows, e2d_w, e2d_b, anchor, g0, n_gens, stride, T, owner, save):
        # `anchor` (= the first generator's W_hh) makes the per-generator weights visible to autograd
        enc_h, ld_enc = _rows2d(enc_h)
        soc, ld_soc = _rows2d(soc)
        noise, xy0, dxdy0 = noise.contiguous(), xy0.contiguous(), dxdy0.contiguous()
        b, EIN = enc_h.shape
        Z = noise.shape[-1]
        H, E, S = g0["w_hh"].shape[1], g0["emb_w"].shape[0], soc.shape[1]
        assert S == H, "social width must equal decoder_h_dim (both 32 by default)"
        R = rows.R
        st = _s()
        ctx.set_materialize_grads(False)
        psz = lib.mggan_lstm_prep_size(H, S, 1)
        # (every generator's tensors: `g0` names the first generator's, the others sit `stride` floats apart)
        prep = _prep_for(owner, tuple(owner.generator_parameters()), n_gens * psz, enc_h,
                         lambda prep: lib.mggan_lstm_fold(_p(g0["emb_w"]), _p(g0["emb_b"]), _p(g0["w_ih"]), _p(g0["b_ih"]),
                                                          _p(g0["b_hh"]), _p(g0["w_hh"]), _p(g0["w1"]), _p(g0["b1"]),
                                                          _p(g0["w2"]), _p(g0["b2"]), stride, n_gens, H, E, S, 1, _p(prep),
                                                          psz, _s())).view(n_gens, psz)
        mk = (lambda *s: _empty(*s, like=enc_h)) if save else (lambda *s: None)
        # tile-blocked saves (16-row tiles, every generator's last tile padded): gates, (c, h) with slot 0 = (0, h_0), ...
        tiles = -(-R // 16) + n_gens
        gt_pad, cs_pad = _save_pads()  # (tile records padded apart: csrc/lstm.hip DEC_GT_PAD)
        Gt, Cs = mk(tiles, T * H * 64 + gt_pad), mk(tiles, (T + 1) * H * 32 + cs_pad)
        # h0 = W_e2d [enc_h | noise] + b: the enc_h part is the pedestrian's, the same for its K rows and for every
        # generator -> once per pedestrian (Qe), the rows multiply their noise columns only and keep those (Nz)
        # (worth its extra launch from a few rows per pedestrian and a few thousand rows on)
        shared = (R >= 4 * b and R >= E2D_SHARED_MIN_ROWS and EIN % 16 == 0 and Z > 0 and e2d_w.shape[0] == 32
                  and enc_h.data_ptr() % 16 == 0 and ld_enc % 4 == 0)
        Qe = Nz = E2Din = None
        if shared:
            Qe = _empty(b, H, like=enc_h)
            lib.mggan_decoder_e2d_shared(_p(enc_h), ld_enc, b, EIN, _p(e2d_w), EIN + Z, _p(e2d_b), _p(Qe), st)
            Nz = mk(R, Z)
        else:
            E2Din = mk(R, EIN + Z)
        Din, Aact, SocR = mk(tiles, T, 16, 2), mk(tiles, T, 4, 16, 4), mk(R, S)
        out_abs, out_rel = _empty(T, R, 2, like=enc_h), _empty(T, R, 2, like=enc_h)
        lib.mggan_decoder_rollout_fwd(R, T, b, H, EIN, Z, _p(prep), psz, _p(rows.seg), n_gens, _p(rows.row_ped),
                                      _p(rows.row_slot), _p(rows.row_pos), _p(enc_h), ld_enc, _p(noise) or _p(enc_h), _p(soc), ld_soc,
                                      _p(xy0), _p(dxdy0), _p(e2d_w), _p(e2d_b), _p(out_abs), _p(out_rel), R, _p(Gt),
                                      _p(Cs), _p(Din), _p(Aact), _p(E2Din), _p(SocR), _p(Qe), _p(Nz), st)
        if save:
            ctx.shared, ctx.ld_enc = shared, ld_enc
            ctx.meta = (rows, g0, n_gens, stride, T, owner, (b, EIN, Z, H, E, S, psz))
            # `soc` is the last column block of `enc_h` itself (TrunkJoinFn hands out both): its gradient is folded into
            # d enc_h by the gather below instead of travelling as a second tensor that autograd would have to add
            ctx.soc_in_enc = (b > 0 and soc.data_ptr() == enc_h.data_ptr() + 4 * (EIN - S) and ld_soc == ld_enc)
            if shared:
                ctx.save_for_backward(e2d_w, e2d_b, prep, Gt, Cs, Din, Aact, Nz, SocR, enc_h)
            else:
                ctx.save_for_backward(e2d_w, e2d_b, prep, Gt, Cs, Din, Aact, E2Din, SocR)
        return out_abs, out_rel

    @staticmethod
    def backward(ctx, gabs, grel):
        if ctx.shared:
            e2d_w, e2d_b, prep, Gt, Cs, Din, Aact, Nz, SocR, enc_h = ctx.saved_tensors
        else:
            e2d_w, e2d_b, prep, Gt, Cs, Din, Aact, E2Din, SocR = ctx.saved_tensors
        rows, g0, n_gens, stride, T, owner, (b, EIN, Z, H, E, S, psz) = ctx.meta
        root = root_of(owner)
        R, Hh = rows.R, H // 2
        st = _s()
        dev = prep.device
        gabs = None if gabs is None else gabs.contiguous()
        grel = None if grel is None else grel.contiguous()
        mk = lambda *s: _empty(*s, like=prep)
        dH0, dQ, dEnc, dSocR = mk(R, H), mk(R, Hh), None if ctx.shared else mk(R, EIN), mk(R, S)
        # persistent workgroups per generator; each leaves one partial block of weight gradients
        # (16-row tiles; about two resident workgroups per CU, each looping over its generator's tiles).  The split of
        # the rows between the generators is a device-side draw: below the cap every generator gets room for ALL tiles
        # (an uneven draw then does not send workgroups through a second tile); a workgroup without a tile writes a
        # zero block and leaves
        NW = max(1, min(-(-R // 16) + 1, 512 // n_gens))
        lay = _fused_layout()
        wpart = mk(n_gens * NW, lay["wlen"])
        train_w = g0["w_hh"].requires_grad
        if train_w:
            gp = root.grad_ptr
            # attach (and zero on first touch) every generator's gradient slots BEFORE any kernel writes into
            # them: all decoders are in the graph, a generator without rows gets a zero gradient
            for p in owner.generator_parameters():
                gp(p)
            ptr = {k: gp(v) for k, v in g0.items()}
        lib.mggan_decoder_rollout_bwd_fused(n_gens, NW, T, H, EIN, Z, _p(rows.seg), _p(rows.row_pos), _p(g0["w_hh"]),
                                            _p(g0["w1"]), _p(g0["w2"]), stride, _p(e2d_w), _p(prep), psz, _p(Gt), _p(Cs),
                                            _p(Din), _p(Aact), _p(gabs), _p(grel), R, _p(dH0), _p(dQ),
                                            _p(dEnc), _p(dSocR), _p(wpart), _p(SocR) if train_w else 0, st)
        if train_w:
            ng, wl, P = n_gens, lay["wlen"], wpart.data_ptr()
            dprep = mk(n_gens, 12 * H)  # scratch: the reduction stores into it (has_bias bit 1)
            now = (_ReduceDesc * 2)(
                _ReduceDesc(P + 4 * lay["A"], dprep.data_ptr(), None, 12 * H, 0, 4 * H, 2, 2, 2, NW, ng, wl, 0),
                _ReduceDesc(P + 4 * lay["bias"], dprep.data_ptr() + 4 * 8 * H, None, 12 * H, 0, 1, 4 * H, 2, 4 * H, NW, ng,
                            wl, 0))
            unfold = lambda: lib.mggan_lstm_unfold_grads(_p(g0["emb_w"]), _p(g0["emb_b"]), _p(g0["w_ih"]), ptr["emb_w"],
                                                         ptr["emb_b"], ptr["w_ih"], ptr["b_ih"], ptr["b_hh"], stride, ng,
                                                         H, E, _p(dprep), 12 * H, _s())
            if _DEFER["on"]:  # off the critical chain: with the batched reductions at the end of the backward pass
                _note_branch_partials()
                _DEFER["descs"].extend(now)
                _DEFER["after"].append(unfold)
                _DEFER["keep"].extend((dprep, wpart, now))
            else:
                lib.mggan_grad_reduce_multi(ctypes.addressof(now), 2, st)
                unfold()
            # (W1s = dW1[:, H:], the social half of hidden2pos: summed inside the launch, tile by tile -- as a grouped GEMM
            #  over all R rows behind the backward pass it was a launch of its own)
            parts = [(0, ptr["w_hh"], 4 * H, H, H), (lay["W1"], ptr["w1"], Hh, H, H + S), (lay["b1"], ptr["b1"], 1, Hh, Hh),
                     (lay["W2"], ptr["w2"], 2, Hh, Hh), (lay["b2"], ptr["b2"], 1, 2, 2),
                     (lay["W1s"], ptr["w1"] + 4 * H, Hh, S, H + S)]
            if _DEFER["on"]:
                for i, (off, dst, M, N, ld) in enumerate(parts):
                    _queue_reduce(P + 4 * off, dst, 0, M, N, 0, ld, NW, ng, wl, stride, 0, keep=(wpart,) if i == 0 else ())
            else:
                arr = (_ReduceDesc * len(parts))(*[_ReduceDesc(P + 4 * off, dst, None, stride, 0, M, N, 0, ld, NW, ng, wl, 0)
                                                  for off, dst, M, N, ld in parts])
                lib.mggan_grad_reduce_multi(ctypes.addressof(arr), len(parts), st)
        dQe = None
        d_enc = d_soc = None
        # shared h0, d enc_h wanted, the social block inside enc_h (the trainer's case): the whole per-pedestrian tail -- dH0 and
        # dSocR folded over a pedestrian's K rows, d enc_h = dQe W_e2d[:, :EIN] -- is ONE launch (it was three in a row)
        fused_tail = (ctx.shared and ctx.needs_input_grad[0] and H == 32 and S <= 32 and
                      (not ctx.needs_input_grad[1] or ctx.soc_in_enc))
        if fused_tail:
            dQe, d_enc = mk(b, H), mk(b, EIN)
            with_soc = ctx.needs_input_grad[1] and ctx.soc_in_enc
            lib.mggan_rollout_ped_adjoint(_p(dH0), _p(dSocR) if with_soc else None, _p(rows.inv), _p(e2d_w), EIN + Z, _p(dQe),
                                          _p(d_enc), EIN, b, rows.K, EIN, S if with_soc else 0, st)
        elif ctx.shared and (e2d_w.requires_grad or ctx.needs_input_grad[0]):
            # adjoint of the per-pedestrian part of h0: dH0 folded over the K rows of a pedestrian, then ONE product
            # (weight gradient over b rows instead of R, d enc_h over b rows instead of R)
            dQe = mk(b, H)
            lib.mggan_gather_sum(_p(dH0), H, _p(rows.inv), _p(dQe), H, b, rows.K, H, 0, st)
        if e2d_w.requires_grad:
            pw, pb = root.grad_ptr(e2d_w), root.grad_ptr(e2d_b)
            if ctx.shared:
                with side_stream(dQe, enc_h, dH0, Nz):
                    wgrad(dQe, H, enc_h, ctx.ld_enc, pw, EIN + Z, pb, b, EIN, H)
                    wgrad(dH0, H, Nz, Z, pw + 4 * EIN, EIN + Z, 0, R, Z, H)
            else:
                with side_stream(dH0, E2Din):
                    wgrad(dH0, H, E2Din, EIN + Z, pw, EIN + Z, pb, R, EIN + Z, H)
        if fused_tail:
            return (d_enc, None) + (None,) * 13
        if ctx.needs_input_grad[0]:
            d_enc = mk(b, EIN)
            if ctx.shared:
                lib.mggan_linear_bwd_data(_p(dQe), H, _p(e2d_w), EIN + Z, _p(d_enc), EIN, b, EIN, H, 0, 0, 0, ACT_NONE, 0.0,
                                          st)
            else:
                lib.mggan_gather_sum(_p(dEnc), EIN, _p(rows.inv), _p(d_enc), EIN, b, rows.K, EIN, 0, st)
        if ctx.needs_input_grad[1]:
            if ctx.soc_in_enc and d_enc is not None:
                lib.mggan_gather_sum(_p(dSocR), S, _p(rows.inv), d_enc.data_ptr() + 4 * (EIN - S), EIN, b, rows.K, S, 1, st)
            else:
                d_soc = mk(b, S)
                lib.mggan_gather_sum(_p(dSocR), S, _p(rows.inv), _p(d_soc), S, b, rows.K, S, 0, st)
        return (d_enc, d_soc) + (None,) * 13


# ------------------------------------------------------------------------------------------
class DAssembleFn(Function):
    """classifier_inp rows k*b+ped = [soc | in_enc | pred_enc | scene]
    (discriminators.py:141,179-196).  soc_all=False: soc0 (b rows) belongs to sample block 0 only and the other
    blocks get zeros (list-repeat of seq_start_end); soc_all=True: soc0 has K*b rows (K independent
    single-sample passes batched into one)."""

    @staticmethod
    def forward(ctx, soc0, in_enc, pred_enc, scene, K, soc_all=False):
        soc0, in_enc, pred_enc, scene = (t.contiguous() for t in (soc0, in_enc, pred_enc, scene))
        b = in_enc.shape[0]
        ws, wi, wp, wc = soc0.shape[1], in_enc.shape[1], pred_enc.shape[1], scene.shape[1]
        soc_all = int(soc_all)  # 0: block 0 only, 1: soc0 has K*b rows, 2: soc0 (b rows) broadcast to every block
        assert soc0.shape[0] == (K * b if soc_all == 1 else b), (soc0.shape, K, b, soc_all)
        X = _empty(K * b, ws + wi + wp + wc, like=in_enc)
        lib.mggan_d_assemble_fwd(b, K, ws, wi, wp, wc, soc_all, _p(soc0), _p(in_enc), _p(pred_enc), _p(scene),
                                 _p(X), _s())
        ctx.dims = (b, K, ws, wi, wp, wc, soc_all)
        return X

    @staticmethod
    def backward(ctx, dX):
        b, K, ws, wi, wp, wc, soc_all = ctx.dims
        dX = dX.contiguous()
        need = ctx.needs_input_grad
        mk = lambda n, r, c: _empty(r, c, like=dX) if n else None
        dsoc, din = mk(need[0], K * b if soc_all == 1 else b, ws), mk(need[1], b, wi)
        dpred, dsc = mk(need[2], K * b, wp), mk(need[3], b, wc)
        lib.mggan_d_assemble_bwd(b, K, ws, wi, wp, wc, soc_all, _p(dX), _p(dsoc), _p(din), _p(dpred), _p(dsc),
                                 _s())
        return dsoc, din, dpred, dsc, None, None


class StepsToRowsFn(Function):
    """(T, n, 2) time-major steps -> (n, 2T) rows and back (discriminators.py:129-131 permute + reshape), one launch
    per direction."""

    @staticmethod
    def forward(ctx, a):
        T, n = a.shape[0], a.shape[-2] * (a.shape[1] if a.dim() == 4 else 1)
        ctx.shape = a.shape
        return steps_to_rows(a.reshape(T, n, 2))

    @staticmethod
    def backward(ctx, g):
        g, ld = _rows2d(g)
        n, T = g.shape[0], g.shape[1] // 2
        out = _empty(T, n, 2, like=g)
        lib.mggan_rows_to_steps(_p(g), ld, T, n, _p(out), _s())
        return out.view(ctx.shape)


def _d_dims(D, in_w, scene_w):
    pe, Wat = D.pred_encoder, D.social.attention.W
    Hs, w_pe = Wat.weight.shape[0], pe[2].weight.shape[0]
    assert Hs == in_w + w_pe == Wat.weight.shape[1], "social width must equal h_dim (discriminators.py:58-60)"
    return Hs, in_w, w_pe, scene_w


class DRowsBodyFn(Function):
    """First node of the discriminator's pass over K*b (pedestrian, sample) rows (discriminators.py:113-196, pool_type
    'sways', unmasked): pred_encoder -> X = [soc | in_enc | pred_enc | (scene)] -> social attention over the sample
    blocks that have social features.  X is a single (K*b, 192) buffer every producer writes into in place (column
    offset + row stride): no torch.cat / .repeat / slice copies, no gradient adds between autograd nodes.  The scene
    block is filled by the second node (DRowsHeadsFn), so that the scene gradient leaves the backward pass before this
    node's adjoint (social attention, pred_encoder) runs.
      soc_blocks: number of leading sample blocks with social features (1: the list-repeat quirk of one K-sample call,
                  SURVEY A.1; K: K independent single-sample calls batched -- the real/fake pair pass of a D step)"""

    @staticmethod
    def forward(ctx, in_enc, pred, pred2, anchor, D, tb, K, soc_blocks, xy_last, dxdy_last, xy_mod, w_sc, save, lean=False):
        in_enc, ld_in = _rows2d(in_enc)
        b = in_enc.shape[0]
        R = K * b
        pe = D.pred_encoder
        fc, Wat = D.social.feature_embedder.fc, D.social.attention.W
        Hs, w_in, w_pe, w_sc = _d_dims(D, in_enc.shape[1], w_sc)
        W = Hs + w_in + w_pe + w_sc
        c_in, c_pe, c_sc = Hs, Hs + w_in, Hs + w_in + w_pe
        st = _s()
        T = pred.shape[0]
        X = _empty(R, W, like=in_enc)
        # pred_encoder, last layer straight into its column block of X
        spec_pe = ((ACT_LEAKY, 0.2), (ACT_NONE, 0.0))
        Wpe, bpe = (pe[0].weight, pe[2].weight), (pe[0].bias, pe[2].bias)
        if pred_encoder_kernel_ok(pe, T, w_pe):
            # from the time-major steps in one launch (csrc/dheads.hip) instead of steps_to_rows + a chain launch
            a = pred.reshape(T, -1, 2).contiguous()
            n = a.shape[1]
            a2 = None if pred2 is None else pred2.reshape(T, n, 2).contiguous()
            assert n * (1 if pred2 is None else 2) == R, (a.shape, K, b)
            x = _empty(R, 2 * T, like=in_enc) if save else None
            h_pe = _empty(R, 64, like=in_enc) if save else None
            lib.mggan_pred_encoder_fwd(_p(a), _p(a2), T, n, n, R, _p(Wpe[0]), _p(bpe[0]), _p(Wpe[1]), _p(bpe[1]), _p(X), W, c_pe,
                                       _p(h_pe), _p(x), st)
            outs_pe = [h_pe, None]
        else:
            x = steps_to_rows(pred, pred2)  # predictions as rows (K*b, 2T)
            assert x.shape[0] == R, (x.shape, K, b)
            outs_pe = _chain_fwd(x, 2 * T, R, spec_pe, Wpe, bpe, save, out_into=(alias_cols(X, c_pe, c_sc), W))
        # broadcast in_enc into the sample blocks, clear the social block of the blocks without social features
        # (lean heads, dheads_lean_ok: only block 0 is ever read outside its pred_enc columns)
        lib.mggan_d_rows_fill(b, 1 if lean else K, soc_blocks, Hs, c_in, w_in, c_sc, 0, _p(in_enc), ld_in, 0, 0, _p(X), W, st)
        # social attention: h = X[:, in_enc | pred_enc] of the first soc_blocks*b rows, S -> X[:, soc block]
        nsoc = soc_blocks * b
        xy_last, dxdy_last = xy_last.contiguous(), dxdy_last.contiguous()
        sw = (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias, Wat.weight, Wat.bias)
        soc_saved = _social_fwd(xy_last, dxdy_last, _p(X) + 4 * c_in, W, nsoc, w_in + w_pe, tb, *sw, _p(X), W, save, xy_mod,
                                X)
        if save:
            ctx.cfg = (D, tb, K, soc_blocks, b, T, W, (Hs, w_in, w_pe, w_sc),
                       (Wpe[0].requires_grad, fc[0].weight.requires_grad, fc[4].weight.requires_grad), pred.shape,
                       None if pred2 is None else pred2.shape, xy_mod)
            ctx.save_for_backward(X, x, outs_pe[0], xy_last, dxdy_last, *soc_saved)
        return X

    @staticmethod
    def backward(ctx, dX):
        (D, tb, K, soc_blocks, b, T, W, (Hs, w_in, w_pe, w_sc), train, pshape, p2shape, xy_mod) = ctx.cfg
        sv = ctx.saved_tensors
        X, x, h_pe, xy_last, dxdy_last = sv[:5]
        soc_saved = sv[5:]
        R = K * b
        c_in, c_pe, c_sc = Hs, Hs + w_in, Hs + w_in + w_pe
        pe = D.pred_encoder
        fc, Wat = D.social.feature_embedder.fc, D.social.attention.W
        train_pe, train_s1, train_s3 = train
        st = _s()
        need_in, need_pred, need_pred2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if dX.stride(1) != 1 or (R > 1 and dX.stride(0) != W):
            dX = dX.contiguous()
        elif dX.data_ptr() % 16:  # contiguous() is a no-op on a contiguous view at an odd storage offset
            dX = dX.clone()
        # social attention: dS = dX[:, soc block], dh ADDED into dX[:, in_enc | pred_enc] in place (dX comes from
        # DRowsHeadsFn.backward and has no other reader)
        nsoc = soc_blocks * b
        sw = (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias, Wat.weight, Wat.bias)
        _social_bwd(soc_saved, xy_last, dxdy_last, xy_mod, _p(X) + 4 * c_in, W, X, nsoc, w_in + w_pe, tb, *sw, _p(dX), W,
                    _p(dX) + 4 * c_in, W, 1, train_s1, train_s3, D.social, X)
        din = None
        if need_in:  # adjoint of the broadcast
            din = _empty(b, w_in, like=X)
            lib.mggan_d_rows_reduce(b, K, c_in, w_in, c_sc, 0, _p(dX), W, _p(din), w_in, 0, 0, st)
        dpred = dpred2 = None
        if need_pred or need_pred2 or train_pe:  # pred_encoder adjoint: dy = dX[:, pred_enc block]
            spec_pe = ((ACT_LEAKY, 0.2), (ACT_NONE, 0.0))
            dx = _chain_bwd(alias_cols(dX, c_pe, c_sc), W, x, 2 * T, R, (h_pe, alias_cols(X, c_pe, c_sc)), spec_pe,
                            (pe[0].weight, pe[2].weight), (pe[0].bias, pe[2].bias), need_pred or need_pred2, train_pe,
                            pe[0], ld_last=W)
            if need_pred or need_pred2:
                n1 = R if p2shape is None else R // 2
                if need_pred:
                    dpred = _empty(T, n1, 2, like=X)
                    lib.mggan_rows_to_steps(_p(dx), 2 * T, T, n1, _p(dpred), st)
                    dpred = dpred.view(pshape)
                if need_pred2:
                    dpred2 = _empty(T, R - n1, 2, like=X)
                    lib.mggan_rows_to_steps(_p(dx) + 4 * n1 * 2 * T, 2 * T, T, R - n1, _p(dpred2), st)
                    dpred2 = dpred2.view(p2shape)
        return (din, dpred, dpred2) + (None,) * 11


# from this many rows on, both discriminator heads run as one weight-stationary launch (csrc/dheads.hip)
# (re-measured at the configs[0] shape -- 204 pair-pass rows, 2,040 rows in the generator step: 0.91 ms per iteration
#  with the round's first thresholds 8,192 / 2,048, 0.85 ms with these; no difference at configs[1])
DHEADS_MIN_ROWS = int(os.environ.get("MGGAN_DHEADS_MIN_ROWS", "1024"))
DHEADS_PAIR_MIN_ROWS = int(os.environ.get("MGGAN_DHEADS_PAIR_MIN_ROWS", "64"))


def dheads_lean_ok(D, in_enc, scene, K, soc_blocks, row0):
    """True when the K-sample row pass can take the lean heads (csrc/dheads.hip: P per pedestrian + the pred_enc
    product per row): a frozen discriminator whose history / scene features carry no gradient (the generator step, the
    evaluation passes), social features on sample block 0 only, the default widths."""
    if os.environ.get("MGGAN_DHEADS_LEAN", "1") == "0":
        return False
    b = in_enc.shape[0]
    if not (D.gan_type == "mgan" and row0 == 0 and soc_blocks == 1 and K >= 2 and K * b >= DHEADS_MIN_ROWS):
        return False
    d0, r, pe = D.discs[0], D.gen_id_reconstructor, D.pred_encoder
    if torch.is_grad_enabled() and (in_enc.requires_grad or scene.requires_grad or d0[0].weight.requires_grad
                                    or d0[2].weight.requires_grad or r[0].weight.requires_grad or r[2].weight.requires_grad):
        return False
    Hs = D.social.attention.W.weight.shape[0]
    return (tuple(d0[0].weight.shape) == (96, 192) and tuple(r[0].weight.shape) == (96, 192) and r[2].weight.shape[0] <= 15
            and Hs == 64 and in_enc.shape[1] == 32 and pe[2].weight.shape[0] == 32 and scene.shape[1] == 64)


def pred_encoder_kernel_ok(pe, T, w_pe):
    """The discriminator's pred_encoder in its default shape (2*12 -> 64 -> 32): mggan_pred_encoder_fwd applies."""
    return (os.environ.get("MGGAN_PRED_ENC_KERNEL", "1") != "0" and T == 12 and w_pe == 32
            and tuple(pe[0].weight.shape) == (64, 24) and tuple(pe[2].weight.shape) == (32, 64))


def d_rows_lean_ok(D, in_enc, scene, pred, K, soc_blocks, row0):
    """dheads_lean_ok and, on top, everything DRowsLeanFn assumes: the whole discriminator frozen (pred_encoder and social
    attention included), the default prediction length."""
    if os.environ.get("MGGAN_DROWS_LEAN", "1") == "0" or not dheads_lean_ok(D, in_enc, scene, K, soc_blocks, row0):
        return False
    pe, soc = D.pred_encoder, D.social
    if torch.is_grad_enabled() and (pe[0].weight.requires_grad or pe[2].weight.requires_grad
                                    or any(q.requires_grad for q in soc.parameters())):
        return False
    return pred.shape[0] == 12 and tuple(pe[0].weight.shape) == (64, 24) and tuple(pe[2].weight.shape) == (32, 64)


# Blocks 1 .. K-1 of the frozen discriminator's row pass beside block 0's chain of small launches (branch stream 3), forward
# and backward: -12 us per iteration at 25,600 rows (three alternating pairs on one box: 1.450-1.455 vs 1.463-1.466 ms), +15 us
# at 163,840 rows, where the big launch fills the chip by itself (5.129-5.138 vs 5.108-5.124 ms).  MGGAN_DROWS_BRANCH=0 / 1
# forces it off / on.
_DROWS_KNOB = os.environ.get("MGGAN_DROWS_BRANCH", "")
DROWS_BRANCH_MAX_ROWS = 65536


def _drows_branch(R):
    if _DROWS_KNOB in ("0", "1"):
        return _DROWS_KNOB == "1"
    return R <= DROWS_BRANCH_MAX_ROWS


_DROWS_B0_BRANCH = os.environ.get("MGGAN_DROWS_B0", "branch") != "main"  # (A/B knob: block 0's first half on this stream)


class DRowsLeanFn(Function):
    """The K-sample row pass of a FROZEN discriminator (the generator step, the evaluation passes) in its lean form
    (discriminators.py:113-219, pool_type 'sways', unmasked):
      sample block 0 (the only rows with social features): pred_encoder -> X0 = [soc | in_enc | pred_enc |
        scene] (b, 192) -> social attention -> both heads, through the kernels of the generic path;
      blocks 1 .. K-1: ONE launch from the predicted steps to both head outputs (csrc/dheads.hip: pred_encoder ->
        P[ped] + W1[:, pred_enc] pred_enc -> heads), one launch back to the gradient of the steps.
    -> (score (K*b, 1), id logits (K*b, g)); the only gradient: d pred."""

    @staticmethod
    def forward(ctx, in_enc, scene, pred, D, tb, K, xy_last, dxdy_last, save):
        in_enc, ld_in = _rows2d(in_enc)
        scene, ld_sc = _rows2d(scene)
        b = in_enc.shape[0]
        R, T = K * b, pred.shape[0]
        pshape = pred.shape
        pred = pred.reshape(T, R, 2).contiguous()
        st = _s()
        ctx.set_materialize_grads(False)
        pe, d0, r = D.pred_encoder, D.discs[0], D.gen_id_reconstructor
        fc, Wat = D.social.feature_embedder.fc, D.social.attention.W
        W, c_in, c_pe, c_sc = 192, 64, 96, 128
        g, act = r[2].weight.shape[0], D._out_act()
        # Block 0 (the b rows with social features: pred_encoder -> social attention -> heads, six small launches in a row)
        # and blocks 1 .. K-1 (one launch) meet only in the outputs: what the big launch needs -- P, the per-pedestrian part of
        # the heads' first layers, from the in_enc and scene columns of X -- is produced first, then the big launch goes to
        # branch stream 3 beside block 0's chain (configs[1]: 25 us of the 160 us between the rollout and its adjoint).
        # (the caller may have had the scene CNN write its features into the scene columns of a classifier-input buffer of its
        #  own -- discriminators.history_context(scene_out=...): that buffer is X, and the broadcast launch below is skipped)
        X = getattr(scene, "_mggan_X", None)
        scene_in_place = (X is not None and tuple(X.shape) == (b, W) and X.stride(0) == W
                          and scene.data_ptr() == X.data_ptr() + 4 * c_sc and ld_sc == W)
        if not scene_in_place:
            X = _empty(b, W, like=in_enc)
        Wpe, bpe = (pe[0].weight, pe[2].weight), (pe[0].bias, pe[2].bias)
        wa = (d0[0].weight, d0[0].bias, d0[2].weight, d0[2].bias)
        wb = (r[0].weight, r[0].bias, r[2].weight, r[2].bias)
        lib.mggan_d_rows_fill(b, 1, 1, c_in, c_in, c_pe - c_in, c_sc, 0, _p(in_enc), ld_in, 0, 0, _p(X), W, st)
        # ---- block 0 (rows 0 .. b-1), first half: pred_encoder and the social attention need the history encoding and the
        # predictions only -- they go to branch stream 3, BEFORE this stream waits for the scene CNN's branch (round 6: the
        # generator step no longer joins that branch ahead of the discriminator pass), and run beside the per-pedestrian part
        # of the heads and the one big launch of blocks 1 .. K-1 on this stream (they write the pred_enc / soc columns of X,
        # this stream the scene columns and P) ----
        x0 = _empty(b, 2 * T, like=in_enc) if save else None
        h_pe0 = _empty(b, 64, like=in_enc) if save else None
        xy_last, dxdy_last = xy_last.contiguous(), dxdy_last.contiguous()
        sw = (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias, Wat.weight, Wat.bias)
        with branch(3) if (_drows_branch(R) and _DROWS_B0_BRANCH) else contextlib.nullcontext():
            lib.mggan_pred_encoder_fwd(_p(pred), 0, T, R, b, b, _p(Wpe[0]), _p(bpe[0]), _p(Wpe[1]), _p(bpe[1]), _p(X), W, c_pe,
                                       _p(h_pe0), _p(x0), _s())
            soc_saved = _social_fwd(xy_last, dxdy_last, _p(X) + 4 * c_in, W, b, c_sc - c_in, tb, *sw, _p(X), W, save, 0, X)
        outs_pe = [h_pe0, None]
        join_branch(scene, which=0)  # the scene CNN's branch has to be there now
        if not scene_in_place:
            lib.mggan_d_rows_fill(b, 1, 1, 0, 0, 0, c_sc, W - c_sc, 0, 0, _p(scene), ld_sc, _p(X), W, st)
        P = _empty(b, W, like=X)
        lib.mggan_dheads_shared(_p(X), W, b, c_in, c_sc, _p(wa[0]), _p(wa[1]), _p(wb[0]), _p(wb[1]), _p(P), st)
        ya, yb = _empty(R, 1, like=X), _empty(R, g, like=X)
        # ---- blocks 1 .. K-1 (rows b .. R-1 of ya / yb): one launch ----
        mask = torch.empty(-(-(R - b) // 16) * 64, dtype=torch.int64, device=X.device) if save else None
        lib.mggan_d_rows_lean_fwd(_p(pred), T, b, R, b, g, act, _p(Wpe[0]), _p(bpe[0]), _p(Wpe[1]), _p(bpe[1]), _p(P), c_pe,
                                  _p(wa[0]), _p(wa[2]), _p(wa[3]), _p(wb[0]), _p(wb[2]), _p(wb[3]), _p(mask), _p(ya), _p(yb),
                                  st)
        join_branch(X, x0, h_pe0, *[t for t in soc_saved if torch.is_tensor(t)], which=3)
        # ---- block 0, second half: both heads (the scene columns are in place) ----
        ha = _empty(b, 96, like=X) if save else None
        hb = _empty(b, 96, like=X) if save else None
        lib.mggan_dheads_fwd(_p(X), W, b, g, act, _p(wa[0]), _p(wa[1]), _p(wa[2]), _p(wa[3]), _p(wb[0]), _p(wb[1]),
                             _p(wb[2]), _p(wb[3]), _p(ha), _p(hb), _p(ya), _p(yb), st)
        if save:
            ctx.cfg = (D, tb, K, b, T, pshape, g, act)
            ctx.save_for_backward(X, x0, outs_pe[0], xy_last, dxdy_last, ha, ya, hb, mask, *soc_saved)
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb):
        D, tb, K, b, T, pshape, g, act = ctx.cfg
        sv = ctx.saved_tensors
        X, x0, h_pe, xy_last, dxdy_last, ha, ya, hb, mask = sv[:9]
        soc_saved = sv[9:]
        R = K * b
        st = _s()
        pe, d0, r = D.pred_encoder, D.discs[0], D.gen_id_reconstructor
        fc, Wat = D.social.feature_embedder.fc, D.social.attention.W
        W, c_in, c_pe, c_sc = 192, 64, 96, 128
        dya = torch.zeros(R, 1, dtype=F32, device=ya.device) if dya is None else dya.reshape(R, 1).contiguous()
        dyb = torch.zeros(R, g, dtype=F32, device=ya.device) if dyb is None else dyb.reshape(R, g).contiguous()
        dpred = _empty(T, R, 2, like=ya)
        # blocks 1 .. K-1 (rows b .. R-1 of dpred) on branch stream 3 beside block 0's chain (rows 0 .. b-1)
        with branch(3) if _drows_branch(R) else contextlib.nullcontext():
            lib.mggan_d_rows_lean_bwd(_p(dya), _p(dyb), _p(ya), _p(mask), T, b, R, g, act, _p(pe[0].weight), _p(pe[2].weight),
                                      c_pe, _p(d0[0].weight), _p(d0[2].weight), _p(r[0].weight), _p(r[2].weight), _p(dpred),
                                      _s())
        # block 0: heads -> social attention (dh added into the in_enc | pred_enc columns) -> pred_encoder
        dX = _empty(b, W, like=ya)
        lib.mggan_dheads_bwd_data(_p(dya), _p(dyb), _p(ya), _p(ha), _p(hb), b, g, act, _p(d0[0].weight), _p(d0[2].weight),
                                  _p(r[0].weight), _p(r[2].weight), _p(dX), W, st)
        sw = (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias, Wat.weight, Wat.bias)
        _social_bwd(soc_saved, xy_last, dxdy_last, 0, _p(X) + 4 * c_in, W, X, b, c_sc - c_in, tb, *sw, _p(dX), W,
                    _p(dX) + 4 * c_in, W, 1, False, False, D.social, X)
        spec_pe = ((ACT_LEAKY, 0.2), (ACT_NONE, 0.0))
        dx0 = _chain_bwd(alias_cols(dX, c_pe, c_sc), W, x0, 2 * T, b, (h_pe, alias_cols(X, c_pe, c_sc)), spec_pe,
                         (pe[0].weight, pe[2].weight), (pe[0].bias, pe[2].bias), True, False, pe[0], ld_last=W)
        lib.mggan_rows_to_steps_n(_p(dx0), 2 * T, T, b, R, _p(dpred), st)
        join_branch(dpred, which=3)
        return (None, None, dpred.view(pshape)) + (None,) * 6


class DRowsHeadsFn(Function):
    """Second node of the row pass (discriminators.py:186-219): broadcasts the scene features into their column block
    of X (in place) and runs the score head over all rows and the generator-id head over rows [row0, K*b).
    Backward: both heads' input gradients land in ONE dX buffer (the second head accumulates inside its launch), the
    scene gradient is the K-sum of its block.  -> (score (K*b, 1), id logits (K*b - row0, g) | None)"""

    @staticmethod
    def forward(ctx, X, scene, anchor, D, K, row0, save, lean=False):
        scene, ld_sc = _rows2d(scene)
        b, w_sc = scene.shape
        R, W = X.shape
        assert R == K * b
        c_sc = W - w_sc
        st = _s()
        ctx.set_materialize_grads(False)
        lib.mggan_d_rows_fill(b, 1 if lean else K, K, 0, 0, 0, c_sc, w_sc, 0, 0, _p(scene), ld_sc, _p(X), W, st)
        d0 = D.discs[0]
        if lean:
            # block 0 (the rows with social features) through the full-width kernels, the other K-1 blocks from the
            # per-pedestrian part P and their pred_enc columns
            r = D.gen_id_reconstructor
            g, act = r[2].weight.shape[0], D._out_act()
            c_in, c_pe = 64, 96
            wa = (d0[0].weight, d0[0].bias, d0[2].weight, d0[2].bias)
            wb = (r[0].weight, r[0].bias, r[2].weight, r[2].bias)
            P = _empty(b, 192, like=X)
            lib.mggan_dheads_shared(_p(X), W, b, c_in, c_sc, _p(wa[0]), _p(wa[1]), _p(wb[0]), _p(wb[1]), _p(P), st)
            ha = _empty(b, 96, like=X) if save else None
            hb = _empty(b, 96, like=X) if save else None
            ya, yb = _empty(R, 1, like=X), _empty(R, g, like=X)
            lib.mggan_dheads_fwd(_p(X), W, b, g, act, _p(wa[0]), _p(wa[1]), _p(wa[2]), _p(wa[3]), _p(wb[0]), _p(wb[1]),
                                 _p(wb[2]), _p(wb[3]), _p(ha), _p(hb), _p(ya), _p(yb), st)
            mask = torch.empty(-(-(R - b) // 16) * 64, dtype=torch.int64, device=X.device) if save else None
            lib.mggan_dheads_lean_fwd(_p(X), W, c_pe, b, R, b, g, act, _p(P), _p(wa[0]), _p(wa[2]), _p(wa[3]), _p(wb[0]),
                                      _p(wb[2]), _p(wb[3]), _p(mask), _p(ya), _p(yb), st)
            if save:
                ctx.cfg = (D, K, row0, b, W, w_sc, True, False, False, "lean")
                ctx.save_for_backward(ha, ya, hb, mask)
            return ya, yb
        spec_a = ((ACT_LEAKY, 0.2), (D._out_act(), 0.0))
        Wa, ba = (d0[0].weight, d0[2].weight), (d0[0].bias, d0[2].bias)
        mgan = D.gan_type == "mgan"
        outs_b, Wb = [None, None], None
        r = D.gen_id_reconstructor if mgan else None
        # many rows, both heads over all of them (the generator step's K*b rows): ONE weight-stationary launch
        # (row0 > 0 -- the real / fake pair pass of the discriminator step: the id head's rows [0, row0) are computed and
        #  dropped; one launch instead of two chain launches from a couple of thousand rows on)
        big = (mgan and R >= (DHEADS_MIN_ROWS if row0 == 0 else DHEADS_PAIR_MIN_ROWS) and W == 192
               and tuple(Wa[0].shape) == (96, 192) and tuple(r[0].weight.shape) == (96, 192) and r[2].weight.shape[0] <= 15)
        if big:
            g = r[2].weight.shape[0]
            Wb, bb = (r[0].weight, r[2].weight), (r[0].bias, r[2].bias)
            ha = _empty(R, 96, like=X) if save else None
            hb = _empty(R, 96, like=X) if save else None
            ya, yb = _empty(R, 1, like=X), _empty(R, g, like=X)
            lib.mggan_dheads_fwd(_p(X), W, R, g, D._out_act(), _p(Wa[0]), _p(ba[0]), _p(Wa[1]), _p(ba[1]), _p(Wb[0]),
                                 _p(bb[0]), _p(Wb[1]), _p(bb[1]), _p(ha), _p(hb), _p(ya), _p(yb), st)
            outs_a, outs_b = [ha, ya], [hb if hb is None or not row0 else hb[row0:], yb if not row0 else yb[row0:]]
            big = True if row0 == 0 else "pair"
        else:
            outs_a = _chain_fwd(X, W, R, spec_a, Wa, ba, save)
            if mgan:
                Wb, bb = (r[0].weight, r[2].weight), (r[0].bias, r[2].bias)
                outs_b = _chain_fwd(_p(X) + 4 * row0 * W, W, R - row0, ((ACT_LEAKY, 0.2), (ACT_NONE, 0.0)), Wb, bb, save)
        if save:
            ctx.cfg = (D, K, row0, b, W, w_sc, mgan, Wa[0].requires_grad, bool(mgan and Wb[0].requires_grad), big)
            ctx.save_for_backward(X, outs_a[0], outs_a[1], outs_b[0], outs_b[1])
        return outs_a[-1], outs_b[-1]

    @staticmethod
    def backward(ctx, dya, dyb):
        D, K, row0, b, W, w_sc, mgan, train_a, train_b, big = ctx.cfg
        R = K * b
        d0 = D.discs[0]
        st = _s()
        if big == "lean":
            # rows [0, b): all 192 columns (the social block and its h columns feed DRowsBodyFn.backward); the other
            # blocks: the pred_enc columns only -- nothing else of dX is read (no gradient for in_enc / scene here)
            ha, ya, hb, mask = ctx.saved_tensors
            r = D.gen_id_reconstructor
            g, act = r[2].weight.shape[0], D._out_act()
            dX = _empty(R, W, like=ya)
            dya = torch.zeros(R, 1, dtype=F32, device=ya.device) if dya is None else dya.reshape(R, 1).contiguous()
            dyb = torch.zeros(R, g, dtype=F32, device=ya.device) if dyb is None else dyb.reshape(R, g).contiguous()
            lib.mggan_dheads_bwd_data(_p(dya), _p(dyb), _p(ya), _p(ha), _p(hb), b, g, act, _p(d0[0].weight),
                                      _p(d0[2].weight), _p(r[0].weight), _p(r[2].weight), _p(dX), W, st)
            lib.mggan_dheads_lean_bwd(_p(dya), _p(dyb), _p(ya), _p(mask), b, R, g, act, _p(d0[0].weight), _p(d0[2].weight),
                                      _p(r[0].weight), _p(r[2].weight), 96, _p(dX), W, st)
            return (dX if ctx.needs_input_grad[0] else None,) + (None,) * 7
        X, ha, ya, hb, yb = ctx.saved_tensors
        dX = _empty(R, W, like=X)
        if big and train_a and train_b and dya is not None and dyb is not None:
            # trainable heads (the discriminator step's pair pass): one launch for the input gradient of both heads; it
            # leaves the gate gradients of the first layers and of head A's output, the operands of the four
            # weight-gradient products (two chain launches otherwise)
            r = D.gen_id_reconstructor
            g = r[2].weight.shape[0]
            dya = dya.reshape(R, 1).contiguous()
            dyb = dyb.reshape(R - row0, g).contiguous()
            dH, dza = _empty(R, 192, like=X), _empty(R, 1, like=X)
            lib.mggan_dheads_bwd_train(_p(dya), _p(dyb), _p(ya), _p(ha), _p(hb), R, row0, g, D._out_act(), _p(d0[0].weight),
                                       _p(d0[2].weight), _p(r[0].weight), _p(r[2].weight), _p(dX), W, _p(dH), _p(dza), st)
            root = root_of(d0[0])
            gp = root.grad_ptr
            with side_stream(dH, dza, dyb, X, ha, hb):
                wgrad(dH, 192, X, W, gp(d0[0].weight), 192, gp(d0[0].bias), R, 192, 96)
                wgrad(dza, 1, ha, 96, gp(d0[2].weight), 96, gp(d0[2].bias), R, 96, 1)
                wgrad(dH[row0:, 96:], 192, X[row0:], W, gp(r[0].weight), 192, gp(r[0].bias), R - row0, 192, 96)
                wgrad(dyb, g, hb, 96, gp(r[2].weight), 96, gp(r[2].bias), R - row0, 96, g)
            if _DEFER["on"]:
                _DEFER["keep"].append(X)
        elif big is True and not train_a and not train_b:  # frozen discriminator (generator step): input gradient only, one launch
            r = D.gen_id_reconstructor
            g = r[2].weight.shape[0]
            dya = torch.zeros(R, 1, dtype=F32, device=X.device) if dya is None else dya.reshape(R, 1).contiguous()
            dyb = torch.zeros(R, g, dtype=F32, device=X.device) if dyb is None else dyb.reshape(R, g).contiguous()
            lib.mggan_dheads_bwd_data(_p(dya), _p(dyb), _p(ya), _p(ha), _p(hb), R, g, D._out_act(), _p(d0[0].weight),
                                      _p(d0[2].weight), _p(r[0].weight), _p(r[2].weight), _p(dX), W, st)
        else:
            if dya is None:
                dX.zero_()
            else:
                dya, ld = _rows2d(dya.reshape(R, -1))
                _chain_bwd(dya, ld, X, W, R, (ha, ya), ((ACT_LEAKY, 0.2), (D._out_act(), 0.0)),
                           (d0[0].weight, d0[2].weight), (d0[0].bias, d0[2].bias), True, train_a, d0[0],
                           dx_into=(_p(dX), W, 0))
            if mgan and dyb is not None:
                r = D.gen_id_reconstructor
                dyb, ld = _rows2d(dyb.reshape(R - row0, -1))
                _chain_bwd(dyb, ld, _p(X) + 4 * row0 * W, W, R - row0, (hb, yb), ((ACT_LEAKY, 0.2), (ACT_NONE, 0.0)),
                           (r[0].weight, r[2].weight), (r[0].bias, r[2].bias), True, train_b, r[0],
                           dx_into=(_p(dX) + 4 * row0 * W, W, 1))
                if _DEFER["on"]:
                    _DEFER["keep"].append(X)
        dsc = None
        if ctx.needs_input_grad[1]:
            dsc = _empty(b, w_sc, like=X)
            lib.mggan_d_rows_reduce(b, K, 0, 0, W - w_sc, w_sc, _p(dX), W, 0, 0, _p(dsc), w_sc, st)
        return (dX if ctx.needs_input_grad[0] else None, dsc) + (None,) * 6


# ---------------------------------------- losses -------------------------------------------
_UNIT_GRADS = set()


def register_unit_grad(t):
    """Tell the loss Functions that this device scalar is exactly 1.0 (skips the scaling launch)."""
    _UNIT_GRADS.add(t.data_ptr())


def _scaled(grad, g):
    """grad *= g (g: 0-dim device tensor from autograd) without a host sync."""
    if g is None:
        return None
    if g.data_ptr() in _UNIT_GRADS:
        return grad
    g = g.reshape(1).to(F32)
    lib.mggan_scale(_p(grad), grad.numel(), _p(g), _s())
    return grad


class BceMeanFn(Function):
    """mean over rows of w_r * BCE(p_r, label)  (abstract_train.py:62-67 'NS'; train.py:92-97 re-weighting)."""

    @staticmethod
    def forward(ctx, p_rows, label, row_gen, inv_count, out, norm=None, kind=0, sign=1.0):
        # kind 0: BCE ('NS'; 'MM' on the generator side with sign = -1), kind 1: squared error ('LS')
        p_rows = p_rows.contiguous()
        rows = p_rows.numel()
        loss_rows = _empty(rows, like=p_rows)
        dp = _empty(rows, like=p_rows)
        if isinstance(label, tuple):  # (u (1,) device tensor, lo, hi): label = lo + (hi-lo)*u drawn on the GPU
            lab, lab_u, lo, hi = 0.0, _p(label[0]), float(label[1]), float(label[2])
        else:
            lab, lab_u, lo, hi = float(label), 0, 0.0, 0.0
        lib.mggan_bce_rows(rows, kind, _p(p_rows), lab, lab_u, lo, hi, float(sign) / float(norm or max(rows, 1)), _p(row_gen),
                           _p(inv_count),
                           _p(loss_rows), _p(dp), _s())
        lib.mggan_sum(_p(loss_rows), rows, 1.0, _p(out), 0, _s())
        ctx.dp, ctx.shape = dp, p_rows.shape
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx.dp, g).view(ctx.shape), None, None, None, None, None, None, None


class _GanLossArgs(ctypes.Structure):  # mirrors csrc/loss_opt.hip:GanLossArgs
    _fields_ = [("p", ctypes.c_void_p), ("label_u", ctypes.c_void_p * 2), ("row_gen", ctypes.c_void_p),
                ("seg", ctypes.c_void_p), ("inv_count", ctypes.c_void_p), ("logits", ctypes.c_void_p),
                ("target", ctypes.c_void_p), ("dp", ctypes.c_void_p), ("dlogits", ctypes.c_void_p),
                ("out", ctypes.c_void_p * 3), ("total", ctypes.c_void_p), ("partial", ctypes.c_void_p),
                ("ticket", ctypes.c_void_p), ("dims", ctypes.c_void_p),
                ("label", ctypes.c_float * 2), ("lo", ctypes.c_float * 2), ("hi", ctypes.c_float * 2),
                ("scale", ctypes.c_float * 3), ("sign_a", ctypes.c_float), ("grad_c", ctypes.c_float),
                ("nA", ctypes.c_int), ("nB", ctypes.c_int), ("nC", ctypes.c_int), ("g", ctypes.c_int),
                ("ld", ctypes.c_int), ("kind", ctypes.c_int), ("weighted_c", ctypes.c_int), ("bmod", ctypes.c_int)]


_LOSS_SCRATCH = {}


def _loss_scratch(kind="gan", doubles=768):
    """Per-device scratch of a loss kernel: (partial sums, one ticket word the kernel always leaves at zero).
    The loss launches of an iteration are ordered on the main stream, so one scratch per kind is enough - and keying it
    by stream would allocate (and zero-fill, two ATen launches) again inside a graph capture, whose stream is new."""
    key = (kind, torch.cuda.current_device())
    hit = _LOSS_SCRATCH.get(key)
    if hit is None:
        hit = _LOSS_SCRATCH[key] = (torch.zeros(doubles, dtype=torch.float64, device="cuda"),
                                    torch.zeros(1, dtype=torch.int32, device="cuda"))
    return hit


class GanLossesFn(Function):
    """Every adversarial loss term of one optimizer step as ONE launch and ONE autograd node
    (abstract_train.py:62-75, train.py:92-111,184): term A over rows [0,nA) of p, term B over the next nB rows,
    term C = cross entropy over the classifier logits.  -> A + B + grad_c * C; the terms go to `outs`."""

    @staticmethod
    def forward(ctx, p, logits, o):
        # o: dict(nA, nB, labels=(lab_a, lab_b), norms=(nA, nB, nC) global row counts, kind, sign_a, grad_c,
        #         row_gen, seg, inv_count, target, weighted_c, outs=(tA, tB, tC) one-element tensors or None)
        p = p.contiguous()
        a = _GanLossArgs()
        a.nA, a.nB = int(o["nA"]), int(o.get("nB", 0))
        assert p.numel() == a.nA + a.nB
        dp = _empty(p.numel(), like=p)
        a.p, a.dp = _p(p), _p(dp)
        for q, lab in enumerate(o["labels"]):
            if isinstance(lab, tuple):  # (u (1,) device tensor, lo, hi): label = lo + (hi-lo)*u drawn on the GPU
                a.label_u[q], a.lo[q], a.hi[q] = _p(lab[0]), float(lab[1]), float(lab[2])
            elif lab is not None:
                a.label[q] = float(lab)
        norms = o["norms"]
        a.scale[0] = 1.0 / float(norms[0] or max(a.nA, 1))
        a.scale[1] = 1.0 / float(norms[1] or max(a.nB, 1))
        a.kind, a.sign_a, a.grad_c = int(o.get("kind", 0)), float(o.get("sign_a", 1.0)), float(o.get("grad_c", 1.0))
        row_gen = o.get("row_gen")
        a.row_gen, a.seg, a.inv_count = _p(row_gen), _p(o.get("seg")), _p(o.get("inv_count"))
        dl = None
        if logits is not None:
            logits = logits.contiguous()
            a.nC, a.g = logits.shape
            a.ld = a.g
            dl = _empty(a.nC, a.g, like=p)
            a.logits, a.dlogits, a.target = _p(logits), _p(dl), _p(o["target"])
            a.scale[2] = 1.0 / float(norms[2] or max(a.nC, 1))
            a.weighted_c = int(bool(o.get("weighted_c", False)))
        elif row_gen is not None:
            a.g = int(o["g"])
        for q, t in enumerate(o["outs"]):
            a.out[q] = _p(t)
        total = _empty(1, like=p)
        a.total = _p(total)
        partial, ticket = _loss_scratch()
        a.partial, a.ticket = _p(partial), _p(ticket)
        if _PAD["dims"] is not None:  # padded batch: row r of every term belongs to pedestrian r % b_pad
            assert a.seg is None or not a.seg, "padded batches count generators without the phantom rows (inv_count)"
            a.dims, a.bmod = _pad_ptr(), _PAD["b"]
            assert a.bmod > 0 and a.nA % a.bmod == 0 and a.nB % a.bmod == 0 and a.nC % a.bmod == 0
        lib.mggan_gan_losses(ctypes.addressof(a), _s())
        ctx.dp, ctx.dl, ctx.shape = dp, dl, p.shape
        return total.view(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx.dp, g).view(ctx.shape), (_scaled(ctx.dl, g) if ctx.dl is not None else None), None


class CeMeanFn(Function):
    """mean over rows of w_r * cross_entropy(logits_r, target_r)  (train.py:105-111,184)."""

    @staticmethod
    def forward(ctx, logits, target, inv_count, out, norm=None):
        logits = logits.contiguous()
        rows, g = logits.shape
        loss_rows = _empty(rows, like=logits)
        dl = _empty(rows, g, like=logits)
        lib.mggan_ce_rows(rows, g, _p(logits), g, _p(target), _p(inv_count), 1.0 / float(norm or max(rows, 1)),
                          _p(loss_rows), _p(dl), g, _s())
        lib.mggan_sum(_p(loss_rows), rows, 1.0, _p(out), 0, _s())
        ctx.dl = dl
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx.dl, g), None, None, None, None


class L2MinSceneFn(Function):
    """sum_scenes min_k sum_{ped,t} |abs - gt| / b   (train.py:58-75)."""

    @staticmethod
    def forward(ctx, gen_abs, gt, tb, b_norm, out):
        gen_abs, gt = gen_abs.contiguous(), gt.contiguous()
        T, K, b, _ = gen_abs.shape
        scene_loss = _empty(tb.S, like=gen_abs)
        scene_arg = torch.empty(tb.S, dtype=torch.int32, device=gen_abs.device)
        gabs = _empty(T, K, b, 2, like=gen_abs)
        lib.mggan_l2_min_scene(tb.S, T, K, b, _p(tb.scenes), _p(tb.ped_scene), _p(gen_abs), _p(gt), 1.0 / b_norm,
                               _p(scene_loss), _p(scene_arg), _p(gabs), _pad_ptr(), _s())
        lib.mggan_sum(_p(scene_loss), tb.S, 1.0 / b_norm, _p(out), 0, _s())
        ctx.gabs = gabs
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx.gabs, g), None, None, None, None


class PmMlFn(Function):
    """PM-network 'ml' objective (train.py:626-639): target = softmax_g(mean_E sum log N(err;0,sigma))."""

    @staticmethod
    def forward(ctx, logits, gen_abs, gt, sigma, out, probs_out, norm=None):
        logits, gen_abs, gt = logits.contiguous(), gen_abs.contiguous(), gt.contiguous()
        T, E, g, b, _ = gen_abs.shape
        loss_rows = _empty(b, like=logits)
        dl = _empty(b, g, like=logits)
        probs = _empty(b, g, like=logits)
        n = float(norm or b)
        scratch = _loss_scratch("pm", 17 * 64)
        lib.mggan_pm_ml_loss_mean(b, T, E, g, _p(gen_abs), _p(gt), _p(logits), float(sigma), 1.0 / n, _p(loss_rows),
                                  _p(dl), _p(probs), _p(scratch[0]), _p(scratch[1]), _p(out), _p(probs_out),
                                  float(b) / n, _pad_ptr(), _s())
        ctx.dl = dl
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx.dl, g), None, None, None, None, None, None


class PmMganFn(Function):
    """PM-network 'mgan' objective as the reference computes it (train.py:606-614): the softmax that should turn the
    discriminator's generator-id logits into targets runs over the singleton sample axis, every target is 1 and the
    product broadcasts over the batch -> -(1/g) sum_{r,j} log p_rj - 0.9^epoch * mean_r H(p_r)."""

    @staticmethod
    def forward(ctx, logits, reg, out, probs_out, norm=None):
        """reg: a Python number, or a 0-dim device tensor (read by the kernel at run time: graph replays follow the epoch)"""
        logits = logits.contiguous()
        b, g = logits.shape
        n = float(norm or b)
        loss_rows, dl, probs = _empty(b, like=logits), _empty(b, g, like=logits), _empty(b, g, like=logits)
        on_dev = torch.is_tensor(reg)
        lib.mggan_pm_mgan_loss(b, g, _p(logits), n / g, 0.0 if on_dev else float(reg), _p(reg) if on_dev else 0, 1.0 / n,
                               _p(loss_rows), _p(dl), _p(probs), _s())
        lib.mggan_sum(_p(loss_rows), b, 1.0, _p(out), 0, _s())
        if probs_out is not None:
            lib.mggan_colmean(_p(probs), b, g, float(b) / n, _p(probs_out), _s())
        ctx.dl = dl
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx.dl, g), None, None, None, None
