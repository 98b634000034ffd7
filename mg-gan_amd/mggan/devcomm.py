"""Peer-mapped all-reduce between the GPUs of one node (csrc/comm.hip): the collectives of the scene-sharded iteration as
plain HIP kernels, so that the iteration stays ONE captured graph with its branch streams (torch.distributed / RCCL
calls would cut the capture at each of the ~18 exchange points of an iteration).

Every rank allocates one uncached arena per CHANNEL, exports it through hipIpc and maps its peers' arenas; the handles
travel once, at start-up, through torch.distributed (any backend).  A channel is a stream: collectives issued on the
same stream use the same arena and are ordered by the stream; two branch streams never share one.  All ranks run the
same program, so they create channels -- and issue collectives on them -- in the same order."""
import ctypes
import os
import socket

import torch
import torch.distributed as dist

from mggan.hip.lib import lib

_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2}


CHANNELS = 1 + 4 + 1


def current_channel(device):
    """The channel (stream ROLE) of torch's current stream: 0 = the main chain, 1 + i = branch stream i of
    mggan.hip.functions, CHANNELS - 1 = the Gram side stream.  All ranks run the same program: they agree on the roles."""
    from mggan.hip import functions as HF

    cur = torch.cuda.current_stream(device)
    if HF._GRAM.get("stream") is not None and cur == HF._GRAM["stream"]:
        return CHANNELS - 1
    for which, st in HF._BR["streams"].items():
        if cur == st:
            if not 0 <= int(which) < CHANNELS - 2:
                raise RuntimeError("device all-reduce: branch stream {} has no channel".format(which))
            return 1 + int(which)
    return 0


class DeviceComm:
    MAX_ELEMS = 1 << 16  # 8-byte elements per slot: 512 KB (the flat gradient buffers are 211-360 KB at the default widths)
    # A channel is a logical ROLE, not a raw stream: 0 = the main chain (the caller's stream and every stream that is
    # ordered with it by fork/join -- the capture stream, the autograd-backward stream), 1 + i = branch stream i of
    # mggan.hip.functions.  A process that trains eagerly and then captures, or captures several times, keeps using the
    # same arenas; all ranks run the same program, so they agree on the roles.
    # ... and one for the Gram side stream (its launch and exchange run beside the first scene-CNN pass of an iteration).
    CHANNELS = CHANNELS

    def __init__(self, group, device, max_elems=None):
        """Collective: every rank of `group` calls it.  Raises on EVERY rank if any rank fails (the phases end with an
        exchange of verdicts, so the ranks never wait for a peer that has given up).  max_elems: slot capacity in 8-byte
        elements (sized by the caller from the largest vector it will reduce; the same on every rank)."""
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._local, self._opened, self._arenas, self._chan_dev = [], [], [], {}
        if max_elems is not None:
            self.MAX_ELEMS = max(int(max_elems), 1 << 10)
        lib.mggan_comm_set_timeout(float(os.environ.get("MGGAN_COMM_TIMEOUT_S", "30")))
        hp = ctypes.POINTER(ctypes.c_uint)()
        lib.mggan_comm_host_error(ctypes.byref(hp))
        self._host_error = hp  # host-mapped word: set by any collective of this process whose wait timed out

        def agree(payload, what):
            everyone = [None] * self.world
            dist.all_gather_object(everyone, payload, group=group)
            bad = [j for j, p in enumerate(everyone) if p is None]
            if bad:
                self.close()
                raise RuntimeError("{} failed on rank(s) {}".format(what, bad))
            return everyone

        # phase 1: allocate and export this rank's arenas
        mine = None
        try:
            if self.world > 8:
                raise RuntimeError("at most 8 ranks (one node)")
            nbytes = lib.mggan_comm_arena_bytes(self.MAX_ELEMS)
            handles = []
            with torch.cuda.device(self.device):
                for _ in range(self.CHANNELS):
                    p = ctypes.c_void_p()
                    lib.mggan_comm_alloc(nbytes, ctypes.byref(p))
                    self._local.append(p)
                    h = ctypes.create_string_buffer(64)
                    lib.mggan_comm_ipc_handle(p, h)
                    handles.append(h.raw)
                torch.cuda.synchronize(self.device)
            mine = (socket.gethostname(), handles)
        except Exception as exc:  # noqa: BLE001
            print("[mggan] device all-reduce, rank {}: {}: {}".format(self.rank, type(exc).__name__, exc))
        everyone = agree(mine, "arena allocation / export")
        if len({h for h, _ in everyone}) != 1:
            self.close()
            raise RuntimeError("the ranks are not on one node")
        # phase 2: map the peers' arenas
        ok = None
        try:
            with torch.cuda.device(self.device):
                for ch in range(self.CHANNELS):
                    arr = (ctypes.c_void_p * self.world)()
                    for j, (_, hs) in enumerate(everyone):
                        if j == self.rank:
                            arr[j] = self._local[ch]
                        else:
                            q = ctypes.c_void_p()
                            lib.mggan_comm_ipc_open(hs[ch], ctypes.byref(q))
                            self._opened.append(q)
                            arr[j] = q
                    self._arenas.append(arr)
            ok = True
        except Exception as exc:  # noqa: BLE001
            print("[mggan] device all-reduce, rank {}: {}: {}".format(self.rank, type(exc).__name__, exc))
        agree(ok, "peer mapping")  # also the barrier: every arena is mapped everywhere before the first collective
        # the host-mapped error word is one per process: a timed-out collective of an EARLIER communicator (closed since)
        # must not fail this one's first check -- its own arenas are fresh.  Cleared here, where every rank has agreed that
        # the new arenas are mapped and no collective of this communicator has run yet.
        self._host_error[0] = 0
        # phase 3: known-answer exchanges
        agree(True if self.self_test() else None, "self test (wrong sums or a timed-out wait)")

    def self_test(self):
        """One f32, one f64 and one i32 exchange on the first channel with known answers, and no timed-out wait."""
        w = self.world
        want = float(w * (w + 1) // 2)
        ok = True
        with torch.cuda.device(self.device):
            for dt, n in ((torch.float32, 3000), (torch.float64, 33), (torch.int32, 9)):
                x = torch.full((n,), self.rank + 1, dtype=dt, device=self.device)
                self.all_reduce_(x)
                torch.cuda.synchronize(self.device)
                ok = ok and bool((x == want).all())
        try:
            self.check()
        except RuntimeError:
            ok = False
        return ok

    def _channel(self):
        return current_channel(self.device)

    def channel_args(self):
        """(arenas, rank, world, max_elems) of the current stream's channel, for kernels that embed the small exchange
        (mggan_bn_sync_finalize)."""
        return self._arenas[self._channel()], self.rank, self.world, self.MAX_ELEMS

    def channel_dev(self):
        """Device copy of the current stream's channel arguments (include/mggan_hip.h: the `comm` argument of the scene-CNN
        entries that fold their BatchNorm exchange into their last workgroup); created at the channel's first use."""
        ch = self._channel()
        p = self._chan_dev.get(ch)
        if p is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("device all-reduce: channel {} meets its first in-launch exchange inside a capture (run "
                                   "an eager iteration first)".format(ch))
            q = ctypes.c_void_p()
            lib.mggan_comm_channel_create(self._arenas[ch], self.rank, self.world, self.MAX_ELEMS, ctypes.byref(q))
            p = self._chan_dev[ch] = q
        return p

    def chunks_fit(self, n_f32):
        """Can the optimizer's launch run the exchange of n_f32 floats + a tail chunk itself (one flag per 1,024-element
        chunk, csrc/loss_opt.hip)?"""
        return (n_f32 + 1023) // 1024 + 1 <= (self.MAX_ELEMS * 2 + 1023) // 1024

    def supports(self, t, tail=None):
        """Does `t` (and an f64 tail behind it, at the next 256-byte boundary of the slot) fit an arena slot?"""
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES):
            return False
        nbytes = t.numel() * t.element_size()
        if tail is not None:
            if tail.dtype != torch.float64 or not tail.is_contiguous():
                return False
            nbytes = (nbytes + 255) // 256 * 256 + tail.numel() * 8
        return nbytes <= self.MAX_ELEMS * 8

    def all_reduce_(self, t, tail=None):
        """Sum over the ranks, in place, on the current stream (capturable); `tail`: an f64 vector summed in the same
        exchange (csrc/comm.hip: mggan_comm_allreduce2)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        if tail is None:
            lib.mggan_comm_allreduce(self._arenas[self._channel()], self.rank, self.world, self.MAX_ELEMS, t.data_ptr(),
                                     t.numel(), _DTYPES[t.dtype], st)
        else:
            lib.mggan_comm_allreduce2(self._arenas[self._channel()], self.rank, self.world, self.MAX_ELEMS, t.data_ptr(),
                                      t.numel(), _DTYPES[t.dtype], tail.data_ptr(), tail.numel(), st)
        return t

    def failed(self):
        """True once any collective of this process has timed out.  A plain host read (no device sync, no HIP call): cheap
        enough for every iteration of the training loop; the device may still be running, so a False is final only after
        a synchronisation."""
        return bool(self._host_error[0])

    def check(self, sync=True):
        """Raise if a wait timed out on any channel (a peer was lost or the ranks issued different collectives).
        sync=False: only look at the host-mapped word (what has been reported so far)."""
        if not sync:
            if self.failed():
                raise RuntimeError("device all-reduce: a wait timed out on rank {} (a peer was lost or the ranks issued "
                                   "different collectives); the reduced buffers hold NaN".format(self.rank))
            return
        bad = []
        for ch, p in enumerate(self._local):
            e = ctypes.c_uint()
            lib.mggan_comm_error(p, ctypes.byref(e))
            if e.value:
                bad.append(ch)
        if bad:
            raise RuntimeError("device all-reduce: timed-out wait on channel(s) {} of rank {}".format(bad, self.rank))

    def close(self):
        for q in getattr(self, "_chan_dev", {}).values():
            lib.mggan_comm_channel_free(q)
        self._chan_dev = {}
        for q in getattr(self, "_opened", []):
            lib.mggan_comm_ipc_close(q)
        for p in getattr(self, "_local", []):
            lib.mggan_comm_free(p)
        self._opened, self._local = [], []


def create(group, device, max_elems=None):
    """-> DeviceComm or None (disabled by MGGAN_DEVICE_COMM=0, or the mapping failed on some rank: every rank then
    falls back to torch.distributed together)."""
    if os.environ.get("MGGAN_DEVICE_COMM", "1") == "0":
        return None
    try:
        return DeviceComm(group, device, max_elems)  # raises on every rank together
    except Exception as exc:  # noqa: BLE001
        print("[mggan] device all-reduce unavailable ({}: {}); torch.distributed collectives between graph segments "
              "instead".format(type(exc).__name__, exc))
        return None


def _device_identity(device):
    """Something that tells two physical GPUs of one host apart whatever the ranks' device numbering is."""
    props = torch.cuda.get_device_properties(device)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(props, attr, None)
        if v is not None:
            return "{}:{}".format(attr, v)
    return "index:{}:{}".format(os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")),
                                torch.device(device).index)


class RcclComm:
    """RCCL all-reduce INSIDE the iteration graph (csrc/rccl.hip): ncclAllReduce bound from librccl.so by this package's
    own library and issued on torch's current stream -- an ordinary capturable launch, unlike torch.distributed's
    collectives, which cut the capture into graph segments.  The transport north_star names; the peer-mapped kernels
    (DeviceComm) stay the default on one node because every message of an iteration is latency bound (<= 360 KB).

    One communicator per CHANNEL (stream role, as DeviceComm's arenas): collectives of one communicator are ordered by
    their stream, two streams never share one -- no cross-stream ordering inside a capture.  Rank 0 draws the ids once,
    every rank initialises a channel's communicator at its first collective there (all ranks run the same program; the
    eager warm-up iterations in front of a capture touch every channel the iteration uses)."""

    def __init__(self, group, device):
        """Collective.  Raises on every rank together when RCCL is unusable (library missing, two ranks on one device)."""
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._comms = {}
        ids = None
        try:
            if not lib.mggan_rccl_available():
                raise RuntimeError("librccl.so does not resolve in this process")
            if self.rank == 0:
                ids = []
                for _ in range(CHANNELS):
                    b = ctypes.create_string_buffer(128)
                    lib.mggan_rccl_unique_id(b)
                    ids.append(b.raw)
            mine = (socket.gethostname(), _device_identity(self.device), ids)
        except Exception as exc:  # noqa: BLE001
            print("[mggan] in-graph RCCL, rank {}: {}: {}".format(self.rank, type(exc).__name__, exc))
            mine = None
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        if any(e is None for e in everyone):
            raise RuntimeError("RCCL unavailable on rank(s) {}".format([j for j, e in enumerate(everyone) if e is None]))
        if len({(h, d) for h, d, _ in everyone}) != self.world:
            raise RuntimeError("several ranks share one device (RCCL wants one GPU per rank)")
        self._ids = everyone[0][2]
        ok = None
        try:
            ok = True if self.self_test() else None
        except Exception as exc:  # noqa: BLE001
            print("[mggan] in-graph RCCL, rank {}: {}: {}".format(self.rank, type(exc).__name__, exc))
        dist.all_gather_object(everyone, ok, group=group)
        if any(e is None for e in everyone):
            self.close()
            raise RuntimeError("RCCL self test failed on rank(s) {}".format([j for j, e in enumerate(everyone) if e is None]))

    def _comm(self, ch):
        c = self._comms.get(ch)
        if c is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("in-graph RCCL: channel {} meets its first collective inside a capture (run an eager "
                                   "iteration first)".format(ch))
            c = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                lib.mggan_rccl_comm_init(self._ids[ch], self.rank, self.world, ctypes.byref(c))
            self._comms[ch] = c
        return c

    def self_test(self):
        w = self.world
        want = float(w * (w + 1) // 2)
        ok = True
        with torch.cuda.device(self.device):
            for dt, n in ((torch.float32, 3000), (torch.float64, 33), (torch.int32, 9)):
                x = torch.full((n,), self.rank + 1, dtype=dt, device=self.device)
                tail = torch.full((7,), self.rank + 1, dtype=torch.float64, device=self.device) if dt == torch.float32 else None
                self.all_reduce_(x, tail)
                torch.cuda.synchronize(self.device)
                ok = ok and bool((x == want).all()) and (tail is None or bool((tail == want).all()))
        return ok

    def supports(self, t, tail=None):
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES):
            return False
        return tail is None or (tail.is_cuda and tail.dtype == torch.float64 and tail.is_contiguous())

    def all_reduce_(self, t, tail=None):
        """Sum over the ranks, in place, on the current stream (capturable); `tail`: an f64 vector summed in the same RCCL
        group (one launch)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        lib.mggan_rccl_allreduce(self._comm(current_channel(self.device)), t.data_ptr(), t.numel(), _DTYPES[t.dtype],
                                 tail.data_ptr() if tail is not None else 0, tail.numel() if tail is not None else 0, st)
        return t

    def failed(self):
        for c in self._comms.values():
            e = ctypes.c_int()
            lib.mggan_rccl_async_error(c, ctypes.byref(e))
            if e.value:
                return True
        return False

    def check(self, sync=True):
        if sync:
            torch.cuda.synchronize(self.device)
        if self.failed():
            raise RuntimeError("in-graph RCCL: asynchronous error on rank {}".format(self.rank))

    def close(self):
        comms, self._comms = getattr(self, "_comms", {}), {}
        for c in comms.values():
            try:
                lib.mggan_rccl_comm_destroy(c)
            except Exception:  # noqa: BLE001
                pass


def create_rccl(group, device):
    """-> RcclComm or None (MGGAN_RCCL_GRAPH=0, librccl missing, ranks sharing a device: every rank falls back to
    torch.distributed between graph segments together)."""
    if os.environ.get("MGGAN_RCCL_GRAPH", "1") == "0":
        return None
    try:
        return RcclComm(group, device)
    except Exception as exc:  # noqa: BLE001
        print("[mggan] in-graph RCCL unavailable ({}: {}); torch.distributed collectives between graph segments "
              "instead".format(type(exc).__name__, exc))
        return None
