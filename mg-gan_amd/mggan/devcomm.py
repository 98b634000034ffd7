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


class DeviceComm:
    MAX_ELEMS = 1 << 16  # 8-byte elements per slot: 512 KB (the flat gradient buffers are 211-360 KB)
    CHANNELS = 6

    def __init__(self, group, device):
        """Collective: every rank of `group` calls it.  Raises on EVERY rank if any rank fails (the phases end with an
        exchange of verdicts, so the ranks never wait for a peer that has given up)."""
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._local, self._opened, self._arenas, self._channel_of = [], [], [], {}

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
        sid = torch.cuda.current_stream(self.device).cuda_stream
        ch = self._channel_of.get(sid)
        if ch is None:
            ch = len(self._channel_of)
            if ch >= self.CHANNELS:
                raise RuntimeError("device all-reduce: more than {} streams issue collectives".format(self.CHANNELS))
            self._channel_of[sid] = ch
        return ch

    def channel_args(self):
        """(arenas, rank, world, max_elems) of the current stream's channel, for kernels that embed the small exchange
        (mggan_bn_sync_finalize)."""
        return self._arenas[self._channel()], self.rank, self.world, self.MAX_ELEMS

    def supports(self, t):
        cap = self.MAX_ELEMS * (1 if t.dtype == torch.float64 else 2)
        return t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES and t.numel() <= cap

    def all_reduce_(self, t):
        """Sum over the ranks, in place, on the current stream (capturable)."""
        lib.mggan_comm_allreduce(self._arenas[self._channel()], self.rank, self.world, self.MAX_ELEMS, t.data_ptr(),
                                 t.numel(), _DTYPES[t.dtype], torch.cuda.current_stream(self.device).cuda_stream)
        return t

    def check(self):
        """Raise if a wait timed out on any channel (a peer was lost or the ranks issued different collectives)."""
        bad = []
        for ch, p in enumerate(self._local):
            e = ctypes.c_uint()
            lib.mggan_comm_error(p, ctypes.byref(e))
            if e.value:
                bad.append(ch)
        if bad:
            raise RuntimeError("device all-reduce: timed-out wait on channel(s) {} of rank {}".format(bad, self.rank))

    def close(self):
        for q in getattr(self, "_opened", []):
            lib.mggan_comm_ipc_close(q)
        for p in getattr(self, "_local", []):
            lib.mggan_comm_free(p)
        self._opened, self._local = [], []


def create(group, device):
    """-> DeviceComm or None (disabled by MGGAN_DEVICE_COMM=0, or the mapping failed on some rank: every rank then
    falls back to torch.distributed together)."""
    if os.environ.get("MGGAN_DEVICE_COMM", "1") == "0":
        return None
    try:
        return DeviceComm(group, device)  # raises on every rank together
    except Exception as exc:  # noqa: BLE001
        print("[mggan] device all-reduce unavailable ({}: {}); torch.distributed collectives between graph segments "
              "instead".format(type(exc).__name__, exc))
        return None
