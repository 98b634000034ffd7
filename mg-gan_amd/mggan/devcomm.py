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
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise RuntimeError("device all-reduce: at most 8 ranks (one node)")
        nbytes = lib.mggan_comm_arena_bytes(self.MAX_ELEMS)
        self._local, handles = [], []
        with torch.cuda.device(self.device):
            for _ in range(self.CHANNELS):
                p = ctypes.c_void_p()
                lib.mggan_comm_alloc(nbytes, ctypes.byref(p))
                h = ctypes.create_string_buffer(64)
                lib.mggan_comm_ipc_handle(p, h)
                self._local.append(p)
                handles.append(h.raw)
            torch.cuda.synchronize(self.device)
            everyone = [None] * self.world
            dist.all_gather_object(everyone, (socket.gethostname(), os.getpid(), handles), group=group)
            if len({h for h, _, _ in everyone}) != 1:
                raise RuntimeError("device all-reduce: the ranks are not on one node")
            self._arenas, self._opened = [], []
            for ch in range(self.CHANNELS):
                arr = (ctypes.c_void_p * self.world)()
                for j, (_, pid, hs) in enumerate(everyone):
                    if j == self.rank:
                        arr[j] = self._local[ch]
                    else:
                        q = ctypes.c_void_p()
                        lib.mggan_comm_ipc_open(hs[ch], ctypes.byref(q))
                        self._opened.append(q)
                        arr[j] = q
                self._arenas.append(arr)
        dist.barrier(group=group)  # every arena is mapped everywhere before the first collective
        self._channel_of = {}

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
    comm, ok = None, 1
    try:
        comm = DeviceComm(group, device)
    except Exception as exc:  # noqa: BLE001 -- the verdict is agreed on below
        print("[mggan] device all-reduce unavailable on rank {}: {}: {}".format(dist.get_rank(group), type(exc).__name__, exc))
        ok = 0
    verdict = [None] * dist.get_world_size(group)
    dist.all_gather_object(verdict, ok, group=group)
    if not all(verdict):
        if comm is not None:
            comm.close()
        return None
    return comm
