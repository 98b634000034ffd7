#!/usr/bin/env python
"""Times mggan_grad_reduce_multi alone on the partial-buffer shapes of a real backward pass (the [M, Naug, splits, groups,
pitch] lists printed by MGGAN_REDUCE_DUMP=1): python tools/bench_reduce.py '<json list>' ..."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-gan_amd"))
import torch  # noqa: E402

from mggan.hip import functions as HF  # noqa: E402
from mggan.hip import lib  # noqa: E402

dev = torch.device("cuda")
for spec in sys.argv[1:]:
    shapes = json.loads(spec)
    keep, descs, nbytes = [], [], 0
    for M, Naug, splits, groups, pitch in shapes:
        P = torch.randn(groups * splits * pitch, device=dev)
        dW = torch.zeros(groups * M * Naug, device=dev)
        keep += [P, dW]
        nbytes += 4 * M * Naug * splits * groups
        descs.append(HF._ReduceDesc(P.data_ptr(), dW.data_ptr(), None, M * Naug, 0, M, Naug, 0, Naug, splits, groups, pitch, 0))
    arr = (HF._ReduceDesc * len(descs))(*descs)
    s = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(5):
        lib.mggan_grad_reduce_multi(ctypes.addressof(arr), len(descs), s)
    ev[0].record()
    for _ in range(50):
        lib.mggan_grad_reduce_multi(ctypes.addressof(arr), len(descs), s)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    print("{} buffers, {:.1f} MB: {:.1f} us per launch = {:.2f} TB/s".format(len(shapes), nbytes / 1e6, us, nbytes / us / 1e6))
