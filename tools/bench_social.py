#!/usr/bin/env python
"""Times mggan_social_rows_fwd / _bwd alone (the launches of one context pass): python tools/bench_social.py [S n H]...
(default: the headline shape, 64 scenes x 20 pedestrians, both widths).  MGGAN_SOC_SPLITS forces the row splits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-gan_amd"))
import torch  # noqa: E402

from mggan.hip import functions as HF  # noqa: E402
from mggan.hip import lib  # noqa: E402

dev = torch.device("cuda")
PROF = bool(os.environ.get("BENCH_SOC_PROFILE"))  # with a -DSR_PROFILE build of the library: segment cycle counts
_p = HF._p


def run(S, n, H, reps=200):
    torch.manual_seed(0)
    b, F = S * n, H
    scenes = torch.tensor([[i * n, (i + 1) * n] for i in range(S)], dtype=torch.int32, device=dev)
    xy, dxy = torch.randn(b, 2, device=dev) * 5, torch.randn(b, 2, device=dev)
    W1, b1, W2, b2 = (torch.randn(*s, device=dev) * 0.2 for s in ((32, 3), (32,), (64, 32), (64,)))
    W3, b3, Wat, bat = (torch.randn(*s, device=dev) * 0.2 for s in ((F, 64), (F,), (F, H), (F,)))
    h, dS = torch.randn(b, H, device=dev), torch.randn(b, H, device=dev)
    Sout, dvc, Wh, dWh, dh = (torch.empty(b, c, device=dev) for c in (H, 68, F, F, H))
    grid, pf = lib.mggan_social_rows_grid(S, n), lib.mggan_social_rows_partial_floats()
    rs = lib.mggan_social_rows_splits(S, n)
    part = torch.empty(grid * pf, device=dev) if not os.environ.get("BENCH_SOC_NOTRAIN") else None
    scr = torch.empty(max(rs, 1) * b * (65 + H), device=dev)
    tick = torch.zeros(S, dtype=torch.int32, device=dev)
    def fwd():
        st = torch.cuda.current_stream().cuda_stream
        lib.mggan_social_rows_fwd(S, _p(scenes), H, F, n, _p(xy), _p(dxy), 0, _p(W1), _p(b1), _p(W2), _p(b2), _p(W3), _p(b3),
                                  _p(Wat), _p(bat), _p(h), H, _p(Sout), H, st)

    def bwd():
        st = torch.cuda.current_stream().cuda_stream
        lib.mggan_social_rows_bwd(S, _p(scenes), H, F, n, _p(xy), _p(dxy), 0, _p(W1), _p(b1), _p(W2), _p(b2), _p(W3), _p(b3),
                                  _p(Wat), _p(bat), _p(h), H, _p(dS), H, _p(dvc), 68, b, _p(Wh), _p(dWh), _p(dh), H, 0,
                                  _p(part), _p(scr) if rs > 1 or PROF else None, _p(tick) if rs > 1 else None, st)

    if PROF:
        names = ["issue weights", "weights staged", "h dS staged", "Wh DA", "v c", "row head", "pass 1", "softmax", "sums+dz2",
                 "dz1 MFMA", "tiles+wgrad", "loop tail", "fold+AtdS", "dense adj", "partials"]
        bwd()
        torch.cuda.synchronize()
        c = scr[:16].cpu().tolist()
        print("  cycles of workgroup 0 / wave 0 (total {:.0f}): ".format(sum(c)) + ", ".join("{} {:.0f}".format(k, v) for k, v in zip(names, c)))
    out = []
    for fn in (fwd, bwd):
        for _ in range(10):
            fn()
        # a graph of `reps` launches: the launch gaps of the eager loop are not what is measured
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        g.replay()
        ev[1].record()
        torch.cuda.synchronize()
        out.append(ev[0].elapsed_time(ev[1]) / reps * 1e3)
    print("S {:4d} n {:3d} H {:2d} splits {} grid {:3d}: fwd {:6.1f} us   bwd {:6.1f} us   (checksum {:.6e} {:.6e})".format(
        S, n, H, rs, grid, out[0], out[1], float(Sout.double().sum()), float(dh.double().sum() + (part.double().sum() if part is not None else 0))))


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(64, 20, 32), (64, 20, 64)]
    for S, n, H in shapes:
        run(S, n, H)
