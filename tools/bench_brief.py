#!/usr/bin/env python
"""Print the per-kernel stand-alone times of bench_detail.json (all kernels of every measured workload), optionally beside a
second file:  python tools/bench_brief.py [a.json] [b.json] [filter]"""
import json
import sys


def load(path):
    d = json.load(open(path))
    out = {}
    for c in d["configs"]:
        rows = {r["kernel"]: (r["ms_per_step"], r["calls"]) for r in c.get("breakdown", [])}
        out[c["config"]] = (c["ms_per_step"], rows, {r["kernel"]: r for r in c.get("roofline_top_kernels", [])})
    return out


def main(a, b=None, flt=""):
    A, Bm = load(a), (load(b) if b else None)
    for cfg, (ms, rows, top) in A.items():
        print("{}: {:.4f} ms per iteration{}".format(cfg, ms, "  (other: {:.4f})".format(Bm[cfg][0]) if Bm and cfg in Bm else ""))
        for k, (t, calls) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
            if flt and flt not in k:
                continue
            extra = ""
            if k in top:
                extra = "  in-graph {:.1f} us/launch".format(top[k]["avg_launch_ms"] * 1e3)
            other = ""
            if Bm and cfg in Bm and k in Bm[cfg][1]:
                other = "   other {:8.1f}".format(Bm[cfg][1][k][0] / max(Bm[cfg][1][k][1], 1) * 1e3)
            print("  {:<44s} {:5.2f} x {:8.1f} us alone{}{}".format(k, calls, t / max(calls, 1) * 1e3, other, extra))


if __name__ == "__main__":
    args = sys.argv[1:] or ["bench_detail.json"]
    main(args[0], args[1] if len(args) > 1 and args[1].endswith(".json") else None,
         args[-1] if len(args) > 1 and not args[-1].endswith(".json") else "")
