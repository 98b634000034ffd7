#!/usr/bin/env python
"""Timeline view of a rocprofv3 --kernel-trace result: for the steady-state part of the run (the last
`frac` of the dispatches), the wall span, the union of kernel-busy intervals, the idle gaps between kernels
and the per-kernel totals.   python tools/timeline.py <results.db> [frac] [iterations_in_that_part]"""
import sqlite3
import sys


def main(path, frac=0.5, iters=None):
    c = sqlite3.connect(path)
    rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    n = len(rows)
    rows = rows[int(n * (1 - frac)):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, gaps, cur_end = 0, [], rows[0][0]
    for s, e, _ in rows:
        if s > cur_end:
            gaps.append(s - cur_end)
            busy += 0
            cur_start = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    span = t1 - t0
    tot = sum(e - s for s, e, _ in rows)
    print("# {}: last {} of {} dispatches".format(path, len(rows), n))
    print("span {:.1f} us  busy(union) {:.1f} us  idle {:.1f} us ({:.1f}%)  sum of durations {:.1f} us (overlap {:.1f} us)".format(
        span / 1e3, busy / 1e3, (span - busy) / 1e3, 100.0 * (span - busy) / span, tot / 1e3, (tot - busy) / 1e3))
    if gaps:
        gaps.sort()
        print("gaps: n={} median {:.2f} us  p90 {:.2f} us  max {:.2f} us  total {:.1f} us".format(
            len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3, sum(gaps) / 1e3))
    if iters:
        print("per iteration: span {:.1f} us, busy {:.1f} us, dispatches {:.1f}".format(span / 1e3 / iters, busy / 1e3 / iters,
                                                                                       len(rows) / iters))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, int(sys.argv[3]) if len(sys.argv) > 3 else None)
