#!/usr/bin/env python
"""One steady-state iteration of a rocprofv3 --kernel-trace run as a listing (start offset, duration, queue, kernel)
so that the critical path through the two-stream step graph can be read off.
    python tools/iter_trace.py <results.db> <optimizer_steps_per_iteration = 3> [iteration_from_end]"""
import re
import sqlite3
import sys


def short(sym):
    m = re.match(r"_Z(\d+)", sym)
    if not m:
        return sym[:40]
    n = int(m.group(1))
    return sym[m.end():m.end() + n] + sym[m.end() + n:m.end() + n + 8]


def main(path, per, back=2):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = c.execute("select d.start, d.end, d.{}, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start".format(qcol)).fetchall()
    # iteration boundary = every `per`-th launch of the optimizer kernel (3 optimizer steps per iteration)
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3]]
    hi = marks[-1 - per * (back - 1)] + 1
    lo = marks[-1 - per * back] + 1
    it = rows[lo:hi]
    t0 = it[0][0]
    print("# columns: start_us dur_us queue kernel   (available dispatch columns: {})".format(",".join(cols)))
    last_end = {}
    for s, e, q, k in it:
        gap = (s - last_end.get(q, s)) / 1e3
        print("{:9.1f} {:7.1f} q{} {}{}".format((s - t0) / 1e3, (e - s) / 1e3, q, short(k), "   <- queue idle {:.1f} us".format(gap) if gap > 3 else ""))
        last_end[q] = e
    print("# iteration span {:.1f} us".format((max(r[1] for r in it) - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2)
