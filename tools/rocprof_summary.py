#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) into a per-kernel table:
    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x_kernel_stats.txt
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd", "mggan", "hip"))
from ksym import short  # noqa: E402


def main(path, top=60):
    c = sqlite3.connect(path)
    tot, n = c.execute("select sum(end-start)/1e3, count(*) from rocpd_kernel_dispatch").fetchone()
    print("# rocprofv3 --kernel-trace summary of {}".format(path))
    print("# total kernel time {:.1f} us over {} dispatches".format(tot, n))
    print("# kernel = name<every template argument> (mggan/hip/ksym.py; the spelling of bench.py's roofline block and of the "
          "counter tables), mangled symbol in the last column")
    print("# {:<56s} {:>7s} {:>11s} {:>9s} {:>8s} {:>9s} {:>6s}  {}".format("kernel", "calls", "total_us", "avg_us", "min_us",
                                                                            "max_us", "pct", "symbol"))
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, "
         "max(d.end-d.start)/1e3 from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.kernel_name order by 3 desc limit {}".format(top))
    for r in c.execute(q):
        print("  {:<56s} {:>7d} {:>11.1f} {:>9.2f} {:>8.2f} {:>9.2f} {:>5.1f}%  {}".format(short(r[0])[:56], r[1], r[2], r[3], r[4],
                                                                                          r[5], 100 * r[2] / tot, r[0][:60]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
