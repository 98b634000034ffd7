#!/usr/bin/env python
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass:  python tools/pmc_table.py <results.db> [filter]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd", "mggan", "hip"))
from ksym import short  # noqa: E402


def main(path, flt=""):
    c = sqlite3.connect(path)
    q = """select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from rocpd_pmc_event e
 join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
 join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name"""
    tab, names = {}, []
    for k, n, v, cnt in c.execute(q):
        if flt and flt not in k:
            continue
        k = short(k)
        tab.setdefault(k, {})[n] = v / cnt
        tab[k]["_launches"] = cnt
        if n not in names:
            names.append(n)
    names.sort()
    print("{:<44s} {:>6s} ".format("kernel", "calls") + " ".join("{:>14s}".format(n[-14:]) for n in names))
    for k in sorted(tab, key=lambda k: -tab[k].get("SQ_BUSY_CYCLES", tab[k].get(names[0], 0))):
        print("{:<44s} {:>6d} ".format(k[:44], tab[k]["_launches"]) + " ".join("{:>14.4g}".format(tab[k].get(n, 0)) for n in names))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
