#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are collected in
separate runs: their TCC slots do not fit one pass).

    python tools/hbm_traffic.py <fetch_results.db> <write_results.db> > profiles/hbm_traffic.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE tallies
the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as is
(uncalibrated).  bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches."""
import json
import os
import sqlite3
import sys

Q = """select s.kernel_name, sum(e.value), count(distinct d.id) from rocpd_pmc_event e
 join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
 join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by s.kernel_name"""


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd", "mggan", "hip"))
from ksym import short  # noqa: E402  (name<every template argument>: the spelling bench.py's roofline block uses)


def main(fetch_db, write_db):
    out = {}
    for db, name in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
        for sym, total, calls in sqlite3.connect(db).execute(Q, (name,)):
            e = out.setdefault(short(sym), {})
            e[name.lower() + "_kib_per_launch"] = total / calls
            e["launches_" + name.lower()] = calls
    for e in out.values():
        e["bytes_per_launch"] = int((2 * e.get("fetch_size_kib_per_launch", 0.0) + e.get("write_size_kib_per_launch", 0.0)) * 1024)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
