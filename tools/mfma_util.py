#!/usr/bin/env python
"""MFMA-pipe utilisation per kernel from one rocprofv3 --pmc pass that collected SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE (tools/profile_round.sh):

    python tools/mfma_util.py <results.db> > profiles/mfma_util_<cfg>.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 32 CUs x GRBM_GUI_ACTIVE)  -- the gfx94x `MfmaUtil` formula
(ROCm 7.2 ships no gfx950 derived-counter section, MI355X_MICROARCH.md "rocprofv3 PMC slots") with the XCD
aggregation made explicit: busy cycles are summed over every SIMD of the chip, and rocprofv3 reports GRBM_GUI_ACTIVE
summed over the 8 XCDs (calibration: decoder_bwd_mfma_kernel at 256x32, g=8: 23.7e6 "active cycles" for a 1.20 ms
dispatch = 8 x 2.47 GHz x 1.20 ms), so the per-SIMD time base is GRBM_GUI_ACTIVE / 8 and the chip has
4 x 256 SIMDs: busy / (4 * 256 * GUI / 8).  SQ_VALU_MFMA_BUSY_CYCLES = 32 x SQ_INSTS_MFMA holds for every kernel here.
One `v_mfma_f32_16x16x4_f32` keeps its SIMD's matrix pipe busy for 32 cycles = 2048 FLOP, i.e. 64 FLOP/clk/SIMD:
mfma_util x 157.3 TFLOP/s is the f32 rate the MFMA pipe actually delivered."""
import json
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from hbm_traffic import short  # noqa: E402

Q = """select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from rocpd_pmc_event e
 join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
 join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name"""


def main(db):
    tab = {}
    for sym, name, total, calls in sqlite3.connect(db).execute(Q):
        e = tab.setdefault(short(sym), {})
        e[name] = total / calls
        e["launches"] = calls
    out = {}
    for k, e in tab.items():
        busy, act = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("GRBM_GUI_ACTIVE")
        if not busy or not act:
            continue
        out[k] = {"mfma_busy_cycles_per_launch": busy, "gui_active_cycles_per_launch": act,
                  "mfma_util": round(busy / (4 * 32 * act), 5), "launches": e["launches"]}
        for extra in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
            if extra in e:
                out[k][extra.lower() + "_per_launch"] = e[extra]
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
