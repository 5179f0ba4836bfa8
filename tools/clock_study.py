#!/usr/bin/env python
"""Per-dispatch clock and MFMA utilisation from a rocprofv3 --pmc rocpd database (does the chip throttle the
north-star kernel by CLOCK or does the matrix pipe idle at full clock?).
    clock_ghz  = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
    mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD pipes x GRBM_GUI_ACTIVE / 8)
    python tools/clock_study.py db [db ...]"""
import re
import sqlite3
import sys


def rows(path):
    db = sqlite3.connect(path)
    per = {}
    for did, kn, cn, val, dur in db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
        a = per.setdefault(did, {"kernel": kn, "dur": dur})
        a[cn] = a.get(cn, 0.0) + val
    return [per[k] for k in sorted(per)]


def short(n):
    n = re.sub(r"ec::\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")[:60]


for path in sys.argv[1:]:
    print("#", path)
    print("dispatch,kernel,duration_us,clock_ghz,mfma_util,mfma_busy_cycles_per_simd,gui_cycles,wave_cycles_per_cu")
    for i, r in enumerate(rows(path)):
        gui = r.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui <= 0 or r["dur"] < 20000:
            continue
        busy = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        print(f'{i},"{short(r["kernel"])}",{r["dur"] / 1e3:.1f},{gui / r["dur"]:.3f},{busy / (1024.0 * gui):.4f},{busy / 1024.0:.0f},{gui:.0f},{r.get("SQ_WAVE_CYCLES", 0.0) / 256.0:.0f}')
