#!/usr/bin/env python
"""Per-kernel averages of the counters in a rocprofv3 --pmc rocpd database (sum over instances per dispatch, mean over
dispatches).  python tools/rocpd_pmc.py db [db ...] [--out file.csv]"""
import csv
import re
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
    per = {}
    for did, kn, cn, val, dur in rows:
        k = (did, kn, cn)
        a = per.setdefault(k, [0.0, dur])
        a[0] += val
    agg = {}
    for (did, kn, cn), (v, dur) in per.items():
        a = agg.setdefault((kn, cn), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += v
        a[2] += dur
    return agg


def short(n):
    n = re.sub(r"ec::\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")


def main(argv):
    out = None
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    table = []
    for path in argv:
        for (kn, cn), (n, v, dur) in sorted(summarise(path).items()):
            table.append([short(kn), cn, n, round(v / n, 3), round(dur / n, 1)])
    f = open(out, "w", newline="") if out else sys.stdout
    w = csv.writer(f)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanValuePerDispatch", "MeanDurationNs"])
    w.writerows(table)
    if out:
        f.close()


if __name__ == "__main__":
    main(sys.argv[1:])
