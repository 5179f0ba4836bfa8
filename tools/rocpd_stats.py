#!/usr/bin/env python
"""Summarise a rocprofv3 `*_results.db` (rocpd sqlite) as the `--stats` kernel table:
Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs — written as CSV.

    python tools/rocpd_stats.py gpurun_out/x/prof/r1_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path=None, skip_first=0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows[skip_first:]:
        d = e - s
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = open(out_path, "w", newline="") if out_path else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, (n, tot, mn, mx) in table:
        w.writerow([name, n, tot, round(tot / n, 1), round(100.0 * tot / total, 3), mn, mx])
    if out_path:
        out.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
