#!/usr/bin/env python
"""Summarise a rocprofv3 `*_results.db` (rocpd sqlite) as the `--stats` kernel table:
Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs — written as CSV.

    python tools/rocpd_stats.py gpurun_out/x/prof/r1_results.db profiles/r01_bench_kernel_stats.csv [last_steps]
"""
import csv
import sqlite3
import sys


def step_starts(rows):
    """indices of the launches that open a step / call: its im2col launches (one per image source, back to back on the caller's stream -
    kernels of OTHER streams, the previous step's deferred head, may lie between them in the trace and do not split the group)"""
    idx = [i for i, r in enumerate(rows) if "im2col" in r[0]]
    return [i for j, i in enumerate(idx) if j == 0 or not all("im2col" in rows[x][0] or rows[x][3] != rows[i][3] for x in range(idx[j - 1], i))]


def main(db_path, out_path=None, last_steps=0):
    """last_steps = N > 0: only the launches of the LAST N steps of the trace (a step opens with its im2col launches) - the timed steps of one
    bench leg without its warm-up, the engine's build and the weight uploads - with a per-step column, so per-step sums can be read
    straight from the CSV."""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end, stream_id from kernels order by start").fetchall()
    if last_steps > 0:
        st = step_starts(rows)
        if len(st) < last_steps:
            raise SystemExit(f"trace holds {len(st)} steps, {last_steps} asked for")
        rows = rows[st[-last_steps]:]
    agg = {}
    for name, s, e, _ in rows:
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
    head = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]
    if last_steps > 0:
        head += [f"CallsPerStep(of {last_steps})", "NsPerStep"]
    w.writerow(head)
    for name, (n, tot, mn, mx) in table:
        row = [name, n, tot, round(tot / n, 1), round(100.0 * tot / total, 3), mn, mx]
        if last_steps > 0:
            row += [round(n / last_steps, 2), round(tot / last_steps, 1)]
        w.writerow(row)
    if last_steps > 0:
        w.writerow(["TOTAL (kernel time, all streams)", sum(a[0] for a in agg.values()), total, "", 100.0, "", "", round(sum(a[0] for a in agg.values()) / last_steps, 2), round(total / last_steps, 1)])
    if out_path:
        out.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
