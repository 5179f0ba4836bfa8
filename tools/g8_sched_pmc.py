#!/usr/bin/env python
"""Map the gemm8 dispatches of `tools/g8_sched.py replay` runs under rocprofv3 --pmc back to the manifest's configurations.

    python tools/g8_sched_pmc.py --manifest m.json --iters 3 db1 [db2 ...] [--out file.csv]

Every database is one pass (different counters) of the SAME replay: the i-th group of `iters` consecutive gemm8 dispatches belongs
to configuration i.  Values are summed over instances per dispatch and averaged over the group's dispatches after the first.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled on output (MI355X_MICROARCH.md, 16-B/lane streaming reads on gfx950)."""
import argparse
import csv
import json
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--manifest", required=True)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out")
    a = ap.parse_args()
    man = json.load(open(a.manifest))["configs"]
    cols = {}
    for path in a.dbs:
        db = sqlite3.connect(path)
        rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
        per = {}
        for did, kn, cn, val, dur in rows:
            if "gemm8_bf16_kernel" not in kn:
                continue
            d = per.setdefault(did, {})
            d[cn] = d.get(cn, 0.0) + val
            d["_dur_" + cn] = dur
        ids = sorted(per)
        if len(ids) != len(man) * a.iters:
            print(f"warning: {path}: {len(ids)} gemm8 dispatches, expected {len(man) * a.iters}", file=sys.stderr)
        for i, e in enumerate(man):
            grp = ids[i * a.iters + 1:(i + 1) * a.iters]
            for cn in {c for g in grp for c in per[g]}:
                vals = [per[g][cn] for g in grp if cn in per[g]]
                cols.setdefault(i, {})[cn] = sum(vals) / max(len(vals), 1)
    names = sorted({c for v in cols.values() for c in v if not c.startswith("_dur_")})
    out = open(a.out, "w", newline="") if a.out else sys.stdout
    w = csv.writer(out)
    w.writerow(["shape", "krot_n", "krot_m", "stagger", "us_unprofiled", "operand_MB", "min_fetch_MB(A+8B)"] + names + ["fetch_MB(x2)", "dur_us_profiled"])
    for i, e in enumerate(man):
        v = cols.get(i, {})
        A, B = e["M"] * e["K"] * 2 / 1e6, e["N"] * e["K"] * 2 / 1e6
        fetch = v.get("FETCH_SIZE")
        durs = [v[c] for c in v if c.startswith("_dur_")]
        w.writerow([e["shape"], e["krot"] & 255, e["krot"] >> 8, e["stagger"], round(e["us"], 1), round(A + B, 1), round(A + 8 * B, 1)] +
                   [round(v.get(n, float("nan")), 1) for n in names] +
                   [round(fetch * 2 * 1024 / 1e6, 1) if fetch is not None else "", round(sum(durs) / len(durs) / 1e3, 1) if durs else ""])
    if a.out:
        out.close()


if __name__ == "__main__":
    main()
