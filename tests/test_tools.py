"""CPU: the measurement tooling's own logic (a wrong step boundary halved every per-step figure of round 6's first kernel-statistics files)."""
import csv
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _db(path, rows):
    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer, stream_id integer)")
    db.executemany("insert into kernels values (?, ?, ?, ?)", rows)
    db.commit()
    db.close()


def test_rocpd_stats_step_boundaries_and_per_step_columns(tmp_path):
    """A pipelined step opens with its im2col launches on the caller's stream (one per image source) while the PREVIOUS step's deferred head is
    still running on other streams: kernels of other streams between two im2col launches must not split the group, a GEMM on the caller's stream must."""
    import rocpd_stats
    rows, t = [], 0
    rows += [("weight_upload", t, t + 5, 0)]
    t += 10
    for step in range(4):
        rows += [("im2col14_kernel<3>", t, t + 2, 0), ("chain_kernel", t + 1, t + 4, 3), ("chain_kernel", t + 2, t + 5, 1), ("im2col14_kernel<3>", t + 3, t + 5, 0)]
        t += 6
        for blk in range(3):
            rows += [("gemm8_qkv", t, t + 10, 0), ("layernorm", t + 10, t + 12, 0)]
            t += 12
        rows += [("chain_kernel", t, t + 3, 3)]          # the step's own deferred head starts behind its backbone
        t += 3
    p = str(tmp_path / "r.db")
    _db(p, rows)
    db = sqlite3.connect(p)
    got = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    starts = rocpd_stats.step_starts(got)
    assert len(starts) == 4 and all("im2col" in got[i][0] for i in starts)
    out = str(tmp_path / "s.csv")
    rocpd_stats.main(p, out, 2)                                   # the last two steps only: the upload and the first two steps are cut off
    table = {r["Name"]: r for r in csv.DictReader(open(out))}
    assert "weight_upload" not in table
    assert int(table["gemm8_qkv"]["Calls"]) == 6 and float(table["gemm8_qkv"]["CallsPerStep(of 2)"]) == 3.0 and float(table["gemm8_qkv"]["NsPerStep"]) == 30.0
    assert int(table["im2col14_kernel<3>"]["Calls"]) == 4
    tot = table["TOTAL (kernel time, all streams)"]
    assert int(tot["Calls"]) == sum(int(r["Calls"]) for n, r in table.items() if not n.startswith("TOTAL"))
    rocpd_stats.main(p, out, 0)                                   # the whole trace, no per-step columns
    assert "weight_upload" in {r["Name"] for r in csv.DictReader(open(out))}
