#!/usr/bin/env python
"""Print the kernel timeline of the LAST bench step from a rocprofv3 rocpd database (per-launch start, duration, grid)."""
import re
import sqlite3
import sys


def main(path, head_only=False, which=-1):
    """which: -1 the last step (the last two im2col launches open it: query + support sources); k < -1: the step |k| - 1 steps before
    the end of the trace (e.g. the last HEADLINE step when an episode leg follows it)."""
    db = sqlite3.connect(path)
    rows = db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x,stream_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "im2col" in r[0]]
    # a step opens with its im2col launches (one per image source, back to back on the caller's stream): group consecutive ones
    starts = [i for j, i in enumerate(idx) if j == 0 or not all("im2col" in rows[x][0] or rows[x][7] != rows[i][7] for x in range(idx[j - 1], i))]
    first = starts[which]
    last = starts[which + 1] if which < -1 else len(rows)
    step = rows[first:last]
    t0 = step[0][1]
    agg = {}
    tot = 0
    for r in step:
        n = re.sub(r"ec::\(anonymous namespace\)::", "", r[0])
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        d = (r[2] - r[1]) / 1e3
        tot += d
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += d
        if not head_only or (r[1] - t0) / 1e3 > head_only:
            print(f"{(r[1]-t0)/1e3:9.1f} {d:8.1f}  s{r[7]} grid {r[3]//max(r[6],1):5d}x{r[4]}x{r[5]}  {n[:70]}")
    print("---- per-kernel totals of the step (us)")
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{d:9.1f} {c:4d}  {n[:90]}")
    print("sum", round(tot, 1), "span", round((step[-1][2] - t0) / 1e3, 1))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 and float(sys.argv[2]) > 0 else False, int(sys.argv[3]) if len(sys.argv) > 3 else -1)
