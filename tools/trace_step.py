#!/usr/bin/env python
"""Print the kernel timeline of the LAST bench step from a rocprofv3 rocpd database (per-launch start, duration, grid)."""
import re
import sqlite3
import sys


def main(path, head_only=False):
    db = sqlite3.connect(path)
    rows = db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x,stream_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "im2col" in r[0]]
    step = rows[idx[-2]:]
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
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else False)
