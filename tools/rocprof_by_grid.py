#!/usr/bin/env python3
"""Per-(kernel, grid) durations from a rocprofv3 rocpd sqlite database (--kernel-trace): the same kernel runs at several pyramid
levels inside one window, and the per-kernel average hides which level is inefficient.  usage: rocprof_by_grid.py <db> <out.csv>"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gcols = [k for k in cols if "grid" in k.lower()]
    if not gcols:
        print("no grid columns in", cols); return
    gsel = ", ".join(gcols)
    q = (f"select name, {gsel}, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, sum(duration)/1e3 from kernels "
         f"group by name, {gsel} order by sum(duration) desc")
    rows = list(c.execute(q))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + gcols + ["calls", "avg_us", "min_us", "max_us", "total_us"])
        for r in rows:
            w.writerow([r[0][:90]] + list(r[1:1 + len(gcols)]) + [r[1 + len(gcols)]] + [f"{x:.2f}" for x in r[2 + len(gcols):]])
    print(f"{len(rows)} (kernel, grid) groups -> {out}; grid columns: {gcols}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
