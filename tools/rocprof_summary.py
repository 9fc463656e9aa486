#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats) as a per-kernel CSV (durations in us)."""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
                          "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds_bytes"])
        for r in rows:
            w.writerow([r[0][:160], r[1], f"{r[2]:.1f}", f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}", f"{100 * r[2] / tot:.2f}", r[6], r[7], r[8], r[9]])
    print(f"{len(rows)} kernels, {tot / 1e3:.1f} ms total -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
