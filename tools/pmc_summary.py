#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs (one directory per pass) into per-kernel averages per launch (JSON).

usage: pmc_summary.py OUT.json NOTE DIR [DIR ...]     (each DIR holds the *counter_collection.csv of one --pmc pass)"""
import csv
import glob
import json
import os
import sys


def main(out, note, dirs):
    kern = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"][:110]
                    k = kern.setdefault(name, {})
                    c = k.setdefault(row["Counter_Name"], [0.0, 0])
                    c[0] += float(row["Counter_Value"]); c[1] += 1
    res = {}
    for name, cs in kern.items():
        e = {}
        for cn, (tot, n) in cs.items():
            e[f"{cn}_KB_per_launch" if cn.endswith("_SIZE") else f"{cn}_per_launch"] = round(tot / n, 1)
            e["launches"] = n
        res[name] = e
    json.dump({"note": note, "kernels": res}, open(out, "w"), indent=1)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
