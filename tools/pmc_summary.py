#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs (one directory per pass) of bench.py runs that contain ONLY full windows
(bench.py --no-parity --no-cpu-baseline: the small parity clip of the default run would dilute every per-launch average).

usage: pmc_summary.py OUT.json WINDOWS NOTE DIR [DIR ...]
    WINDOWS = windows in each traced run = warmup + steps + 1 (bench.py's extra per-kernel profiling step)
    each DIR holds the *counter_collection.csv of one --pmc pass (all passes must come from the same bench.py command line)

Output: per kernel symbol the counter totals PER WINDOW (sum over the symbol's launches / WINDOWS) and the launches per window; the
same keyed by (symbol, grid size) as per-launch averages (one pyramid level = one grid size); for FETCH_SIZE / WRITE_SIZE passes the
window totals in GB with HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide
coalesced reads on gfx950)."""
import csv
import glob
import json
import os
import sys

GSTS = ("shiftconv", "ln_gemm_gate", "dw5m", "grp5", "scale_gemm_res", "ca_mlp", "cab_phase1")      # cab_phase1 also matches cab_phase1r


def short(name):
    s = name.replace("(anonymous namespace)::", "").strip()
    if s.startswith("void "):
        s = s[5:]
    return s.split("(")[0].strip()[:80]


def main(out, windows, note, dirs):
    windows = int(windows)
    frac = float(os.environ.get("PMC_WINDOW_FRACTION", "1"))      # the traced steps ran this fraction of a window (config 4: one quadrant of four)
    tot, grid = {}, {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k, cn, v = short(row["Kernel_Name"]), row["Counter_Name"], float(row["Counter_Value"])
                    e = tot.setdefault(k, {}).setdefault(cn, [0.0, 0])
                    e[0] += v; e[1] += 1
                    g = grid.setdefault(f"{k} | grid {row.get('Grid_Size', '?')}", {}).setdefault(cn, [0.0, 0])
                    g[0] += v; g[1] += 1
    unit = lambda cn: "_KB" if cn.endswith("_SIZE") else ""
    per_win, by_grid = {}, {}
    for k, cs in tot.items():
        e = {}
        for cn, (s, n) in cs.items():
            e[cn + unit(cn)] = round(s / windows / frac, 1)
            e["launches_per_window"] = round(n / windows / frac, 3)
        per_win[k] = e
    for k, cs in grid.items():
        e = {}
        for cn, (s, n) in cs.items():
            e[cn + unit(cn) + "_per_launch"] = round(s / n, 1)
            e["launches_per_window"] = round(n / windows, 3)
        by_grid[k] = e
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_files, csrc_hash   # the sources these kernels were compiled from: bench.py ignores the file once one of them changes
    doc = {"note": note, "window_fraction_traced": frac, "csrc_hash": csrc_hash(), "csrc_files": csrc_files(per_win.keys()), "windows_in_trace": windows,
           "kernels_per_window": per_win, "by_grid_per_launch": by_grid}
    if any("FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e for e in per_win.values()):
        gb = lambda e: (2 * e.get("FETCH_SIZE_KB", 0.0) + e.get("WRITE_SIZE_KB", 0.0)) * 1024 / 1e9
        mine = {k: e for k, e in per_win.items() if "at::" not in k and "rocclr" not in k and "elementwise" not in k}
        doc["window_totals_gb"] = {
            "all_kernels": round(sum(gb(e) for e in mine.values()), 1),
            "gsts_kernels": round(sum(gb(e) for k, e in mine.items() if any(s in k for s in GSTS)), 1),
            "dense_conv_kernels": round(sum(gb(e) for k, e in mine.items() if ("conv" in k or "cabp" in k or "cab_fused" in k or "upsample2_add" in k) and "shiftconv" not in k), 1),
            "formula": "2 x FETCH_SIZE + WRITE_SIZE, summed over one window's launches"}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{len(per_win)} kernels, {len(by_grid)} (kernel, grid) groups, {windows} windows -> {out}", doc.get("window_totals_gb", ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
