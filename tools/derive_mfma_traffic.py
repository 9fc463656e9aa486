#!/usr/bin/env python3
"""Derive profiles/<tag>_mfma_util_and_traffic_per_kernel.json from the rocprofv3 summaries of bench.py --no-parity --no-cpu-baseline
(config 2): <tag>_bench_kernel_stats.csv (durations), <tag>_pmc_sq_mfma_bench_window.json and <tag>_pmc_hbm_traffic_bench_window.json.
Every input is normalised PER WINDOW by the run it came from (tools/pmc_summary.py: totals / windows in that trace), so launch counts
and byte counts can no longer come from different runs.  usage: derive_mfma_traffic.py TAG WINDOWS_IN_STATS_RUN"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG_WINDOW_GB, ALG_GSTS_GB = 131.1, 70.8
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary import GSTS, short  # noqa: E402


def main():
    tag, nwin = sys.argv[1], int(sys.argv[2])
    P = os.path.join(ROOT, "profiles")
    stats = {}
    for r in csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))):
        stats[short(r["kernel"])] = r
    sq = json.load(open(os.path.join(P, f"{tag}_pmc_sq_mfma_bench_window.json")))["kernels_per_window"]
    tr = json.load(open(os.path.join(P, f"{tag}_pmc_hbm_traffic_bench_window.json")))["kernels_per_window"]
    out, tot, gsts, conv, ms_all = {}, 0.0, 0.0, 0.0, 0.0
    for k, r in stats.items():
        if k not in tr or "FETCH_SIZE_KB" not in tr[k] or "at::" in k or "rocclr" in k or "elementwise" in k:
            continue
        ms_win = float(r["total_us"]) / nwin / 1e3
        lpw = int(r["calls"]) / nwin
        gb = (2 * tr[k]["FETCH_SIZE_KB"] + tr[k].get("WRITE_SIZE_KB", 0.0)) * 1024 / 1e9
        e = {"launches_per_window": round(lpw, 2), "launches_per_window_in_pmc_run": tr[k]["launches_per_window"],
             "ms_per_window": round(ms_win, 3), "hbm_gb_per_window_2xFETCH_plus_WRITE": round(gb, 2), "hbm_tb_per_s": round(gb / ms_win, 2)}
        if k in sq:
            busy = sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            e.update({"mfma_insts_per_window": sq[k].get("SQ_INSTS_MFMA", 0.0), "valu_insts_per_window": sq[k].get("SQ_INSTS_VALU", 0.0),
                      "mfma_util_pct_at_2.4GHz": round(100 * busy / (ms_win * 1e-3 * 2.4e9 * 1024), 2),
                      "mfma_util_pct_at_2.0GHz": round(100 * busy / (ms_win * 1e-3 * 2.0e9 * 1024), 2)})
        out[k] = e
        tot += gb; ms_all += ms_win
        if any(s in k for s in GSTS):
            gsts += gb
        if "conv" in k and "shiftconv" not in k:
            conv += gb
    doc = {"note": f"Per-kernel time, HBM traffic and MFMA busy of bench.py --no-parity (config 2: Shift-Net-s 1280x720 one_len 16), every figure per "
                   f"WINDOW: {tag}_bench_kernel_stats.csv ({nwin} windows), {tag}_pmc_sq_mfma_bench_window.json, {tag}_pmc_hbm_traffic_bench_window.json "
                   "(each divided by the windows of its own trace).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (time x clock x 1024 SIMDs).  "
                   "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md).  Produced by tools/derive_mfma_traffic.py.",
           "kernels": out,
           "window_totals": {"kernel_ms": round(ms_all, 2), "hbm_gb_all_kernels": round(tot, 1), "hbm_gb_gsts_kernels": round(gsts, 1),
                             "hbm_gb_dense_conv_kernels": round(conv, 1), "algorithmic_gb_whole_window": ALG_WINDOW_GB,
                             "algorithmic_gb_gsts_units": ALG_GSTS_GB,
                             "traffic_over_algorithmic": {"window": round(tot / ALG_WINDOW_GB, 2), "gsts": round(gsts / ALG_GSTS_GB, 2)}}}
    dst = os.path.join(P, f"{tag}_mfma_util_and_traffic_per_kernel.json")
    json.dump(doc, open(dst, "w"), indent=1)
    print(dst, doc["window_totals"])
    for k, e in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_window"])[:14]:
        print(f"  {k:50s} {e['ms_per_window']:7.2f} ms  {e['hbm_gb_per_window_2xFETCH_plus_WRITE']:7.1f} GB  {e['hbm_tb_per_s']:5.2f} TB/s  mfma {e.get('mfma_util_pct_at_2.0GHz', 0):5.1f} %")


if __name__ == "__main__":
    main()
