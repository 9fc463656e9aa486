#!/usr/bin/env python3
"""Derive profiles/<tag>_mfma_util_and_traffic_per_kernel.json from the three committed rocprofv3 summaries of bench.py (config 2):
<tag>_bench_kernel_stats.csv (durations), <tag>_pmc_sq_mfma_bench_window.json (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES) and
<tag>_pmc_hbm_traffic_bench_window.json (FETCH_SIZE, WRITE_SIZE).  usage: derive_mfma_traffic.py [tag] [windows_in_stats]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG_WINDOW_GB, ALG_GSTS_GB = 131.1, 70.8          # DESIGN.md section 3.3 (config 2)


def key(name):
    s = name.replace("(anonymous namespace)::", "").strip()
    if s.startswith("void "):
        s = s[5:]
    return s.split("(")[0].strip()[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 9        # make_profiles: --steps 5 --warmup 2 + 1 profiling step + the parity sample's window
    P = os.path.join(ROOT, "profiles")
    stats = {}
    for r in csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))):
        stats[key(r["kernel"])] = r
    sq = {key(k): v for k, v in json.load(open(os.path.join(P, f"{tag}_pmc_sq_mfma_bench_window.json")))["kernels"].items()}
    tr = {key(k): v for k, v in json.load(open(os.path.join(P, f"{tag}_pmc_hbm_traffic_bench_window.json")))["kernels"].items()}
    out, tot, gsts, conv = {}, 0.0, 0.0, 0.0
    for k, r in stats.items():
        if k not in tr or "FETCH_SIZE_KB_per_launch" not in tr[k] or "at::" in k or "rocclr" in k:
            continue
        calls, avg_us = int(r["calls"]), float(r["avg_us"])
        per_win = calls / nwin
        gb = (2 * tr[k]["FETCH_SIZE_KB_per_launch"] + tr[k].get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024 / 1e9
        e = {"launches_in_stats": calls, "avg_us": avg_us, "total_ms_per_window": round(avg_us * per_win / 1e3, 3),
             "hbm_gb_per_launch_2xFETCH_plus_WRITE": round(gb, 4), "hbm_tb_per_s": round(gb / avg_us * 1e3, 2), "hbm_gb_per_window": round(gb * per_win, 2)}
        if k in sq:
            busy = sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES_per_launch", 0.0)
            e.update({"mfma_insts_per_launch": sq[k].get("SQ_INSTS_MFMA_per_launch", 0.0), "valu_insts_per_launch": sq[k].get("SQ_INSTS_VALU_per_launch", 0.0),
                      "mfma_busy_cycles_per_launch": busy,
                      "mfma_util_pct_at_2.4GHz": round(100 * busy / (avg_us * 1e-6 * 2.4e9 * 1024), 2),
                      "mfma_util_pct_at_2.0GHz": round(100 * busy / (avg_us * 1e-6 * 2.0e9 * 1024), 2)})
        out[k] = e
        tot += e["hbm_gb_per_window"]
        if any(s in k for s in ("shiftconv", "ln_gemm_gate", "dw5m", "grp5", "scale_gemm_res", "ca_mlp")):
            gsts += e["hbm_gb_per_window"]
        if "conv" in k and "shiftconv" not in k:
            conv += e["hbm_gb_per_window"]
    doc = {"note": "Per-kernel MFMA busy and HBM traffic of bench.py (config 2: Shift-Net-s 1280x720 one_len 16) from the committed rocprofv3 passes: "
                   f"{tag}_bench_kernel_stats.csv (durations, {nwin} windows), {tag}_pmc_sq_mfma_bench_window.json, {tag}_pmc_hbm_traffic_bench_window.json.  "
                   "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs), the rocprofv3 MfmaUtil formula with GRBM_GUI_ACTIVE replaced by "
                   "duration x clock (2.4 GHz nominal = lower bound; the chip runs nearer 2.0 GHz under these kernels).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                   "(MI355X_MICROARCH.md).  Produced by tools/derive_mfma_traffic.py.",
           "kernels": out,
           "window_totals_gb": {"all_kernels": round(tot, 1), "gsts_kernels": round(gsts, 1), "dense_conv_kernels": round(conv, 1),
                                "algorithmic_whole_window": ALG_WINDOW_GB, "algorithmic_gsts_units": ALG_GSTS_GB,
                                "note": f"per window = per-launch average x launches per window (rocprofv3 stats run: {nwin} windows)"}}
    dst = os.path.join(P, f"{tag}_mfma_util_and_traffic_per_kernel.json")
    json.dump(doc, open(dst, "w"), indent=1)
    print(dst, doc["window_totals_gb"])
    for k, e in sorted(out.items(), key=lambda kv: -kv[1]["total_ms_per_window"])[:12]:
        print(f"  {k:50s} {e['total_ms_per_window']:7.2f} ms  {e['hbm_gb_per_window']:7.1f} GB  {e['hbm_tb_per_s']:5.2f} TB/s  mfma {e.get('mfma_util_pct_at_2.0GHz', 0):5.1f} %")


if __name__ == "__main__":
    main()
