#!/usr/bin/env python3
"""<DIR>/<TAG>_mfma_util_and_traffic_per_kernel_<NAME>.json from the three summaries of one configuration written by tools/make_profiles_r05.sh:
<TAG>_<NAME>_kernel_stats.csv (rocprofv3 --kernel-trace --stats: durations), <TAG>_pmc_hbm_traffic_<NAME>.json (FETCH_SIZE / WRITE_SIZE passes) and
<TAG>_pmc_sq_<NAME>.json (one SQ pass).  Every figure is per WINDOW of the configuration, each input normalised by the windows of its own trace.
usage: derive_r05.py DIR TAG NAME WINDOWS_IN_STATS_RUN [WINDOW_FRACTION]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import GSTS, short  # noqa: E402


def main():
    d, tag, name, nwin = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    frac = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    stats = {}
    for r in csv.DictReader(open(os.path.join(d, f"{tag}_{name}_kernel_stats.csv"))):
        stats[short(r["kernel"])] = r
    tr = json.load(open(os.path.join(d, f"{tag}_pmc_hbm_traffic_{name}.json")))
    sq = json.load(open(os.path.join(d, f"{tag}_pmc_sq_{name}.json")))["kernels_per_window"]
    trk = tr["kernels_per_window"]
    out, tot, gsts, conv, ms_all = {}, 0.0, 0.0, 0.0, 0.0
    for k, r in stats.items():
        if k not in trk or "FETCH_SIZE_KB" not in trk[k] or "at::" in k or "rocclr" in k or "elementwise" in k:
            continue
        ms_win = float(r["total_us"]) / nwin / frac / 1e3
        gb = (2 * trk[k]["FETCH_SIZE_KB"] + trk[k].get("WRITE_SIZE_KB", 0.0)) * 1024 / 1e9
        e = {"launches_per_window": round(int(r["calls"]) / nwin / frac, 2), "ms_per_window": round(ms_win, 3),
             "hbm_gb_per_window_2xFETCH_plus_WRITE": round(gb, 2), "hbm_tb_per_s": round(gb / ms_win, 2)}
        q = sq.get(k)
        if q:
            busy, wave = q.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), max(q.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            e.update({"mfma_insts_per_window": q.get("SQ_INSTS_MFMA", 0.0), "valu_insts_per_window": q.get("SQ_INSTS_VALU", 0.0),
                      "mfma_util_pct_at_2.4GHz": round(100 * busy / (ms_win * 1e-3 * 2.4e9 * 1024), 2),
                      "mfma_util_pct_at_2.0GHz": round(100 * busy / (ms_win * 1e-3 * 2.0e9 * 1024), 2),
                      "wave_cycles_waiting_pct": round(100 * q.get("SQ_WAIT_ANY", 0.0) / wave, 1),
                      "wave_cycles_issuing_pct": round(100 * q.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 1),
                      "lds_bank_conflict_pct_of_lds_cycles": round(100 * q.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(q.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 1)})
        out[k] = e
        tot += gb; ms_all += ms_win
        if any(s in k for s in GSTS):
            gsts += gb
        if ("conv" in k or "cabp" in k or "cab_fused" in k or "upsample2_add" in k) and "shiftconv" not in k:
            conv += gb
    doc = {"note": f"Per-kernel time, HBM traffic, MFMA busy, wave wait / issue share and LDS bank conflicts of bench.py --no-parity --no-cpu-baseline ({name}), every "
                   f"figure per WINDOW: {tag}_{name}_kernel_stats.csv ({nwin} traced steps, each {frac} of a window), {tag}_pmc_hbm_traffic_{name}.json, "
                   f"{tag}_pmc_sq_{name}.json (each divided by the windows of its own trace).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (time x clock x 1024 SIMDs); "
                   "waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked at s_waitcnt / barrier), issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; HBM bytes = "
                   "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md).  Produced by tools/derive_r05.py.",
           "csrc_files": tr.get("csrc_files"), "kernels": out,
           "window_totals": {"kernel_ms": round(ms_all, 2), "hbm_gb_all_kernels": round(tot, 1), "hbm_gb_gsts_kernels": round(gsts, 1),
                             "hbm_gb_dense_conv_kernels": round(conv, 1)}}
    dst = os.path.join(d, f"{tag}_mfma_util_and_traffic_per_kernel_{name}.json")
    json.dump(doc, open(dst, "w"), indent=1)
    print(dst, doc["window_totals"])
    for k, e in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_window"])[:12]:
        print(f"  {k:46s} {e['ms_per_window']:7.2f} ms {e['hbm_gb_per_window_2xFETCH_plus_WRITE']:7.1f} GB {e['hbm_tb_per_s']:5.2f} TB/s  mfma {e.get('mfma_util_pct_at_2.0GHz', 0):5.1f} %"
              f"  wait {e.get('wave_cycles_waiting_pct', 0):5.1f} %  lds-conflict {e.get('lds_bank_conflict_pct_of_lds_cycles', 0):5.1f} %")


if __name__ == "__main__":
    main()
