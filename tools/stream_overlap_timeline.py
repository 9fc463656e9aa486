#!/usr/bin/env python3
"""How much of the GSTS kernels' time overlaps under the streams schedule?  Reads a rocprofv3 rocpd database (--kernel-trace) and reports, for the
LAST traced window of bench.py: the union (wall) and the sum of the kernel intervals per kernel family, and the time during which a phase-2
kernel (scale_gemm_res_kernel, K4) or K0 (shiftconv_kernel) runs WHILE a phase-1 kernel (cab_phase1r_kernel) is running.
usage: stream_overlap_timeline.py trace.db out.txt [windows_in_trace]"""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def intersect(a, b):
    """total length of the intersection of two interval unions"""
    a, b = sorted(a), sorted(b)
    ev = [(s, 0, 1) for s, e in a] + [(e, 0, -1) for s, e in a] + [(s, 1, 1) for s, e in b] + [(e, 1, -1) for s, e in b]
    ev.sort(key=lambda x: (x[0], x[2]))
    cnt = [0, 0]
    last, tot = None, 0
    for t, k, d in ev:
        if last is not None and cnt[0] > 0 and cnt[1] > 0:
            tot += t - last
        cnt[k] += d
        last = t
    return tot


def main(db, out, nwin=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sc = "start" if "start" in cols else "start_timestamp"
    ec = "end" if "end" in cols else "end_timestamp"
    qc = next((k for k in ("stream_id", "queue_id", "stream", "queue") if k in cols), None)
    rows = list(c.execute(f"select name, {sc}, {ec}" + (f", {qc}" if qc else ", 0") + f" from kernels order by {sc}"))
    lines = [f"columns of `kernels`: {cols}", f"{len(rows)} kernel dispatches in the trace"]
    # one window = the dispatches between two ingest kernels
    starts = [i for i, r in enumerate(rows) if "ingest_kernel" in r[0]]
    if len(starts) >= 2:
        lo, hi = starts[-2], starts[-1]          # the second-to-last window is complete on both sides
    else:
        lo, hi = 0, len(rows)
    win = rows[lo:hi]
    fam = {"phase 1 (cab_phase1r_kernel)": "cab_phase1r_kernel", "K4 (scale_gemm_res_kernel)": "scale_gemm_res_kernel", "K0 (shiftconv_kernel)": "shiftconv_kernel",
           "dense convs (conv3_fast / conv_mfma)": "conv"}
    iv = {k: [(r[1], r[2]) for r in win if p in r[0]] for k, p in fam.items()}
    allk = [(r[1], r[2]) for r in win]
    lines.append(f"window: dispatches {lo}..{hi}: {len(win)} kernels on {len(set(r[3] for r in win))} queues/streams, wall {(win[-1][2] - win[0][1]) / 1e6:.2f} ms, "
                 f"sum of kernel durations {sum(e - s for s, e in allk) / 1e6:.2f} ms, union {union(allk) / 1e6:.2f} ms")
    for k, v in iv.items():
        lines.append(f"  {k:40s} n={len(v):5d} sum {sum(e - s for s, e in v) / 1e6:8.2f} ms   union {union(v) / 1e6:8.2f} ms")
    gsts = iv["phase 1 (cab_phase1r_kernel)"] + iv["K4 (scale_gemm_res_kernel)"] + iv["K0 (shiftconv_kernel)"]
    lines.append(f"  GSTS kernels: sum {sum(e - s for s, e in gsts) / 1e6:.2f} ms, union (wall time they occupy) {union(gsts) / 1e6:.2f} ms")
    p1 = iv["phase 1 (cab_phase1r_kernel)"]
    for k in ("K4 (scale_gemm_res_kernel)", "K0 (shiftconv_kernel)"):
        lines.append(f"  {k}: {intersect(iv[k], p1) / 1e6:.2f} ms of its {union(iv[k]) / 1e6:.2f} ms (union) run while a phase-1 kernel is running")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
