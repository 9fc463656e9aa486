#!/usr/bin/env python3
"""Dense CAB (gshift_deblur1.py:141-156) on MI355X: the two-launch form (sn_conv2d x 2) against the fused tile form (sn_cab_stats + sn_cab_fused,
csrc/sn_cabf.hip) with 8- and 16-row tiles.  Every variant is timed once per round, the rounds alternate the order (clock / power state moves
back-to-back timings by 10 % on this part); min / median per variant in us per CAB, plus the per-launch split of one profiled call.
usage: cab_ab.py [--rounds 6] [--reps 10] [--cases 16x20x720x1280,24x20x360x640,24x52x720x1280]   (channels x T x h x w)"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

CABS = {14: ("gshift_deblur2", "stage1.concat."), 18: ("gshift_deblur2", "orb1.encoder_level2.1."), 22: ("gshift_deblur2", "orb1.encoder_level3.1."),
        24: ("gshift_deblur1", "stage1.concat."), 36: ("gshift_deblur1", "orb1.encoder_level2.0."), 48: ("gshift_deblur1", "orb1.encoder_level3.0."),
        64: ("gshift_deblur2", "stage1.skip_attn1."), 80: ("gshift_deblur1", "stage1.skip_attn1.")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cases", default="14x20x720x1280,18x20x360x640,22x20x180x320,24x52x720x1280")
    ap.add_argument("--variants", default="0,8,16")
    a = ap.parse_args()
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    dev = torch.device("cuda:0")
    plans = {}
    for case in a.cases.split(","):
        c, T, h, w = (int(v) for v in case.split("x"))
        name, pre = CABS[c]
        if name not in plans:
            plans[name] = Plan(VARIANTS[name], synth_state_dict(name), dev)
        engs = {}
        for v in a.variants.split(","):
            vfull, v = v, v.split("/")[0]
            e = Engine(plans[name])                                 # "0": two launches on the streaming conv kernel; "t": on the tile kernel; "wN": N workgroups
            e.cab_fused = v if v in ("8", "16", "s8", "s16", "p", "p16") else "0"   # per CU; "8" / "16": the fused tile form; "s8": with the statistics pass on the streaming kernel
            e.conv_tiles = v == "t"
            e.conv_wgs = int(v[1:]) if v.startswith("w") else 0
            e.conv_stream_all = v != "d"                            # "d": the library's default routing (16-channel conv2 on the tile kernel)
            e.conv_res_regs = v.startswith("r") and v[1:].isdigit() or v == "r"                     # "r": residual through registers; "rN": with N workgroups per CU
            if v.startswith("r") and len(v) > 1:
                e.conv_wgs = int(v[1:])
            if "/" in vfull:                                        # "<variant>/dN/wM": prefetch depth code N and M workgroups per CU on top of a variant
                for opt in vfull.split("/")[1:]:
                    if opt[0] == "d":
                        e.conv_depth = int(opt[1:])
                    if opt[0] == "w":
                        e.conv_wgs = int(opt[1:])
                    if opt[0] == "x":                               # ablations (wrong results): xN = SN_CONV_DBG bits
                        e.conv_dbg = int(opt[1:])
            v = vfull
            engs[v] = e
        cs = (c + 7) // 8 * 8
        x = torch.zeros((T, h, w, cs), dtype=torch.bfloat16, device=dev)
        x[..., :c] = torch.randn((T, h, w, c), device=dev).to(torch.bfloat16)
        xa = Act(x, c)
        times = {v: [] for v in engs}
        outs = {}
        for v, e in engs.items():                                   # warm up + bit identity
            outs[v] = e.cab(pre, xa).t
        torch.cuda.synchronize()
        ident = {v: bool(torch.equal(outs[next(iter(outs))], o)) for v, o in outs.items()}
        for r in range(a.rounds):
            order = list(engs) if r % 2 == 0 else list(engs)[::-1]
            for v in order:
                e = engs[v]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    e.cab(pre, xa)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        gb = 2 * T * h * w * cs * 2 / 1e9
        print(f"== CAB {c} channels (storage {cs}), T={T} {h}x{w}: algorithmic {gb:.3f} GB (x in, out), bit-identical to the first variant: {ident}", flush=True)
        for v in engs:
            e = engs[v]
            e.prof = []
            e.cab(pre, xa)
            torch.cuda.synchronize()
            split = ", ".join(f"{fn.replace('sn_', '')} {e0.elapsed_time(e1) * 1e3:.0f}" for fn, _, _, e0, e1 in e.prof)
            e.prof = None
            mn, md = min(times[v]), statistics.median(times[v])
            print(f"AB cab{c}_{T}x{h}x{w} variant {v:>2s}: min {mn:8.1f} us  median {md:8.1f} us  = {gb / (md * 1e-6) / 1e3:5.2f} TB/s algorithmic   [{split}]", flush=True)


if __name__ == "__main__":
    main()
