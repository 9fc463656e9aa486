#!/usr/bin/env python3
"""Interleaved A/B timing of phase-1 builds on one GPU: the round-4 library (lib/dev/libshiftnet_r04.so, built from git history with the old ABI),
the current kernel and its measurement variants (lib/dev/libp1r_*.so, tools/p1r_variants.py build ...).  Every round times every variant once
(1 warm-up + N launches between events), rounds alternate the order; min and median over the rounds are printed -- single back-to-back
measurements on this part differ by 10-15 % with the order they are taken in (clock / power state), which is more than most of the effects
being measured.   usage: p1_ab.py [--rounds 6] [--reps 4]"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
DEV = os.path.join(ROOT, "shift-net_amd", "lib", "dev")


class OldWeights(C.Structure):      # sn_phase1_weights of ABI 13
    _fields_ = [("wfrag1", C.c_void_p), ("wfragx", C.c_void_p), ("w3", C.c_void_p), ("w5", C.c_void_p), ("wfrag2", C.c_void_p), ("wgrp", C.c_void_p), ("layout", C.c_int)]


def main():
    import torch
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Plan
    from shiftnet_amd.spec import VARIANTS as SV
    from shiftnet_amd.weights import synth_state_dict

    def arg(k, d):
        return sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
    rounds, reps = int(arg("--rounds", "6")), int(arg("--reps", "4"))
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    vp, ci = C.c_void_p, C.c_int
    libs = {}
    for f in sorted(os.listdir(DEV)):
        if f == "libshiftnet_r04.so":
            libs["r04"] = (C.CDLL(os.path.join(DEV, f)), True)
        elif f.startswith("libp1r_") and f.endswith(".so"):
            libs[f[7:-3]] = (C.CDLL(os.path.join(DEV, f)), False)
    k0libs = {"r04": libs["r04"][0]} if "r04" in libs else {}
    k0libs["cur"] = C.CDLL(L.LIB_PATH)
    for f in sorted(os.listdir(DEV)):          # whole-library variants (build.py build(out=..., flags=[...])): K0 / K4 rows
        if f.startswith("libshiftnet_") and f.endswith(".so") and f != "libshiftnet_r04.so":
            k0libs[f[12:-3]] = C.CDLL(os.path.join(DEV, f))
    for lib in k0libs.values():
        lib.sn_gsts_shiftconv.argtypes = [C.POINTER(L.UnitSrc), vp, vp, vp, vp]
        lib.sn_gsts_cab2_phase2.argtypes = [C.POINTER(L.UnitSrc), vp, vp, vp, vp, vp, vp]
        lib.sn_cab1_phase2.argtypes = [C.POINTER(L.UnitSrc), vp, vp, vp, vp, vp, vp]
    if "--k0-only" in sys.argv:
        libs = {}
    for name, (lib, old) in libs.items():
        W = OldWeights if old else L.Phase1Weights
        lib.sn_gsts_cab2_phase1.argtypes = [C.POINTER(L.UnitSrc), vp, C.POINTER(W), vp, vp, vp, vp, vp]
        lib.sn_cab1_phase1.argtypes = [C.POINTER(L.UnitSrc), C.POINTER(W), vp, vp, vp, vp, vp]
    cases = [("gshift_deblur2", (20, 360, 640)), ("gshift_deblur2", (20, 180, 320)), ("gshift_deblur1", (52, 360, 640)), ("gshift_deblur1", (16, 540, 960))]
    if "--quick" in sys.argv:
        cases = cases[:1]
    for model, (T, h, w) in cases:
        V = SV[model]
        P = Plan(V, synth_state_dict(model), dev)
        Cc = V.c1
        xd = torch.randn(T, h, w, Cc, device=dev).to(torch.bfloat16)
        hwb = torch.randn(T, h, w, Cc // 2, device=dev).to(torch.bfloat16)
        g2 = torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev)
        pool = torch.zeros((T, 4096, Cc), dtype=torch.float32, device=dev)          # larger than any build's pool
        calls = {}
        for name, (lib, old) in libs.items():
            teams = [0] if (old or name != "base") else [0, 1, 2, 4, 8]
            for team in teams:
                for mode, unit in ((0, "encoder_level1.1."), (1, "encoder_level1.0.")):
                    d = P.units["stage1.decoder_level1." + unit]["p1r"]
                    if old:
                        wt = OldWeights(d["wfrag1"].data_ptr(), None, d["w3"].data_ptr(), None, d["wfrag2"].data_ptr(), d["wgrp"].data_ptr(), 1)
                    else:
                        wt = d["desc"]
                    src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 0)
                    opt = None if old else L.Phase1Opts(None, 0, team)
                    op = C.byref(opt) if opt is not None else None
                    if mode:
                        f = (lambda lib=lib, src=src, wt=wt, op=op: lib.sn_gsts_cab2_phase1(C.byref(src), hwb.data_ptr(), C.byref(wt), g2.data_ptr(), pool.data_ptr(), None, op, st))
                    else:
                        f = (lambda lib=lib, src=src, wt=wt, op=op: lib.sn_cab1_phase1(C.byref(src), C.byref(wt), g2.data_ptr(), pool.data_ptr(), None, op, st))
                    f.keep = (src, wt, opt)
                    calls[(name + (f"_team{team}" if team else ""), "CAB2" if mode else "CAB1")] = f
        # K0 (sn_gsts_shiftconv): round-4 library against the current one, forward and reverse units
        for name, lib in k0libs.items():
            for mode, unit in ((1, "encoder_level1.0."), (2, "encoder_level1_1.0.")):
                u = P.units["stage1.decoder_level1." + unit]
                src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 1 if V.wrap else 0)
                f = (lambda lib=lib, src=src, u=u: lib.sn_gsts_shiftconv(C.byref(src), P.offs.data_ptr(), u["w1"].data_ptr(), hwb.data_ptr(), st))
                f.keep = (src,)
                calls[("K0_" + name, "CAB2" if mode == 2 else "CAB1")] = f          # (columns: forward unit / reverse unit)
        # K4 (sn_cab1_phase2 / sn_gsts_cab2_phase2): columns CAB1 / CAB2
        cav = torch.rand(T, Cc, device=dev)
        yv = torch.empty_like(g2)
        for name, lib in k0libs.items():
            for mode, unit in ((0, "encoder_level1.1."), (1, "encoder_level1.0.")):
                u = P.units["stage1.decoder_level1." + unit]
                src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 1 if (mode and V.wrap) else 0)
                bo = u["b_out"].data_ptr() if u["b_out"] is not None else None
                fn = lib.sn_gsts_cab2_phase2 if mode else lib.sn_cab1_phase2
                f = (lambda fn=fn, src=src, u=u, bo=bo: fn(C.byref(src), g2.data_ptr(), cav.data_ptr(), u["w_out"].data_ptr(), bo, yv.data_ptr(), st))
                f.keep = (src,)
                calls[("K4_" + name, "CAB2" if mode else "CAB1")] = f
        res = {k: [] for k in calls}
        keys = list(calls)
        for r in range(rounds):
            order = keys if r % 2 == 0 else keys[::-1]
            for k in order:
                f = calls[k]
                assert f() == 0, k
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    f()
                e1.record(); torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / reps * 1e3)
        names = sorted({k[0] for k in keys}, key=lambda n: (n != "r04", n != "base", n))
        print(f"== {model} {T}x{h}x{w}  (us per launch: min / median over {rounds} alternating rounds of {reps})", flush=True)
        for n in names:
            a, b = res[(n, "CAB1")], res[(n, "CAB2")]
            print(f"AB {model} {T}x{h}x{w} {n:16s} CAB1 {min(a):8.1f} / {statistics.median(a):8.1f}   CAB2 {min(b):8.1f} / {statistics.median(b):8.1f}", flush=True)


if __name__ == "__main__":
    main()
