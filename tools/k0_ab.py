#!/usr/bin/env python3
"""K0 (CAB2.conv1 of the displaced neighbour-frame half) on the matrix cores against the VALU kernel: every variant timed once per round, rounds in
alternating order, min / median (us per launch), at the level-1 sizes of configs 2, 3 and 6.   usage: k0_ab.py [--rounds 6] [--reps 5]"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict

    def arg(k, d):
        return int(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
    rounds, reps = arg("--rounds", 6), arg("--reps", 5)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for name, (T, h, w) in (("gshift_deblur2", (20, 360, 640)), ("gshift_deblur2", (20, 180, 320)), ("gshift_deblur1", (52, 360, 640)), ("gshift_deblur1", (20, 540, 960))):
        P = Plan(VARIANTS[name], synth_state_dict(name), dev)
        eng = Engine(P)
        Cc = VARIANTS[name].c1
        x = torch.randn((T, h, w, Cc), device=dev).to(torch.bfloat16)
        src = eng._unit_src(Act(x, Cc), 1)
        pre = "stage1.decoder_level1.encoder_level1.0."
        out = {v: torch.empty((T, h, w, Cc // 2), dtype=torch.bfloat16, device=dev) for v in ("valu", "mfma")}
        DEV = os.path.join(ROOT, "shift-net_amd", "lib", "dev")
        vlibs = {}
        if os.path.isdir(DEV):                          # measurement builds: build.build(out=lib/dev/libshiftnet_k0m<N>.so, flags=["-DK0M_SKIP=<N>"])
            for f in sorted(os.listdir(DEV)):
                if f.startswith("libshiftnet_k0m") and f.endswith(".so"):
                    vl = C.CDLL(os.path.join(DEV, f))
                    vl.sn_gsts_shiftconv_mfma.argtypes = [C.POINTER(L.UnitSrc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
                    vlibs[f[12:-3]] = vl
                    out[f[12:-3]] = torch.empty_like(out["mfma"])

        def launch(v):
            if v == "valu":
                L.check(eng.lib.sn_gsts_shiftconv(C.byref(src), P.offs.data_ptr(), P.units[pre]["w1"].data_ptr(), out[v].data_ptr(), st), v)
            elif v in vlibs:
                L.check(vlibs[v].sn_gsts_shiftconv_mfma(C.byref(src), P.offs.data_ptr(), P.units[pre]["w1"].data_ptr(), out[v].data_ptr(), st), v)
            else:
                L.check(eng.lib.sn_gsts_shiftconv_mfma(C.byref(src), P.offs.data_ptr(), P.units[pre]["w1"].data_ptr(), out[v].data_ptr(), st), v)
        times = {v: [] for v in out}
        for v in out:
            launch(v)
        torch.cuda.synchronize()
        for r in range(rounds):
            for v in (list(out) if r % 2 == 0 else list(out)[::-1]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                launch(v)
                e0.record()
                for _ in range(reps):
                    launch(v)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / reps * 1e3)
        gb = 2 * T * h * w * (Cc // 2) * 2 / 1e9
        dmax = (out["valu"].float() - out["mfma"].float()).abs().max().item()
        for v in out:
            mn, md = min(times[v]), statistics.median(times[v])
            print(f"K0 {name} C={Cc} {T}x{h}x{w} {v}: min {mn:8.1f} us  median {md:8.1f} us = {gb / (md * 1e-6) / 1e3:5.2f} TB/s of its {gb:.2f} GB   (max |valu - mfma| {dmax:.4f})", flush=True)


if __name__ == "__main__":
    main()
