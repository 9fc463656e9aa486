#!/usr/bin/env python3
"""Do the VALU-bound fused phase 1 and the HBM-bound phase 2 (K4) overlap when they run CONCURRENTLY on two streams (independent frame
groups of a window)?  Times, at the level-1 size of config 2 with T = 10 frames per group: phase 1 alone, K4 alone, both back to back on
one stream, both concurrently on two streams -- for the product library and for a build of the phase-1 kernel with ONE workgroup per CU
(tools/p1_variants.py build onewg=-DP1_LDS_PAD=16384), which leaves registers for K4's waves.
usage: overlap_probe.py [path of the one-workgroup phase-1 library]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    name = "gshift_deblur2"
    dev = torch.device("cuda:0")
    eng = Engine(Plan(VARIANTS[name], synth_state_dict(name), dev))
    lib, P = eng.lib, eng.P
    libs = {"product": lib}
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        l1 = C.CDLL(sys.argv[1]); vp, ci = C.c_void_p, C.c_int
        l1.sn_gsts_cab2_phase1.argtypes = [C.POINTER(L.UnitSrc), vp, C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
        l1.sn_cab1_phase1.argtypes = [C.POINTER(L.UnitSrc), C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
        l1.sn_phase1_pool_blocks.argtypes = [ci, ci, ci]
        libs["one workgroup per CU"] = l1
    T, h, w, c = 10, 360, 640, 64
    u1 = P.units["stage1.decoder_level1.encoder_level1.1."]
    mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)           # noqa: E731
    xa, xb, g2a, g2b, ya = mk(T, h, w, c), mk(T, h, w, c), mk(T, h, w, c), mk(T, h, w, c), mk(T, h, w, c)
    ca = torch.ones((T, c), dtype=torch.float32, device=dev)
    srca, srcb = L.UnitSrc(xa.data_ptr(), T, h, w, c, 0, 0), L.UnitSrc(xb.data_ptr(), T, h, w, c, 0, 0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    b_out = u1["b_out"].data_ptr() if u1["b_out"] is not None else None

    def timed(fn, n=8):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for label, pl in libs.items():
        nblk = pl.sn_phase1_pool_blocks(T, h, w)
        pool = torch.zeros((T, nblk, c), dtype=torch.float32, device=dev)
        p1 = lambda st: L.check(L.cab_phase1(pl, srca, None, u1["p1r"]["desc"], g2a.data_ptr(), pool.data_ptr(), st), "p1")          # noqa: E731
        k4 = lambda st: L.check(lib.sn_cab1_phase2(C.byref(srcb), g2b.data_ptr(), ca.data_ptr(), u1["w_out"].data_ptr(), b_out, ya.data_ptr(), st), "k4")   # noqa: E731
        cur = torch.cuda.current_stream().cuda_stream
        t_p1, t_k4 = timed(lambda: p1(cur)), timed(lambda: k4(cur))
        t_seq = timed(lambda: (p1(cur), k4(cur)))

        def both():
            main_s = torch.cuda.current_stream()
            s1.wait_stream(main_s); s2.wait_stream(main_s)
            p1(s1.cuda_stream); k4(s2.cuda_stream)
            main_s.wait_stream(s1); main_s.wait_stream(s2)
        t_con = timed(both)

        def both2():                                        # two K4 launches (two units' worth of phase 2) beside one phase 1
            main_s = torch.cuda.current_stream()
            s1.wait_stream(main_s); s2.wait_stream(main_s)
            p1(s1.cuda_stream); k4(s2.cuda_stream); k4(s2.cuda_stream)
            main_s.wait_stream(s1); main_s.wait_stream(s2)
        t_con2 = timed(both2)
        print(f"{label:22s} (T={T}): phase 1 {t_p1:7.1f} us   K4 {t_k4:7.1f} us   back to back {t_seq:7.1f} us   concurrent {t_con:7.1f} us"
              f"   phase 1 || 2 x K4 {t_con2:7.1f} us (sequential {t_p1 + 2 * t_k4:7.1f})", flush=True)


if __name__ == "__main__":
    main()
