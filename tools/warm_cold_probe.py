#!/usr/bin/env python3
"""How much do the GSTS kernels gain when their input is resident in the 256 MB Infinity Cache?  Each kernel of a level-1 unit
(360x640, C = 64) is timed in a loop on the SAME buffers with T = 3 (working set ~180 MB: warm after the first iteration) and with
T = 20 (1.2 GB: every launch streams from HBM); the per-frame times bound what frame-wise scheduling could win.
usage: warm_cold_probe.py [variant]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    name = sys.argv[1] if len(sys.argv) > 1 else "gshift_deblur2"
    dev = torch.device("cuda:0")
    V = VARIANTS[name]
    eng = Engine(Plan(V, {k: v.bfloat16() for k, v in synth_state_dict(name).items()}, dev))
    lib, P = eng.lib, eng.P
    c, h, w = V.c1, 360, 640
    pre = "stage1.decoder_level1.encoder_level1.0."
    u = P.units[pre]
    st = torch.cuda.current_stream().cuda_stream
    mst = not V.grouped_rep
    res = {}
    for T in (3, 20):
        x = Act(torch.randn(T, h, w, c, device=dev).to(torch.bfloat16), c)
        src = eng._unit_src(x, 1)
        hwb = eng._new(T, h, w, c // 2)
        g1 = torch.empty((T, h, c, lib.sn_planar_pitch(w)), dtype=torch.bfloat16, device=dev) if mst else eng._new(T, h, w, c)
        g2 = eng._new(T, h, w, c)
        y = eng._new(T, h, w, c)
        nb = lib.sn_dw5m_blocks(h, w) if mst else lib.sn_grp5_blocks(h, w)
        pool2 = torch.empty((T, nb, c), dtype=torch.float32, device=dev)
        ca2 = torch.ones((T, c + 16), dtype=torch.float32, device=dev)
        b_out = u["b_out"].data_ptr() if u["b_out"] is not None else None
        fused = "p1r" in u                 # Shift-Net-s deblur: phase 1 is one kernel (sn_gsts_cab2_phase1)
        pool1 = torch.empty((T, max(lib.sn_phase1_pool_blocks(T, h, w), 1), c), dtype=torch.float32, device=dev)
        calls = {
            "K0": lambda: lib.sn_gsts_shiftconv(C.byref(src), P.offs.data_ptr(), u["w1"].data_ptr(), hwb.data_ptr(), st),
            "P1": lambda: L.cab_phase1(lib, src, hwb.data_ptr(), u["p1r"]["desc"], g2.data_ptr(), pool1.data_ptr(), st),
            "K12": lambda: lib.sn_ln_gemm_gate(C.byref(src), hwb.data_ptr(), u["w_ln"].data_ptr(), u["b_ln"].data_ptr(), u["w_dw3_h2"].data_ptr(),
                                               g1.data_ptr(), None, 2 if mst else 0, st),
            "K3": (lambda: lib.sn_dw5m_gemm_gate(g1.data_ptr(), None, u["w_toep5"].data_ptr(), u["w_gate"].data_ptr(), g2.data_ptr(), pool2.data_ptr(), T, h, w, c, st))
            if mst else (lambda: lib.sn_grp5_gemm_gate(g1.data_ptr(), None, u["w_grp"].data_ptr(), u["w_gate"].data_ptr(), g2.data_ptr(), pool2.data_ptr(), T, h, w, c, st)),
            "K4": lambda: (lib.sn_gsts_cab2_phase2 if src.mode else lib.sn_cab1_phase2)(C.byref(src), g2.data_ptr(), ca2.data_ptr(), u["w_out"].data_ptr(), b_out, y.data_ptr(), st),
        }
        for k in (("K12", "K3") if fused else ("P1",)):
            del calls[k]
        for k, f in calls.items():
            for _ in range(3):
                L.check(f(), k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30 if T == 3 else 10
            e0.record()
            for _ in range(n):
                f()
            e1.record(); torch.cuda.synchronize()
            res[(k, T)] = e0.elapsed_time(e1) / n * 1e3 / T
        # the chain K0 -> (P1 | K12 -> K3) -> K4 back to back (producer/consumer through the cache when T = 3)
        for _ in range(2):
            for f in calls.values():
                f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20 if T == 3 else 6
        e0.record()
        for _ in range(n):
            for f in calls.values():
                f()
        e1.record(); torch.cuda.synchronize()
        res[("chain", T)] = e0.elapsed_time(e1) / n * 1e3 / T
    print(f"{name}: us per FRAME at 360x640 (T=3: Infinity-Cache resident, T=20: streaming)")
    for k in ("K0", "P1", "K12", "K3", "K4", "chain"):
        if (k, 3) not in res:
            continue
        a, b = res[(k, 3)], res[(k, 20)]
        print(f"  {k:6s} warm {a:7.2f}   cold {b:7.2f}   warm/cold {a / b:5.2f}")


if __name__ == "__main__":
    main()
