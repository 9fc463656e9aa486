#!/usr/bin/env python3
"""A/B of sn_ln_gemm_gate (K12) between the production library and the experimental one built with another compile-time shape
(e.g. SN_HIPCC_FLAGS=-DSN_K12_WSTREAM=1): bitwise comparison of g1 on a ragged size, then timing at the level-1 size of config 3."""
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
    name = sys.argv[1] if len(sys.argv) > 1 else "gshift_deblur1"
    dev = torch.device("cuda:0")
    V = VARIANTS[name]
    eng = Engine(Plan(V, {k: v.bfloat16() for k, v in synth_state_dict(name).items()}, dev))
    P = eng.P
    d = os.path.join(ROOT, "shift-net_amd", "lib")
    libs = {"prod": eng.lib, "exp": C.CDLL(os.path.join(d, "libshiftnet_hip_exp.so"))}
    libs["exp"].sn_ln_gemm_gate.argtypes = eng.lib.sn_ln_gemm_gate.argtypes
    c = V.c1
    u = P.units["stage1.decoder_level1.encoder_level1.0."]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(2)

    def run(lib, src, hw, g1, mode):
        rc = lib.sn_ln_gemm_gate(C.byref(src), hw.data_ptr() if mode else None, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(), u["w_dw3_h2"].data_ptr(),
                                 g1.data_ptr(), None, 0, st)
        assert rc == 0, rc

    bad = 0
    for (T, h, w) in ((3, 45, 77), (2, 8, 32), (4, 90, 160)):
        x = Act(torch.randn(T, h, w, c, generator=g).to(torch.bfloat16).to(dev), c)
        hw = torch.randn(T, h, w, c // 2, generator=g).to(torch.bfloat16).to(dev)
        for mode in (1, 0, 2):
            src = eng._unit_src(x, mode)
            outs = {}
            for k, lib in libs.items():
                g1 = torch.full((T, h, w, c), -3.0, dtype=torch.bfloat16, device=dev)
                run(lib, src, hw, g1, mode)
                torch.cuda.synchronize()
                outs[k] = g1
            same = torch.equal(outs["prod"], outs["exp"])
            bad += 0 if same else 1
            print(f"{'ok' if same else 'MISMATCH'} T={T} h={h} w={w} mode={mode}" +
                  ("" if same else f" max diff {(outs['prod'].float() - outs['exp'].float()).abs().max().item():.4g}"), flush=True)
    T, h, w = 20, 360, 640
    x = Act(torch.randn(T, h, w, c, device=dev).to(torch.bfloat16), c)
    hw = torch.randn(T, h, w, c // 2, device=dev).to(torch.bfloat16)
    g1 = torch.empty((T, h, w, c), dtype=torch.bfloat16, device=dev)
    for mode in (1, 0):
        src = eng._unit_src(x, mode)
        for rep in range(2):
            for k, lib in libs.items():
                for _ in range(2):
                    run(lib, src, hw, g1, mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(lib, src, hw, g1, mode)
                e1.record(); torch.cuda.synchronize()
                print(f"{k} mode {mode} rep {rep}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
    print("mismatching cases:", bad)


if __name__ == "__main__":
    main()
