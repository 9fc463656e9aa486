#!/usr/bin/env python3
"""A/B of sn_dw5m_gemm_gate between the production library and the experimental one (built with another kernel shape, e.g.
SN_HIPCC_FLAGS=-DSN_K3M_WAVE=1): bitwise comparison of g2 / pool on ragged and full sizes, then timing of both at the level-1 / level-2
sizes of BASELINE config 2."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    lib.sn_dw5m_gemm_gate.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.sn_dw5m_blocks.argtypes = [ci, ci]
    lib.sn_planar_pitch.argtypes = [ci]
    return lib


def main():
    from shiftnet_amd import lib as L, prep
    L.load()                                   # maps torch's HIP runtime first
    d = os.path.join(ROOT, "shift-net_amd", "lib")
    libs = {"prod": bind(os.path.join(d, "libshiftnet_hip.so")), "exp": bind(os.path.join(d, "libshiftnet_hip_exp.so"))}
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    w5 = torch.randn(25, 64, generator=g) * 0.2
    ttab = prep.pack_toeplitz(w5, 5).to(dev)
    wg = prep.pack_gate_gemm(torch.randn(128, 64, 1, 1, generator=g) * 0.1, 64).to(dev)

    def run(lib, g1p, ca, T, h, w):
        g2 = torch.full((T, h, w, 64), -7.0, dtype=torch.bfloat16, device=dev)
        pool = torch.full((T, lib.sn_dw5m_blocks(h, w), 64), -7.0, dtype=torch.float32, device=dev)
        rc = lib.sn_dw5m_gemm_gate(g1p.data_ptr(), ca.data_ptr() if ca is not None else None, ttab.data_ptr(), wg.data_ptr(), g2.data_ptr(),
                                   pool.data_ptr(), T, h, w, 64, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return g2, pool

    bad = 0
    for (T, h, w, with_ca) in ((1, 8, 64, False), (2, 17, 28, True), (3, 45, 80, False), (5, 90, 160, True), (2, 68, 112, True), (20, 180, 320, False),
                               (4, 360, 640, True), (1, 1, 1, False), (2, 9, 65, True)):
        wr = libs["prod"].sn_planar_pitch(w)
        g1p = torch.randn(T, h, 64, wr, generator=g).to(torch.bfloat16)
        g1p[..., w:] = 0
        g1p = g1p.to(dev)
        ca = (torch.rand(T, 64, generator=g) + 0.5).to(dev) if with_ca else None
        a, pa = run(libs["prod"], g1p, ca, T, h, w)
        for rep in range(3):
            b, pb = run(libs["exp"], g1p, ca, T, h, w)
            ok = torch.equal(a, b) and torch.equal(pa, pb)
            if not ok:
                bad += 1
                dd = (a.float() - b.float()).abs()
                print(f"MISMATCH T={T} h={h} w={w} ca={with_ca} rep={rep}: g2 {int((dd > 0).sum())} differ (max {dd.max().item():.4g}), "
                      f"pool max {(pa - pb).abs().max().item():.4g}; first bad index {torch.nonzero(dd > 0)[:3].tolist()}", flush=True)
                break
        else:
            print(f"ok T={T} h={h} w={w} ca={with_ca}", flush=True)
    for (T, h, w) in ((20, 360, 640), (20, 180, 320)):
        wr = libs["prod"].sn_planar_pitch(w)
        g1p = torch.randn(T, h, 64, wr, device=dev).to(torch.bfloat16)
        for name, lib in libs.items():
            for _ in range(3):
                run(lib, g1p, None, T, h, w)
            g2 = torch.empty((T, h, w, 64), dtype=torch.bfloat16, device=dev)
            pool = torch.empty((T, lib.sn_dw5m_blocks(h, w), 64), dtype=torch.float32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.sn_dw5m_gemm_gate(g1p.data_ptr(), None, ttab.data_ptr(), wg.data_ptr(), g2.data_ptr(), pool.data_ptr(), T, h, w, 64, st)
            e1.record(); torch.cuda.synchronize()
            print(f"{name}: {T}x{h}x{w}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    print("mismatching cases:", bad)


if __name__ == "__main__":
    main()
