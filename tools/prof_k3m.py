#!/usr/bin/env python3
"""In-kernel phase profile of sn_dw5m_gemm_gate (s_memtime accumulators per wave, see sn_debug_buf_set)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
os.environ["SN_EXPERIMENTAL"] = "1"          # the phase clocks exist only in the -DSN_EXPERIMENTAL library
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=20)
    ap.add_argument("--h", type=int, default=360)
    ap.add_argument("--w", type=int, default=640)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--kernel", default="k3m")
    args = ap.parse_args()
    from shiftnet_amd import lib as L, prep
    lib = L.load()
    dev = torch.device("cuda:0")
    T, h, w, C = args.t, args.h, args.w, 64
    wr = lib.sn_planar_pitch(w)
    g1p = torch.randn(T, h, C, wr, device=dev).to(torch.bfloat16)
    w5 = torch.randn(25, C) * 0.2
    ttab = prep.pack_toeplitz(w5, 5).to(dev)
    wg = prep.pack_gate_gemm(torch.randn(2 * C, C, 1, 1) * 0.1, C).to(dev)
    g2 = torch.empty(T, h, w, C, dtype=torch.bfloat16, device=dev)
    pool = torch.empty(T, lib.sn_dw5m_blocks(h, w), C, dtype=torch.float32, device=dev)
    buf = torch.zeros(512 * 8 * 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.sn_debug_set(args.dbg)

    def run():
        L.check(lib.sn_dw5m_gemm_gate(g1p.data_ptr(), None, ttab.data_ptr(), wg.data_ptr(), g2.data_ptr(), pool.data_ptr(), T, h, w, C, st), "k3m")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"plain: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us")
    lib.sn_debug_buf_set.argtypes = [L.C.c_void_p]
    lib.sn_debug_buf_set(buf.data_ptr())
    lib.sn_debug_set(args.dbg | 512)
    run(); torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.sn_debug_buf_set(None)
    lib.sn_debug_set(0)
    print(f"instrumented: {e0.elapsed_time(e1) * 1e3:.1f} us")
    a = buf.view(512, 8, 8).double()
    a = a[a.sum((1, 2)) > 0]                      # workgroups that ran (256 or 512, depending on the kernel shape)
    names = ["stage(write+issue)", "barrier_after_gemm", "-", "toeplitz", "-", "barrier_step", "gemm", "tile_top"]
    tot = a.sum(-1).mean().item()
    print(f"mean wave cycles (100 MHz ticks?) total {tot:.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:12s} {a[:, :, k].mean().item():10.0f}  {100 * a[:, :, k].mean().item() / tot:5.1f} %   (wave0 {a[:, 0, k].mean().item():.0f}, wave7 {a[:, 7, k].mean().item():.0f})")


if __name__ == "__main__":
    main()
