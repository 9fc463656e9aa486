#!/usr/bin/env python3
"""In-kernel phase profile of sn_grp5_gemm_gate (s_memtime accumulators per wave; experimental library)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
os.environ["SN_EXPERIMENTAL"] = "1"
import torch  # noqa: E402


def main():
    from shiftnet_amd import lib as L, prep
    lib = L.load()
    dev = torch.device("cuda:0")
    T, h, w, C = 20, 360, 640, 80
    g1 = torch.randn(T, h, w, C, device=dev).to(torch.bfloat16)
    wg = prep.pack_grouped_frag(torch.randn(C, 8, 5, 5) * 0.05, torch.randn(C, 8, 3, 3) * 0.05).to(dev)
    wf = prep.pack_gate_gemm(torch.randn(2 * C, C, 1, 1) * 0.1, C).to(dev)
    g2 = torch.empty(T, h, w, C, dtype=torch.bfloat16, device=dev)
    pool = torch.empty(T, lib.sn_grp5_blocks(h, w), C, dtype=torch.float32, device=dev)
    buf = torch.zeros(256 * 16 * 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        L.check(lib.sn_grp5_gemm_gate(g1.data_ptr(), None, wg.data_ptr(), wf.data_ptr(), g2.data_ptr(), pool.data_ptr(), T, h, w, C, st), "k3g")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"plain: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us for T={T} {h}x{w} C={C} ({2 * g1.numel() * 2 / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e9:.0f} GB/s algorithmic)")
    lib.sn_debug_buf_set(buf.data_ptr())
    lib.sn_debug_set(1024)
    run(); torch.cuda.synchronize()
    lib.sn_debug_buf_set(None); lib.sn_debug_set(0)
    a = buf.view(256, 16, 8)[:, :10].double()
    tiles = T * ((h + 3) // 4) * ((w + 31) // 32) / 256
    names = ["lds_write(+wait loads)", "barrier1", "issue+grouped mfma", "barrier2", "1x1+gate", "barrier3", "store", "loop top"]
    tot = a.sum(-1).mean().item()
    print(f"mean wave ticks per launch {tot:.0f} = {tot / tiles:.0f} per tile ({tiles:.0f} tiles per workgroup)")
    for k, nm in enumerate(names):
        print(f"  {nm:24s} {a[:, :, k].mean().item() / tiles:9.0f} / tile  {100 * a[:, :, k].mean().item() / tot:5.1f} %")


if __name__ == "__main__":
    main()
