#!/usr/bin/env python3
"""In-kernel phase profile of sn_ln_gemm_gate (s_memtime accumulators of the first 256 workgroups of frame 0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
os.environ["SN_EXPERIMENTAL"] = "1"          # the phase clocks exist only in the -DSN_EXPERIMENTAL library
import torch  # noqa: E402


def main():
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Act, Plan
    from shiftnet_amd.engine_experimental import ExperimentalEngine as Engine
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    dev = torch.device("cuda:0")
    V = VARIANTS["gshift_deblur2"]
    eng = Engine(Plan(V, synth_state_dict("gshift_deblur2"), dev))
    eng.gsts_v = int(sys.argv[1]) if len(sys.argv) > 1 else eng.gsts_v
    x = Act(torch.randn(20, 360, 640, 64, device=dev).to(torch.bfloat16), 64)
    pre = "stage1.decoder_level1.encoder_level1.1."          # CAB1 (mode 0)
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    eng.naf(pre, x, 0); torch.cuda.synchronize()
    eng.lib.sn_debug_buf_set.argtypes = [L.C.c_void_p]
    eng.lib.sn_debug_buf_set(buf.data_ptr())
    eng.lib.sn_debug_set(256)
    eng.naf(pre, x, 0); torch.cuda.synchronize()
    eng.lib.sn_debug_buf_set(None)
    eng.lib.sn_debug_set(0)
    a = buf.view(256, 8, 8).double()
    names = ["LN (loads+VALU)", "gemm_chunk", "barrier", "stencil", "epilogue"]
    tot = a.sum(-1).mean().item()
    print(f"mean wave cycles per workgroup-tile: {tot:.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:18s} {a[:, :, k].mean().item():9.0f}  {100 * a[:, :, k].mean().item() / tot:5.1f} %")


if __name__ == "__main__":
    main()
