#!/usr/bin/env python3
"""Micro-benchmark of the GSTS unit kernels at one level's size (default: Shift-Net-s level 1 of the 720p T_in=20 window)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="gshift_deblur2")
    ap.add_argument("--t", type=int, default=20)
    ap.add_argument("--h", type=int, default=360)
    ap.add_argument("--w", type=int, default=640)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    dev = torch.device("cuda:0")
    V = VARIANTS[args.variant]
    eng = Engine(Plan(V, synth_state_dict(args.variant), dev))
    x = Act(torch.randn(args.t, args.h, args.w, V.c1, device=dev).to(torch.bfloat16), V.c1)
    pre = "stage1.decoder_level1."
    for _ in range(2):
        y = eng.gsts_unit(pre + "encoder_level1.", x, False)
        y = eng.gsts_unit(pre + "encoder_level1_1.", y, True)
    torch.cuda.synchronize()
    eng.prof = []
    for _ in range(args.iters):
        y = eng.gsts_unit(pre + "encoder_level1.", x, False)
        y = eng.gsts_unit(pre + "encoder_level1_1.", y, True)
    torch.cuda.synchronize()
    agg = {}
    for fn, label, meta, e0, e1 in eng.prof:
        a = agg.setdefault(fn, [0.0, 0])
        a[0] += e0.elapsed_time(e1); a[1] += 1
    px = args.t * args.h * args.w
    tot = 0.0
    for fn, (ms, n) in agg.items():
        print(f"{fn:22s} {1e3 * ms / n:9.1f} us/launch  x{n // args.iters // 2}/unit")
        tot += ms / args.iters / 2
    ub = 4 * px * V.c1 * 2
    print(f"unit total {tot * 1e3:.1f} us ; fused-unit algorithmic bytes {ub / 1e9:.3f} GB -> {ub / tot / 1e6:.0f} GB/s "
          f"({ub / tot / 1e6 / 8000:.3f} of 8 TB/s)")


if __name__ == "__main__":
    main()
