#!/usr/bin/env python3
"""Does a dense CAB (conv -> closed-form CALayer -> conv) run faster per frame when its working set stays in the 256 MB Infinity Cache?
Per-frame time of Engine.cab at T = 1, 2, 4 frames (x + mid + out resident) against T = 20 (streaming), for the 16-channel full-resolution CABs
and the 24-channel half-resolution ones of config 2.  If the small-T figure were clearly lower, issuing the CABs per frame group would pay
(every operator of a CAB is per frame).  usage: cab_mall_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    dev = torch.device("cuda:0")
    V = VARIANTS["gshift_deblur2"]
    eng = Engine(Plan(V, synth_state_dict("gshift_deblur2"), dev))
    cases = (("orb1.encoder_level1.0.", 14, 16, 720, 1280), ("orb1.encoder_level2.0.", 18, 24, 360, 640))
    for pre, c, cs, h, w in cases:
        for rnd in range(2):
            for T in (20, 1, 2, 4, 8, 20):
                x = Act(torch.randn(T, h, w, cs, device=dev).to(torch.bfloat16), c)
                x.t[..., c:] = 0
                n = max(2, 40 // T)
                for _ in range(2):
                    eng.cab(pre, x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    eng.cab(pre, x)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                mb = T * h * w * cs * 2 / 1e6
                print(f"CABPROBE {pre} {cs}ch {h}x{w} T={T:2d}: {us:8.1f} us per CAB, {us / T:7.1f} us per frame; tensor {mb:6.1f} MB (x + mid + out = {3 * mb:6.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
