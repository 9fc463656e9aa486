#!/usr/bin/env python3
"""Every sn_conv2d launch of one window by label: duration (stream events), its own algorithmic bytes, TB/s -- which convs are NOT on the streaming
kernel and what they cost.   usage: conv_labels.py [--config 2|3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import importlib  # noqa: E402

import torch  # noqa: E402


def main():
    cfg = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 2
    variant, T = ("gshift_deblur2", 20) if cfg == 2 else ("gshift_deblur1", 52)
    from shiftnet_amd.weights import synth_state_dict
    sys.path.insert(0, ROOT)
    import bench
    net = importlib.import_module(f"basicsr.models.archs.{variant}").GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(variant), strict=True)
    net = net.to(torch.bfloat16).cuda().eval()
    x = torch.rand((1, T, 3, 720, 1280), device="cuda").to(torch.bfloat16)
    eng = net.prepare()
    with torch.no_grad():
        net(x); net(x)
        eng.prof = []
        net(x)
    torch.cuda.synchronize()
    rows = {}
    for fn, label, meta, e0, e1 in eng.prof:
        if fn != "sn_conv2d":
            continue
        ms = e0.elapsed_time(e1)
        b = bench.kernel_alg_bytes(fn, meta)
        k = (label.split("[")[1].rstrip("]"), tuple(meta[1:]))
        r = rows.setdefault(k, [0.0, 0, 0.0])
        r[0] += ms; r[1] += 1; r[2] += b
    tot = 0.0
    print("label, (T, h_out, w_out, cin_total, cs_out, k, stride, in_mode, out_mode): launches, ms total, us each, TB/s of own bytes")
    for (name, meta), (ms, n, b) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
        single3 = meta[5] == 3 and meta[6] == 1 and meta[7] == 0 and meta[8] == 0
        tot += ms
        print(f"{name:42s} {str(meta):52s} n={n:3d} {ms:7.3f} ms {ms / n * 1e3:8.1f} us {b / ms / 1e9:6.2f} TB/s {'' if single3 else '   <- not a plain 3x3'}")
    print("sum", round(tot, 2), "ms")


if __name__ == "__main__":
    main()
