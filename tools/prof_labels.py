#!/usr/bin/env python3
"""Per-launch-label time of one forward (stream events around every C-ABI call): which convs / operators of a variant cost what.
usage: prof_labels.py VARIANT DTYPE T H W [top]     e.g. prof_labels.py gshift_denoise1 fp32 36 272 448"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import importlib
    from shiftnet_amd.weights import synth_state_dict
    name, dts, T, H, W = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    top = int(sys.argv[6]) if len(sys.argv) > 6 else 25
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dts]
    net = importlib.import_module(f"basicsr.models.archs.{name}").GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name), strict=True)
    net = net.to(dt).cuda().eval()
    x = torch.rand(1, T, 3, H, W, device="cuda").to(dt)
    nm = torch.full((1, T, 1, H, W), 30.0 / 255.0, dtype=dt, device="cuda") if "denoise" in name else None
    run = (lambda: net(x, nm)) if nm is not None else (lambda: net(x))
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        eng = net.prepare()
        eng.prof = []
        run(); torch.cuda.synchronize()
    agg = {}
    for fn, label, meta, e0, e1 in eng.prof:
        key = re.sub(r"\b(orb|rorb)\d\.", r"\1N.", label)
        key = re.sub(r"encoder_level1(_\d)?\.(\d)\.", "unit.\\2.", key) if "stage1." in key and "level" in key else key
        if meta and meta[0] in ("conv", "conv32"):
            key += f" {meta[2]}x{meta[3]} cin{meta[4]} cout{meta[5]} k{meta[6]} s{meta[7]}"
        a = agg.setdefault(key, [0.0, 0]); a[0] += e0.elapsed_time(e1); a[1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"{name} {dts} T={T} {H}x{W}: {tot:.1f} ms in {sum(v[1] for v in agg.values())} launches")
    for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"  {ms:9.2f} ms {100 * ms / tot:5.1f} %  x{n:4d}  {ms / n * 1e3:9.1f} us  {k}")


if __name__ == "__main__":
    main()
