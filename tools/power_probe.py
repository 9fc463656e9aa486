#!/usr/bin/env python3
"""Clock and package power while the fused phase 1 (sn_cab1_phase1 / sn_gsts_cab2_phase1, level-1 size of config 2) runs back to back on random and on all-zero
activations: rocm-smi is polled from a thread while the launches loop (the launch time follows the operand data).
  python tools/power_probe.py [seconds per case]
  python tools/power_probe.py [seconds] --variants     the measurement builds of tools/p1r_variants.py (lib/dev/libp1r_*.so), random data: which role draws the power"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()}
    except Exception as e:          # noqa: BLE001
        return {"error": repr(e)}


def main():
    import torch
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Plan
    from shiftnet_amd.spec import VARIANTS as SV
    from shiftnet_amd.weights import synth_state_dict
    secs = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1][0] != "-" else 5.0
    variants = "--variants" in sys.argv
    DEV = os.path.join(ROOT, "shift-net_amd", "lib", "dev")
    dev = torch.device("cuda:0")
    lib = L.load()
    libs = {"shipped": lib}
    if variants:
        libs = {}
        vp, ci = C.c_void_p, C.c_int
        for f in sorted(os.listdir(DEV)):
            if f.startswith("libp1r_") and f.endswith(".so"):
                v = C.CDLL(os.path.join(DEV, f))
                v.sn_phase1_pool_blocks.argtypes = [ci, ci, ci]
                v.sn_gsts_cab2_phase1.argtypes = [C.POINTER(L.UnitSrc), vp, C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
                libs[f[7:-3]] = v
    st = torch.cuda.current_stream().cuda_stream
    print("idle", smi(), flush=True)
    for model, (T, h, w) in (("gshift_deblur2", (20, 360, 640)), ("gshift_deblur1", (52, 360, 640))):
        V = SV[model]
        P = Plan(V, synth_state_dict(model), dev)
        Cc = V.c1
        g2 = torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev)
        nb = lib.sn_phase1_pool_blocks(T, h, w)
        pool = torch.zeros((T, nb, Cc), dtype=torch.float32, device=dev)
        u = P.units["stage1.decoder_level1.encoder_level1.0."]["p1r"]
        for rep, (vname, lib) in [(r, ("shipped", lib)) for r in range(2)] if not variants else list(enumerate(libs.items())):
            for kind in (("randn", "zero") if not variants else ("randn",)):
                xd = (torch.randn(T, h, w, Cc, device=dev) if kind == "randn" else torch.zeros(T, h, w, Cc, device=dev)).to(torch.bfloat16)
                hwb = (torch.randn(T, h, w, Cc // 2, device=dev) if kind == "randn" else torch.zeros(T, h, w, Cc // 2, device=dev)).to(torch.bfloat16)
                src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, 1, 0)
                f = lambda: lib.sn_gsts_cab2_phase1(C.byref(src), hwb.data_ptr(), C.byref(u["desc"]), g2.data_ptr(), pool.data_ptr(), None, None, st)
                assert f() == 0
                torch.cuda.synchronize()
                samples, stop = [], threading.Event()
                th = threading.Thread(target=lambda: [samples.append(smi()) or time.sleep(0.2) for _ in iter(lambda: stop.is_set(), True)])
                th.start()
                n, t0 = 0, time.time()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                while time.time() - t0 < secs:
                    for _ in range(50):
                        f()
                    n += 50
                    torch.cuda.synchronize()
                e1.record(); torch.cuda.synchronize()
                stop.set(); th.join()
                us = e0.elapsed_time(e1) / n * 1e3
                print(f"PWR {model} {T}x{h}x{w} CAB2 {vname:14s} {kind:5s} {us:8.1f} us/launch over {n} launches; rocm-smi samples (first, middle, last of {len(samples)}):", flush=True)
                for smp in (samples[:1] + samples[len(samples) // 2:len(samples) // 2 + 1] + samples[-1:]):
                    print("    ", smp, flush=True)


if __name__ == "__main__":
    main()
