#!/usr/bin/env python3
"""Development aid: A/B timing of compile-time variants of the fused phase-1 kernel (csrc/sn_phase1.hip).  `build` compiles one small
library per variant into shift-net_amd/lib/dev/ (no GPU needed; the .so files travel to the GPU box, they are not committed); `run`
times them at the level-1 size of config 2 and checks that all variants produce the same g2.
usage: p1_variants.py build|run [name=-DFLAG[,-DFLAG...] ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
OUT = os.path.join(ROOT, "shift-net_amd", "lib", "dev")
DEFAULT = ["base=", "d1w4=-DP1_DIST=1,-DP1_WPF=4", "d2w1=-DP1_DIST=2,-DP1_WPF=1", "d1w6=-DP1_DIST=1,-DP1_WPF=6"]


def parse(args):
    out = []
    for a in args or DEFAULT:
        name, _, flags = a.partition("=")
        out.append((name, [f for f in flags.split(",") if f]))
    return out


def lib_path(name):
    return os.path.join(OUT, f"libp1_{name}.so")


def build(variants):
    os.makedirs(OUT, exist_ok=True)
    for name, flags in variants:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", *flags,
               "-Rpass-analysis=kernel-resource-usage", "-o", lib_path(name), os.path.join(ROOT, "shift-net_amd", "csrc", "sn_phase1.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        res = [l.split("remark:")[1].strip() for l in r.stderr.split("\n") if any(k in l for k in (" VGPRs:", "VGPRs Spill", "LDS Size"))]
        print("built", name, flags, " | ".join(res), flush=True)


def run(variants):
    import torch
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    name = "gshift_deblur2"
    dev = torch.device("cuda:0")
    eng = Engine(Plan(VARIANTS[name], synth_state_dict(name), dev))
    st = torch.cuda.current_stream().cuda_stream
    T, h, w, Cc = 20, 360, 640, 64
    g = torch.Generator().manual_seed(1)
    xd = torch.randn(T, h, w, Cc, generator=g).to(torch.bfloat16).to(dev)
    hwb = torch.randn(T, h, w, Cc // 2, generator=g).to(torch.bfloat16).to(dev)
    for mode, unit in ((0, "encoder_level1.1."), (1, "encoder_level1.0.")):
        p1 = eng.P.units["stage1.decoder_level1." + unit]["p1"]
        src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 1 if mode else 0)
        ref = None
        for vname, _ in variants:
            if not os.path.exists(lib_path(vname)):
                continue
            lib = C.CDLL(lib_path(vname))
            vp, ci = C.c_void_p, C.c_int
            lib.sn_gsts_cab2_phase1.argtypes = [C.POINTER(L.UnitSrc), vp, C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
            lib.sn_cab1_phase1.argtypes = [C.POINTER(L.UnitSrc), C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
            lib.sn_phase1_pool_blocks.argtypes = [ci, ci, ci]
            nblk = lib.sn_phase1_pool_blocks(T, h, w)
            pool = torch.zeros((T, nblk, Cc), dtype=torch.float32, device=dev)
            g2 = torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev)
            f = lambda: L.cab_phase1(lib, src, hwb.data_ptr() if mode else None, p1["desc"], g2.data_ptr(), pool.data_ptr(), st)   # noqa: E731
            for _ in range(2):
                assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                f()
            e1.record(); torch.cuda.synchronize()
            same = "" if ref is None else ("  == first variant" if torch.equal(ref, g2) else "  DIFFERS from the first variant")
            ref = g2.clone() if ref is None else ref
            print(f"mode {mode} {vname:10s}: {e0.elapsed_time(e1) / 6 * 1e3:8.1f} us{same}", flush=True)


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)(parse(sys.argv[2:]))
