#!/usr/bin/env python3
"""A/B builds of the role-split fused phase-1 kernel (csrc/sn_phase1r.hip): each variant is a mini library (sn_phase1r.hip with
-D flags) timed on the level-1 sizes of configs 3 / 2.  Variants with P1R_SKIP switch off parts of the roles (wrong results) to show which role paces a step.
  build (CPU, no GPU needed):  python tools/p1r_variants.py build
  time (GPU box):              python tools/p1r_variants.py time [name ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
DEV = os.path.join(ROOT, "shift-net_amd", "lib", "dev")
VARIANTS = {
    "base": [],
    "S_only": ["-DP1R_SKIP=56"], "A_only": ["-DP1R_SKIP=55"], "B_only": ["-DP1R_SKIP=15"], "B_rep_only": ["-DP1R_SKIP=31"], "B_1x1_only": ["-DP1R_SKIP=47"],
    "no_Smath": ["-DP1R_SKIP=1"], "no_loads": ["-DP1R_SKIP=2"], "no_stores": ["-DP1R_SKIP=4"], "no_A": ["-DP1R_SKIP=8"], "no_B": ["-DP1R_SKIP=48"],
    "barriers_only": ["-DP1R_SKIP=63"],
    "noAV": ["-DP1R_AV=0"], "noBV": ["-DP1R_BV=0"], "noAVBV": ["-DP1R_AV=0", "-DP1R_BV=0"], "d86": ["-DP1R_DB=8", "-DP1R_DA=6"], "d22": ["-DP1R_DB=2", "-DP1R_DA=2"],
    "AV6": ["-DP1R_AV=6"], "BV2": ["-DP1R_BV=2"],
    "da2": ["-DP1R_DA=2"], "nsw4": ["-DP1R_NSW64=4"], "no_mem": ["-DP1R_SKIP=6"], "no_sigmoid": ["-DP1R_SKIP=64"], "nsw2": ["-DP1R_NSW64=2"],
    "prio000": ["-DP1R_PRIO_S=0", "-DP1R_PRIO_B=0", "-DP1R_PRIO_A=0"], "prio300": ["-DP1R_PRIO_S=3", "-DP1R_PRIO_B=0", "-DP1R_PRIO_A=0"],
    "prio311": ["-DP1R_PRIO_S=3", "-DP1R_PRIO_B=1", "-DP1R_PRIO_A=1"], "prio312": ["-DP1R_PRIO_S=3", "-DP1R_PRIO_B=1", "-DP1R_PRIO_A=2"],
    # LDS accesses with bank conflicts moved to conflict-free addresses (wrong results): what a re-pitch of each ring could buy at most
    "nc1": ["-DP1R_NOCONF=1"], "nc2": ["-DP1R_NOCONF=2"], "nc4": ["-DP1R_NOCONF=4"], "nc8": ["-DP1R_NOCONF=8"], "nc16": ["-DP1R_NOCONF=16"], "nc31": ["-DP1R_NOCONF=31"],
    "nc7": ["-DP1R_NOCONF=7"], "nc23": ["-DP1R_NOCONF=23"],
    "pair0": ["-DP1R_PAIR=0"], "pair1": ["-DP1R_PAIR=1"], "pair3": ["-DP1R_PAIR=3"], "pair5": ["-DP1R_PAIR=5"], "pair6": ["-DP1R_PAIR=6"], "opad0": ["-DP1R_OPAD=0"], "oalign0": ["-DP1R_OALIGN=0"],
    "pair1opad0": ["-DP1R_PAIR=1", "-DP1R_OPAD=0"],
    "prio210": ["-DP1R_PRIO_S=2", "-DP1R_PRIO_B=1", "-DP1R_PRIO_A=0"], "prio231": ["-DP1R_PRIO_S=2", "-DP1R_PRIO_B=3", "-DP1R_PRIO_A=1"],
}
if os.environ.get("P1R_EXTRA"):       # "name:-Dflag,-Dflag;name2:..."
    for item in os.environ["P1R_EXTRA"].split(";"):
        n, f = item.split(":")
        VARIANTS[n] = f.split(",")


def build(only=None):
    os.makedirs(DEV, exist_ok=True)
    csrc = os.path.join(ROOT, "shift-net_amd", "csrc")
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-function"]
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        out = os.path.join(DEV, f"libp1r_{name}.so")
        src = os.path.join(csrc, "sn_phase1r.hip")
        if any("P1R_NOCONF" in f for f in flags):
            # the conflict-free (wrong-result) addresses are not part of the shipped source: tools/p1r_noconf.patch on a copy beside it
            src = os.path.join(csrc, "_p1r_noconf.hip")
            subprocess.run(["patch", "-s", "-o", src, os.path.join(csrc, "sn_phase1r.hip"), os.path.join(ROOT, "tools", "p1r_noconf.patch")], check=True)
        try:
            subprocess.run(base + ["-shared", "-o", out, *flags, src], check=True)
        finally:
            if src.endswith("_p1r_noconf.hip"):
                os.remove(src)
        print("built", out, flush=True)


def time_all(names):
    import torch
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Plan
    from shiftnet_amd.spec import VARIANTS as SV
    from shiftnet_amd.weights import synth_state_dict
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for model, sizes in (("gshift_deblur1", ((52, 360, 640),)), ("gshift_deblur2", ((20, 360, 640),))):
        V = SV[model]
        P = Plan(V, synth_state_dict(model), dev)
        Cc = V.c1
        for (T, h, w) in sizes:
            xd = torch.randn(T, h, w, Cc, device=dev).to(torch.bfloat16)
            hwb = torch.randn(T, h, w, Cc // 2, device=dev).to(torch.bfloat16)
            if os.environ.get("P1R_DATA") == "zero":      # launch time against operand data: all-zero activations (power-limited clocks, profiles/r06_p1r_lds_*)
                xd.zero_(); hwb.zero_()
            g2 = torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev)
            for name in names:
                path = os.path.join(DEV, f"libp1r_{name}.so")
                if not os.path.exists(path):
                    continue
                lib = C.CDLL(path)
                vp, ci = C.c_void_p, C.c_int
                lib.sn_phase1_pool_blocks.argtypes = [ci, ci, ci]
                lib.sn_gsts_cab2_phase1.argtypes = [C.POINTER(L.UnitSrc), vp, C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
                lib.sn_cab1_phase1.argtypes = [C.POINTER(L.UnitSrc), C.POINTER(L.Phase1Weights), vp, vp, C.POINTER(L.SeFold), C.POINTER(L.Phase1Opts), vp]
                nb = lib.sn_phase1_pool_blocks(T, h, w)
                pool = torch.zeros((T, nb, Cc), dtype=torch.float32, device=dev)
                res = []
                for mode, unit in ((0, "encoder_level1.1."), (1, "encoder_level1.0.")):
                    u = P.units["stage1.decoder_level1." + unit]["p1r"]
                    src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 0)
                    f = (lambda: lib.sn_gsts_cab2_phase1(C.byref(src), hwb.data_ptr(), C.byref(u["desc"]), g2.data_ptr(), pool.data_ptr(), None, None, st)) if mode else \
                        (lambda: lib.sn_cab1_phase1(C.byref(src), C.byref(u["desc"]), g2.data_ptr(), pool.data_ptr(), None, None, st))
                    for _ in range(2):
                        assert f() == 0
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        f()
                    e1.record(); torch.cuda.synchronize()
                    res.append(e0.elapsed_time(e1) / 5 * 1e3)
                print(f"VAR {model} {T}x{h}x{w} {name:16s} CAB1 {res[0]:8.1f} us   CAB2 {res[1]:8.1f} us", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:] or None)
    else:
        time_all(sys.argv[2:] or list(VARIANTS))
