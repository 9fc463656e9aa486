#!/usr/bin/env python3
"""Development aid: time ablated builds of the fused phase-1 kernel (csrc/sn_phase1.hip, -DP1_ABLATE=mask: parts of the kernel are
skipped, results are wrong) to see what its time is made of.  `build` compiles one small library per mask (no GPU needed);
`run` times them on the GPU at the level-1 size of config 2.   usage: p1_variants.py build|run [mask ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
OUT = os.path.join(ROOT, "shift-net_amd", "lib")
MASKS = [0, 1, 2, 4, 8, 12, 16, 32, 48, 64, 128, 256, 512, 60, 126]


def lib_path(m):
    return os.path.join(OUT, f"libp1_ablate_{m}.so")


def build(masks):
    for m in masks:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", f"-DP1_ABLATE={m}",
               "-o", lib_path(m), os.path.join(ROOT, "shift-net_amd", "csrc", "sn_phase1.hip")]
        subprocess.run(cmd, check=True)
        print("built", lib_path(m), flush=True)


def run(masks):
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
    xd = torch.randn(T, h, w, Cc, device=dev).to(torch.bfloat16)
    hwb = torch.randn(T, h, w, Cc // 2, device=dev).to(torch.bfloat16)
    for mode, unit in ((0, "encoder_level1.1."), (1, "encoder_level1.0.")):
        p1 = eng.P.units["stage1.decoder_level1." + unit]["p1"]
        src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 1 if mode else 0)
        for m in masks:
            if not os.path.exists(lib_path(m)):
                continue
            lib = C.CDLL(lib_path(m))
            vp, ci = C.c_void_p, C.c_int
            lib.sn_cab_phase1.argtypes = [C.POINTER(L.UnitSrc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
            lib.sn_cab_phase1_blocks.argtypes = [ci, ci, ci]
            nblk = lib.sn_cab_phase1_blocks(T, h, w)
            pool = torch.zeros((T, nblk, Cc), dtype=torch.float32, device=dev)
            g2 = torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev)
            f = lambda: lib.sn_cab_phase1(C.byref(src), hwb.data_ptr() if mode else None, p1["wfrag1"].data_ptr(), p1["bias"].data_ptr(),
                                          p1["wsum"].data_ptr(), p1["w3"].data_ptr(), p1["w5"].data_ptr(), p1["wfrag2"].data_ptr(),
                                          g2.data_ptr(), pool.data_ptr(), st)
            for _ in range(2):
                assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                f()
            e1.record(); torch.cuda.synchronize()
            print(f"mode {mode} ablate {m:4d}: {e0.elapsed_time(e1) / 6 * 1e3:8.1f} us", flush=True)
            if m & 2048:        # phase clocks: pool[t][blk][16 q + k] = cycles of phase k summed over the rows of one workgroup's wave q
                pc = pool.reshape(T * nblk, 4, 16)[:, :, :8].double().mean(0).cpu()        # [wave][phase]
                names = ["2nd gemm+store", "gemm1+ln", "3x3+gate", "5x5", "r write", "stage(+vmcnt)", "barrier", "loop top"]
                tot = pc.sum(1)
                for wv in range(4):
                    print("   wave", wv, " ".join(f"{names[k]} {100 * pc[wv, k] / tot[wv]:4.1f}%" for k in range(8)), f"| {tot[wv] / 1e3:.0f} kcycles", flush=True)


if __name__ == "__main__":
    masks = [int(a) for a in sys.argv[2:]] or MASKS
    (build if sys.argv[1] == "build" else run)(masks)
