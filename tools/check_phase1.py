#!/usr/bin/env python3
"""Fused phase-1 kernel (sn_gsts_cab2_phase1 / sn_cab1_phase1) against the oracle, with a breakdown of where any error sits (row / column / channel -> wave q,
lane group g, register r), then its time at the level-1 and level-2 sizes of config 2 next to sn_ln_gemm_gate + sn_dw5m_gemm_gate.
usage: check_phase1.py [--no-time] [--name gshift_deblur1|gshift_deblur2] [--sizes T,h,w;T,h,w] [--teams 0,1,2,4,8]
The kernel is csrc/sn_phase1r.hip (role-split, RepConv on the matrix cores, C = 64 / 80); --teams times sn_phase1_opts.team values (0 = the library's choice)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from oracle import shiftnet_oracle as O
    from shiftnet_amd import lib as L, synth
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    def arg(k, d):
        return sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
    name = arg("--name", "gshift_deblur2")
    key = "p1r"
    dev = torch.device("cuda:0")
    V, OV = VARIANTS[name], O.VARIANTS[name]
    sd = synth_state_dict(name)
    eng = Engine(Plan(V, sd, dev))
    lib, P = eng.lib, eng.P
    st = torch.cuda.current_stream().cuda_stream
    Cc = V.c1
    blk = "stage1.decoder_level1."

    def ref_g2(q, v):
        a = O._conv(sd, f"{q}body.0.", v)
        a = O._conv(sd, f"{q}body.1.conv_2.", a, groups=a.shape[1]) + a
        a1, a2 = a.chunk(2, dim=1)
        g = O._rep_conv(sd, f"{q}body.3.", a1 * a2, groups=(Cc // 8 if V.grouped_rep else Cc))
        b1, b2 = O._conv(sd, f"{q}body.4.", g).chunk(2, dim=1)
        return b1 * torch.sigmoid(b2)

    def run_p1(pre, xd, mode, hwb):
        u = P.units[pre][key]
        T, h, w, c = xd.shape
        src = L.UnitSrc(xd.data_ptr(), T, h, w, c, mode, 1 if (V.wrap and mode) else 0)
        g2 = torch.full((T, h, w, c), float("nan"), dtype=torch.bfloat16, device=dev)
        nblk = lib.sn_phase1_pool_blocks(T, h, w)
        pool = torch.zeros((T, nblk, c), dtype=torch.float32, device=dev)
        L.check(L.cab_phase1(lib, src, hwb.data_ptr() if hwb is not None else None, u["desc"], g2.data_ptr(), pool.data_ptr(), st), "phase 1")
        torch.cuda.synchronize()
        return g2, pool, nblk

    bad = 0
    sizes = ((3, 7, 21), (2, 5, 9), (3, 20, 44), (2, 13, 70), (3, 40, 200), (2, 97, 130), (1, 64, 42), (1, 3, 40), (2, 184, 328))
    if "--sizes" in sys.argv:
        sizes = tuple(tuple(int(v) for v in t.split(",")) for t in arg("--sizes", "").split(";"))
    for (T, h, w) in sizes:
        x = torch.from_numpy(synth.unit_noise((T, Cc, h, w), seed=81 + h)).bfloat16().float()
        xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
        for mode, rev, unit in ((0, False, "encoder_level1.1."), (1, False, "encoder_level1.0."), (2, True, "encoder_level1_1.0.")):
            pre = blk + unit
            with torch.no_grad():
                if mode:
                    u = O.gsts_gather(x, rev, OV.wrap)
                    hw = O._conv(sd, pre + "conv1.", u[:, Cc:], groups=Cc // 2).bfloat16().float()
                    ref = ref_g2(pre, O.layer_norm_2d(torch.cat((u[:, :Cc], hw), 1), sd[pre + "norm.weight"], sd[pre + "norm.bias"]))
                    hwb = hw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
                else:
                    ref = ref_g2(pre, O.layer_norm_2d(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"]))
                    hwb = None
            g2, pool, nblk = run_p1(pre, xd, mode, hwb)
            got = g2.float().cpu().permute(0, 3, 1, 2)
            scale = max(1.0, ref.abs().max().item())
            nan = torch.isnan(got).sum().item()
            err = (torch.nan_to_num(got, nan=1e9) - ref).abs()
            e = err.max().item() / scale
            sums = pool.sum(1).cpu()
            es = (sums - ref.sum((2, 3))).abs().max().item() / max(1.0, ref.sum((2, 3)).abs().max().item())
            ok = e < 1e-2 and es < 1e-2 and nan == 0
            print(f"{'ok ' if ok else 'BAD'} T{T} {h}x{w} mode {mode}: max err {e:.2e} x scale {scale:.1f}, pool err {es:.2e}, nan {nan}, blocks {nblk}", flush=True)
            if not ok:
                bad += 1
                em = torch.nan_to_num(err, nan=1e9, posinf=1e9).clamp(max=1e9)
                thr = 1e-2 * scale
                print("   bad fraction per frame:", [(f"{(em[t] > thr).float().mean().item():.3f}") for t in range(T)])
                print("   bad fraction per row   :", " ".join(f"{(em[:, :, y] > thr).float().mean().item():.2f}" for y in range(min(h, 24))))
                print("   bad fraction per column:", " ".join(f"{(em[:, :, :, xx] > thr).float().mean().item():.2f}" for xx in range(min(w, 64))))
                pc = (em > thr).float().mean((0, 2, 3))
                print("   bad fraction per channel:", " ".join(f"{v:.2f}" for v in pc.tolist()))
    print("PHASE1", "ALL OK" if bad == 0 else f"{bad} BAD CASES")
    if "--no-time" in sys.argv:
        return
    tsizes = ((20, 360, 640), (20, 180, 320)) if Cc == 64 else ((52, 360, 640), (52, 180, 320), (16, 540, 960))
    for (T, h, w) in tsizes:
        xd = torch.randn(T, h, w, Cc, device=dev).to(torch.bfloat16)
        hwb = torch.randn(T, h, w, Cc // 2, device=dev).to(torch.bfloat16)
        for mode, unit in ((0, "encoder_level1.1."), (1, "encoder_level1.0.")):
            pre = blk + unit
            u = P.units[pre]
            src = L.UnitSrc(xd.data_ptr(), T, h, w, Cc, mode, 1 if (mode and V.wrap) else 0)
            g2 = torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev)
            hp = hwb.data_ptr() if mode else None
            mst = not V.grouped_rep
            g1 = (torch.empty((T, h, Cc, lib.sn_planar_pitch(w)), dtype=torch.bfloat16, device=dev) if mst else torch.empty((T, h, w, Cc), dtype=torch.bfloat16, device=dev))
            pool2 = torch.empty((T, lib.sn_dw5m_blocks(h, w) if mst else lib.sn_grp5_blocks(h, w), Cc), dtype=torch.float32, device=dev)
            k3 = lib.sn_dw5m_gemm_gate if mst else lib.sn_grp5_gemm_gate
            calls = {}
            pools = {}
            nb = lib.sn_phase1_pool_blocks(T, h, w)
            for team in [int(v) for v in arg("--teams", "0").split(",")]:
                pools[team] = torch.zeros((T, nb, Cc), dtype=torch.float32, device=dev)
                calls[f"phase1 fused team {team}"] = (lambda team=team: L.cab_phase1(lib, src, hp, u["p1r"]["desc"], g2.data_ptr(), pools[team].data_ptr(), st,
                                                                                   None, L.Phase1Opts(None, 0, team)))
            calls["K12"] = lambda: lib.sn_ln_gemm_gate(C.byref(src), hp, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(), u["w_dw3_h2"].data_ptr(), g1.data_ptr(), None, 2 if mst else 0, st)
            calls["K3"] = lambda: k3(g1.data_ptr(), None, u["w_toep5" if mst else "w_grp"].data_ptr(), u["w_gate"].data_ptr(), g2.data_ptr(), pool2.data_ptr(), T, h, w, Cc, st)
            for k, f in calls.items():
                for _ in range(2):
                    L.check(f(), k)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    f()
                e1.record(); torch.cuda.synchronize()
                print(f"TIME {name} {T}x{h}x{w} mode {mode} {k:18s} {e0.elapsed_time(e1) / 6 * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
