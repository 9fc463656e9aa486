#!/usr/bin/env python3
"""Run every kernel of a GSTS unit (and a CAB, a Shift_CAB roll) several times on the same input and report which output is not
bit-reproducible.  usage: determinism_probe.py [variant] [T h w] [repeats]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from shiftnet_amd import lib as L
    from shiftnet_amd.engine import Act, Engine, Plan
    from shiftnet_amd.spec import VARIANTS
    from shiftnet_amd.weights import synth_state_dict
    name = sys.argv[1] if len(sys.argv) > 1 else "gshift_denoise1"
    T, h, w = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (36, 68, 112)
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    dev = torch.device("cuda:0")
    V = VARIANTS[name]
    eng = Engine(Plan(V, {k: v.bfloat16() for k, v in synth_state_dict(name).items()}, dev))
    lib, P = eng.lib, eng.P
    c = V.c1
    g = torch.Generator(device="cpu").manual_seed(1)
    x = Act(torch.randn(T, h, w, c, generator=g).to(torch.bfloat16).to(dev), c)
    st = torch.cuda.current_stream().cuda_stream
    pre = "stage1.decoder_level1.encoder_level1.0."
    u = P.units[pre]
    bad = {}

    def rec(tag, t, store):
        torch.cuda.synchronize()
        ref = store.setdefault(tag, t.clone())
        if not torch.equal(ref, t):
            d = (ref.float() - t.float()).abs()
            bad[tag] = bad.get(tag, 0) + 1
            print(f"  NON-DETERMINISTIC {tag}: {int((d > 0).sum())} elements differ, max {d.max().item():.4g}", flush=True)

    for mode in (1, 0):
        store = {}
        for r in range(reps):
            src = eng._unit_src(x, mode)
            hw_ptr = None
            if mode:
                hwb = eng._new(T, h, w, c // 2)
                L.check(lib.sn_gsts_shiftconv(C.byref(src), P.offs.data_ptr(), u["w1"].data_ptr(), hwb.data_ptr(), st), "k0")
                rec(f"m{mode}.K0", hwb, store)
                hw_ptr = hwb.data_ptr()
            mst = not V.grouped_rep
            g1 = torch.empty((T, h, c, lib.sn_planar_pitch(w)), dtype=torch.bfloat16, device=dev) if mst else eng._new(T, h, w, c)
            pool1 = torch.empty((T, lib.sn_lngate_blocks(h, w), c), dtype=torch.float32, device=dev) if V.denoise else None
            L.check(lib.sn_ln_gemm_gate(C.byref(src), hw_ptr, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(), u["w_dw3_h2"].data_ptr(), g1.data_ptr(),
                                        pool1.data_ptr() if pool1 is not None else None, 2 if mst else 0, st), "k12")
            rec(f"m{mode}.K12.g1", g1, store)
            ca1_ptr = None
            if V.denoise:
                rec(f"m{mode}.K12.pool1", pool1, store)
                ca1 = eng.ca_mlp(pre + "ca1", pool1, h * w)
                rec(f"m{mode}.ca1", ca1, store)
                ca1_ptr = ca1.data_ptr()
            g2 = eng._new(T, h, w, c)
            if mst:
                pool2 = torch.empty((T, lib.sn_dw5m_blocks(h, w), c), dtype=torch.float32, device=dev)
                L.check(lib.sn_dw5m_gemm_gate(g1.data_ptr(), ca1_ptr, u["w_toep5"].data_ptr(), u["w_gate"].data_ptr(), g2.data_ptr(), pool2.data_ptr(), T, h, w, c, st), "k3m")
            else:
                pool2 = torch.empty((T, lib.sn_grp5_blocks(h, w), c), dtype=torch.float32, device=dev)
                L.check(lib.sn_grp5_gemm_gate(g1.data_ptr(), ca1_ptr, u["w_grp"].data_ptr(), u["w_gate"].data_ptr(), g2.data_ptr(), pool2.data_ptr(), T, h, w, c, st), "k3g")
            rec(f"m{mode}.K3.g2", g2, store)
            rec(f"m{mode}.K3.pool2", pool2, store)
            ca2 = eng.ca_mlp(pre + "ca2", pool2, h * w)
            rec(f"m{mode}.ca2", ca2, store)
            y = eng._new(T, h, w, c)
            b_out = u["b_out"].data_ptr() if u["b_out"] is not None else None
            L.check((lib.sn_gsts_cab2_phase2 if src.mode else lib.sn_cab1_phase2)(C.byref(src), g2.data_ptr(), ca2.data_ptr(), u["w_out"].data_ptr(), b_out, y.data_ptr(), st), "k4")
            rec(f"m{mode}.K4.y", y, store)
    store = {}
    c0 = V.c0
    x0 = Act(torch.randn(T, 2 * h, 2 * w, prep_ceil8(c0), generator=g).to(torch.bfloat16).to(dev), c0)
    x0.t[..., c0:] = 0
    for r in range(reps):
        rec("cab_c0", eng.cab("stage1.concat.", x0).t, store)
        rec("down12", eng.down("orb1.down12.", x0).t, store)
        rec("cab_c1", eng.cab("stage1.skip_attn1.", x).t, store)
        if V.shift_cab:
            rec("roll", eng.temporal_roll(x, True).t, store)
    # the units through the engine's own naf() (its allocation pattern, not this script's) with allocator churn in between: this
    # is what exposed the round-2 lifetime bug (scale tensor of CALayer2 freed before the K3 launch that reads it)
    for mode in (0, 1, 2):
        for r in range(4 * reps):
            rec(f"naf.m{mode}", eng.naf(pre, x, mode).t, store)
            junk = [torch.empty(int(torch.randint(1, 400, (1,))) * 1024, dtype=torch.uint8, device=dev) for _ in range(8)]
            del junk
    # the whole module forward at the level-1 size
    import importlib
    net = importlib.import_module(f"basicsr.models.archs.{name}").GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name), strict=True)
    net = net.to(torch.bfloat16).cuda().eval()
    xin = torch.rand(1, T, 3, 4 * h, 4 * w, generator=g).to(torch.bfloat16).to(dev)
    nm = torch.full((1, T, 1, 4 * h, 4 * w), 30.0 / 255.0, device=dev).bfloat16() if V.denoise else None
    with torch.no_grad():
        for r in range(reps):
            rec("forward", net(xin, nm) if V.denoise else net(xin), store)
    print("summary:", bad if bad else "all outputs bit-reproducible")


def prep_ceil8(c):
    return (c + 7) // 8 * 8


if __name__ == "__main__":
    main()
