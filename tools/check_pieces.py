#!/usr/bin/env python3
"""Dev check (GPU): a CAB2 / CAB1 launched in two frame pieces (the temporal split's launch plan) == the one-piece launch, bit for bit,
for every variant and direction; prints the first intermediate tensor that differs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
from shiftnet_amd import engine as E, engine32 as E32, spec, synth           # noqa: E402
from shiftnet_amd.weights import synth_state_dict                              # noqa: E402


def run(name, dt, T=5, h=24, w=40):
    V = spec.VARIANTS[name]
    sd = synth_state_dict(name)
    plan = E.Plan(V, sd, torch.device("cuda:0")) if dt != torch.float32 else None
    eng = E.Engine(plan, dt) if dt != torch.float32 else E32.Engine32(E32.Plan32(V, sd, torch.device("cuda:0")))
    c = V.c1
    x = torch.from_numpy(synth.unit_noise((T, h, w, c), seed=3)).to(dt if dt != torch.float16 else torch.bfloat16).cuda()
    pre = "stage1.decoder_level1." + spec.UNIT_NAMES[0] + "."
    bad = 0
    for mode in (1, 2, 0):
        rec = []
        orig_new = eng._new

        def new(*a, **k):
            t = orig_new(*a, **k); rec.append(t); return t
        eng._new = new
        blk = pre + ("0." if mode else "1.")
        y0 = eng.naf(blk, E.Act(x, c), mode).t.clone(); r0 = [t.clone() for t in rec]; rec.clear()
        if mode:
            bt = 0 if mode == 1 else T - 1
            eng._split_pieces = lambda xx, m, circ: iter([(0, None, 1 if m == 1 else 0, T - 1), (1 if circ else 0, None, bt, 1)])
            eng.split = None
            y1 = eng.naf(blk, E.Act(x, c), mode).t.clone(); r1 = [t.clone() for t in rec]; rec.clear()
            del eng._split_pieces
            same = torch.equal(y0, y1)
            first = next((i for i, (a, b) in enumerate(zip(r0, r1)) if not torch.equal(a, b)), None)
            per_frame = [bool(torch.equal(y0[t], y1[t])) for t in range(T)]
            print(name, dt, "mode", mode, "two pieces == one piece:", same, "first differing _new tensor:", first, per_frame)
            bad += not same
        eng._new = orig_new
    return bad


if __name__ == "__main__":
    bad = 0
    for name, dt in (("gshift_deblur2", torch.bfloat16), ("gshift_deblur1", torch.bfloat16), ("gshift_denoise1", torch.bfloat16),
                     ("gshift_denoise2", torch.bfloat16), ("gshift_denoise1", torch.float32)):
        try:
            bad += run(name, dt)
        except Exception as e:                              # noqa: BLE001
            print(name, dt, "ERROR", type(e).__name__, e); bad += 1
    sys.exit(1 if bad else 0)
