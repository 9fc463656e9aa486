import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    sys.path.insert(0, p)
import torch
from shiftnet_amd.engine import Act, Engine, Plan
from shiftnet_amd.spec import VARIANTS
from shiftnet_amd.weights import synth_state_dict
dev = torch.device("cuda:0")
name = "gshift_deblur2"
plan = Plan(VARIANTS[name], synth_state_dict(name), dev)
new, old = Engine(plan), Engine(plan)
old.conv_tiles = True
for pre, c in (("stage1.concat.", 14), ("orb1.encoder_level2.1.", 18)):
    cs = (c + 7) // 8 * 8
    for T, h, w in ((3, 45, 150), (2, 24, 40), (2, 8, 32), (1, 16, 64), (3, 2, 3), (2, 37, 33), (5, 72, 200)):
        torch.manual_seed(1)
        x = torch.zeros((T, h, w, cs), dtype=torch.bfloat16, device=dev); x[..., :c] = torch.randn((T, h, w, c), device=dev).to(torch.bfloat16)
        m = torch.zeros((T, h, w, cs), dtype=torch.bfloat16, device=dev); m[..., :c] = torch.randn((T, h, w, c), device=dev).to(torch.bfloat16)
        ca = (0.5 + torch.rand(T, 16 * ((cs + 15) // 16), device=dev)).float()
        a = new.conv(pre + "body.2", [Act(m, c)], res=Act(x, c), oscale=ca).t.float()
        b = old.conv(pre + "body.2", [Act(m, c)], res=Act(x, c), oscale=ca).t.float()
        torch.cuda.synchronize()
        d = (a - b).abs()
        bad = (d > 0).nonzero()
        print(pre, c, (T, h, w), "max diff", d.max().item(), "n bad", bad.shape[0], "of", d.numel(), "first", bad[:6].tolist(), flush=True)
        if bad.shape[0]:
            t_, y_, x_, ch_ = bad[0].tolist()
            print("   new", a[t_, y_, x_, :].tolist(), "\n   old", b[t_, y_, x_, :].tolist(), "\n   res", x[t_, y_, x_, :].float().tolist())
            print("   bad frames", sorted(set(bad[:, 0].tolist())), "rows", sorted(set(bad[:, 1].tolist()))[:20], "cols", sorted(set(bad[:, 2].tolist()))[:40], "chans", sorted(set(bad[:, 3].tolist())))
