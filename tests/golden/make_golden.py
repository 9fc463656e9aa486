#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE itself on CPU.

Run ONLY in the build container (needs /root/reference; the GPU box has neither
the reference nor any use for this script):

    python tests/golden/make_golden.py

The reference's arch files import only torch/numpy, so they are loaded straight
from their file path (importing ``basicsr.models.archs...`` would pull in
cv2/lmdb).  Nothing from the reference is copied: the outputs
below are data (inputs, expected outputs, key lists).

Inputs and weights are produced by the integer-only generators in
``shiftnet_amd.synth`` / ``shiftnet_amd.weights`` so they never need storing;
each fixture records a crc32 of the generated input so drift is detected.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "shift-net_amd"))
from shiftnet_amd import synth  # noqa: E402
from shiftnet_amd.spec import VARIANTS  # noqa: E402
from shiftnet_amd.weights import synth_state_dict  # noqa: E402

REF = "/root/reference/basicsr/models/archs"
SEED = 1234


def load_ref(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, f"{REF}/{name}.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def t32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def clip_tensor(frames_u8):
    """[T,H,W,3] uint8 -> [1,T,3,H,W] float32 / 255 exactly as numpy2tensor does (test_deblur.py:191-200)."""
    ts = [torch.from_numpy(np.ascontiguousarray(f.astype("float64").transpose(2, 0, 1))).float().mul_(1.0 / 255) for f in frames_u8]
    return torch.stack(ts).unsqueeze(0)


def psnr(a, b):
    mse = (a.float() - b.float()).pow(2).mean().item()
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def bf16_run(mod, name, pf, x, nm):
    """The reference's OWN CPU bf16 run (net.bfloat16()): the precision yardstick for the bf16 HIP path."""
    nb = mod.GShiftNet(future_frames=pf[1], past_frames=pf[0])
    nb.load_state_dict(synth_state_dict(name, SEED), strict=True)
    nb = nb.eval().bfloat16()
    return (nb(x.bfloat16(), nm.bfloat16()) if nm is not None else nb(x.bfloat16())).float()


def quadrant_case(name="gshift_denoise1", T=6, H=96, W=128, sigma=30.0 / 255.0):
    """G. The denoise CLI's four overlapping quadrants (inference/test_denoise.py:153-173) with a FIXED noise tensor: the reference network on
    the four crops, stitched with the CLI's own slice arithmetic (restated here: the CLI file itself needs imageio / cv2 / skimage)."""
    mod = load_ref(name)
    net = mod.GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name, SEED), strict=True)
    net.eval()
    _, sharp = synth.blurred_clip(T, H, W, seed=11)
    noise = torch.from_numpy(synth.unit_noise((1, T, 3, H, W), seed=12)).float() * sigma
    x = clip_tensor(sharp) + noise
    B, N = 1, T
    std_map = torch.FloatTensor([sigma]).view(1, 1, 1, 1, 1)
    pad_h, pad = 32 - (H // 2 % 16), 32 - (W // 2 % 16)
    hh, ww = H // 2 + pad_h, W // 2 + pad
    out = torch.zeros(N - 4, 3, H, W)
    o1 = net(x[:, :, :, 0:hh, 0:ww], std_map.expand(B, N, 1, hh, ww))
    o2 = net(x[:, :, :, 0:hh, W // 2 - pad:], std_map.expand(B, N, 1, hh, ww))
    o3 = net(x[:, :, :, H // 2 - pad_h:, 0:ww], std_map.expand(B, N, 1, hh, ww))
    o4 = net(x[:, :, :, H // 2 - pad_h:, W // 2 - pad:], std_map.expand(B, N, 1, hh, ww))
    out[..., 0:H // 2, 0:W // 2] = o1.float()[..., 0:-pad_h, 0:-pad]
    out[..., 0:H // 2, W // 2:] = o2.float()[..., 0:-pad_h, pad:]
    out[..., H // 2:, 0:W // 2] = o3.float()[..., pad_h:, 0:-pad]
    out[..., H // 2:, W // 2:] = o4.float()[..., pad_h:, pad:]
    # the stitched result must NOT equal one whole-frame forward (the quadrants see their own borders): keep that distance as a guard
    whole = net(x, std_map.expand(B, N, 1, H, W))
    np.savez_compressed(f"{HERE}/quadrants_{name}.npz", in_crc=np.uint32(synth.crc(sharp)), noise_crc=np.uint32(synth.crc(noise.numpy())),
                        dims=np.array([T, H, W], np.int32), sigma=np.float32(sigma), out=out.numpy(),
                        whole_vs_stitched_maxabs=np.float32((whole - out).abs().max().item()))
    print("quadrants", name, tuple(out.shape), "whole-frame vs stitched max-abs", (whole - out).abs().max().item())


def main():
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    if "--only-quadrants" in sys.argv:
        quadrant_case()
        return

    for name, V in VARIANTS.items():
        mod = load_ref(name)
        sd = synth_state_dict(name, SEED)

        # ---- A. checkpoint layout --------------------------------------------------
        net = mod.GShiftNet()
        ref_sd = net.state_dict()
        groups = {}
        for k, v in ref_sd.items():
            groups.setdefault(v.data_ptr(), []).append(k)
        with open(f"{HERE}/state_keys_{name}.json", "w") as f:
            json.dump({"keys": [[k, list(v.shape)] for k, v in ref_sd.items()],
                       "alias": [g for g in groups.values() if len(g) > 1],
                       "n_params": sum(p.numel() for p in net.parameters())}, f)
        net.load_state_dict(sd, strict=True)
        net.eval()

        # ---- B. GSTS gather on index-coded input (exact) ---------------------------
        C = V.c1
        T, h, w = 4, 24, 24
        idx = (np.arange(T * C * h * w, dtype=np.float32) + 1).reshape(T, C, h, w)
        blk = net.stage1.decoder_level1
        out = {}
        for rev in (False, True):
            u = blk.channel_shift(t32(idx), reverse=rev)
            out["rev" if rev else "fwd"] = u.numpy().astype(np.int32)
        np.savez_compressed(f"{HERE}/gsts_gather_{name}.npz", **out)

        # ---- C. block fixtures ------------------------------------------------------
        T, h, w = 3, 12, 20
        x = t32(synth.unit_noise((T, C, h, w), seed=11))
        res = {"x_crc": np.uint32(synth.crc(x.numpy()))}
        unit = blk.encoder_level1            # Sequential(CAB2, CAB1) of the first (forward) unit
        unit_r = blk.encoder_level1_1        # second (reverse) unit
        u_f = blk.channel_shift(x, reverse=False)
        u_r = blk.channel_shift(x, reverse=True)
        res["cab2_fwd"] = unit[0](u_f).numpy()
        res["cab2_rev"] = unit_r[0](u_r).numpy()
        res["cab1"] = unit[1](x).numpy()
        res["unit_fwd"] = unit(u_f).numpy()
        res["shift_block"] = blk(x).numpy()
        x0 = t32(synth.unit_noise((T, V.c0, h, w), seed=12))
        res["x0_crc"] = np.uint32(synth.crc(x0.numpy()))
        res["cab_c0"] = net.stage1.concat(x0).numpy()
        res["cab_c1"] = net.stage1.skip_attn1(x).numpy()
        res["tfr_unet"] = net.orb1(x0).numpy()
        res["down12_c1"] = net.stage1.down12(x).numpy()
        res["up21_c1"] = net.stage1.up21(net.stage1.down12(x), x).numpy()
        res["pixshuf"] = net.stage1.upsample0(x).numpy()
        res["down01"] = net.stage1.down01(x0).numpy()
        x0b = t32(synth.unit_noise((T, V.c0, 16, 24), seed=13))     # "+" needs H,W % 8 == 0
        res["x0b_crc"] = np.uint32(synth.crc(x0b.numpy()))
        res["stage1"] = net.stage1(x0b).numpy()
        if V.shift_cab:
            res["shift_cab_fwd"] = net.stage1.encoder_level1(x, reverse=False).numpy()
            res["shift_cab_rev"] = net.stage1.encoder_level1_1(x, reverse=True).numpy()
        np.savez_compressed(f"{HERE}/blocks_{name}.npz", **res)

        # ---- D. whole-net fixtures --------------------------------------------------
        T, H, W = 7, 48, 64
        blur, sharp = synth.blurred_clip(T, H, W, seed=3)
        xin = clip_tensor(blur)
        res = {"in_crc": np.uint32(synth.crc(blur))}
        for tag, pf in (("p2f2", (2, 2)), ("default", (V.past, V.future))):
            net.num_fb, net.num_ff = pf
            if V.denoise:
                nm = torch.full((1, T, 1, H, W), 30.0 / 255.0)
                y = net(xin, nm)
            else:
                y = net(xin)
            res[tag] = y.numpy()
            res[tag + "_ref_bf16_psnr"] = np.float32(psnr(bf16_run(mod, name, pf, xin, nm if V.denoise else None), y))
        np.savez_compressed(f"{HERE}/net_{name}.npz", **res)
        print(name, "done", {k: getattr(v, "shape", v) for k, v in res.items()})

    # ---- E. config 1: Shift-Net-s, [1,5,3,256,256], past/future 2/2 ------------------
    name = "gshift_deblur2"
    net = load_ref(name).GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name, SEED), strict=True)
    net.eval()
    blur, sharp = synth.blurred_clip(5, 256, 256, seed=5)
    mod2 = load_ref(name)
    y = net(clip_tensor(blur))
    np.savez_compressed(f"{HERE}/config1_{name}.npz", in_crc=np.uint32(synth.crc(blur)), out=y.numpy(),
                        ref_bf16_psnr=np.float32(psnr(bf16_run(mod2, name, (2, 2), clip_tensor(blur), None), y)))

    # ---- F. CLI windowing: 12 frames, one_len=4 (test_deblur.py:111-137) -------------
    blur, sharp = synth.blurred_clip(12, 32, 40, seed=7)
    one_len = 4
    k_len = (12 - 4) // one_len
    outs, win, outs_b = [], [], []
    for kk in range(k_len):
        lo, hi = kk * one_len, kk * one_len + one_len + 4
        win.append([lo, hi, kk * one_len + 2, kk * one_len + 2 + one_len])
        outs.append(net(clip_tensor(blur[lo:hi])).numpy())
        outs_b.append(bf16_run(mod2, name, (2, 2), clip_tensor(blur[lo:hi]), None).numpy())
    np.savez_compressed(f"{HERE}/windows_{name}.npz", in_crc=np.uint32(synth.crc(blur)),
                        windows=np.array(win, np.int32), out=np.concatenate(outs, 0),
                        ref_bf16_psnr=np.float32(psnr(torch.from_numpy(np.concatenate(outs_b, 0)), torch.from_numpy(np.concatenate(outs, 0)))))
    quadrant_case()
    print("all fixtures written to", HERE)


if __name__ == "__main__":
    main()
