"""Host weight packing (prep.py) x kernel data-flow conventions (tests/emu.py) == oracle, on CPU in fp32.

This is the no-GPU proof that every channel permutation / folding done on the host (gate pairing, LayerNorm and beta
folding, RepConv merging, conv K ordering, pad channels) matches what the kernels assume.
"""
import numpy as np
import pytest
import torch

import emu
from oracle import shiftnet_oracle as O
from shiftnet_amd import prep, synth
from shiftnet_amd.spec import VARIANTS, shift_table
from shiftnet_amd.weights import synth_state_dict


def nhwc(t, cs=None):
    a = t.permute(0, 2, 3, 1).contiguous().numpy()
    if cs is not None and cs > a.shape[-1]:
        a = np.concatenate([a, np.zeros(a.shape[:-1] + (cs - a.shape[-1],), np.float32)], -1)
    return a


def nchw(a, c):
    return torch.from_numpy(np.ascontiguousarray(a[..., :c])).permute(0, 3, 1, 2)


def frag_f32(p):
    # emulate with the fp32 value of the bf16-rounded weights; compare against an oracle using the same rounding
    return p


@pytest.mark.parametrize("name", ["gshift_deblur2", "gshift_denoise2", "gshift_deblur1"])
def test_unit_emulation_matches_oracle(name):
    V, PV = O.VARIANTS[name], VARIANTS[name]
    sd = synth_state_dict(name)
    C, T, h, w = V.c1, 3, 6, 20
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=21))
    offs = np.array(shift_table(C), np.int8)
    pre = "stage1.decoder_level1.encoder_level1_1."        # a reverse unit; forward checked below too
    for reverse, mode in ((True, 2), (False, 1)):
        unit_pre = pre if reverse else "stage1.decoder_level1.encoder_level1."
        cur = nhwc(x)
        for sub, m in (("0.", mode), ("1.", 0)):
            q = unit_pre + sub
            i = 0
            hwb = None
            if m:
                hwb = emu.shiftconv(cur, offs, sd[q + "conv1.weight"].reshape(C // 2, 9).numpy(), m, V.wrap)
            g = prep.pack_ln_gemm(sd[f"{q}body.0.weight"], sd[q + "norm.weight"], sd[q + "norm.bias"], C)
            a = emu.ln_gemm(cur, hwb, g["wfrag"], g["bias"].numpy(), m, V.wrap)
            g1 = emu.dw_gate(a, prep.pack_dw3_gate(sd[f"{q}body.1.conv_2.weight"], C).numpy())
            i = 3
            ca1 = None
            if V.denoise:
                mean = torch.from_numpy(g1.mean((1, 2)))
                wa, wb = sd[f"{q}body.3.conv_du.0.weight"].flatten(1), sd[f"{q}body.3.conv_du.2.weight"].flatten(1)
                ca1 = torch.sigmoid(torch.relu(mean @ wa.T) @ wb.T).numpy()
                i = 4
            if V.grouped_rep:
                dense = prep.pack_grouped_rep(sd[f"{q}body.{i}.conv_1.weight"], sd[f"{q}body.{i}.conv_2.weight"])
                pc = prep.pack_conv(dense, None, [C], C)
                g1 = emu.conv2d([g1 if ca1 is None else g1 * ca1[:, None, None, :]], pc, bias=False)[..., :C]
                w5 = prep.identity_dw5(C).numpy(); ca1 = None
            else:
                w5 = prep.pack_dw5(sd[f"{q}body.{i}.conv_1.weight"], sd[f"{q}body.{i}.conv_2.weight"]).numpy()
            g2, sums = emu.dw_gemm_gate(g1, w5, prep.pack_gate_gemm(sd[f"{q}body.{i + 1}.weight"], C), ca1)
            mean = torch.from_numpy(sums / (h * w))
            wa, wb = sd[f"{q}body.{i + 3}.conv_du.0.weight"].flatten(1), sd[f"{q}body.{i + 3}.conv_du.2.weight"].flatten(1)
            ca2 = torch.sigmoid(torch.relu(mean @ wa.T) @ wb.T).numpy()
            o = prep.pack_out_gemm(sd[f"{q}body.{i + 4}.weight"], sd[q + "beta"], sd.get(f"{q}body.{i + 4}.bias"), C)
            cur = emu.scale_gemm_res(cur, g2, ca2, o["wfrag"], None if o["bias"] is None else o["bias"].numpy(), m, V.wrap)
        # oracle with bf16-rounded GEMM weights is not needed: tolerance covers the bf16 rounding of packed weights
        with torch.no_grad():
            ref = O.gsts_unit(sd, unit_pre, x, reverse, V)
        err = (nchw(cur, C) - ref).abs().max().item()
        assert err < 0.05 * max(1.0, ref.abs().max().item()), (name, reverse, err)


@pytest.mark.parametrize("case", ["c14", "cat3", "s2", "k2s2", "k5", "c64"])
def test_conv_emulation_matches_oracle(case):
    sd = synth_state_dict("gshift_deblur2")
    T, H, W = 2, 6, 20
    F = torch.nn.functional
    if case == "c14":
        w, b, cins, stride, pad = sd["conv_trans.weight"], sd["conv_trans.bias"], [14], 1, 1
    elif case == "cat3":
        w, b, cins, stride, pad = sd["rconcat.weight"], sd["rconcat.bias"], [14, 14, 14], 1, 1
    elif case == "s2":
        w, b, cins, stride, pad = sd["orb1.down12.down.weight"], sd["orb1.down12.down.bias"], [14], 2, 1
    elif case == "k2s2":
        w, b, cins, stride, pad = sd["stage1.down01.0.weight"], None, [14], 2, 0
    elif case == "k5":
        w, b, cins, stride, pad = sd["conv_last.weight"], None, [14], 1, 2
    else:
        w, b, cins, stride, pad = sd["stage1.skip_attn1.body.0.weight"], None, [64], 1, 1
    xs = [torch.from_numpy(synth.unit_noise((T, c, H, W), seed=31 + i)) for i, c in enumerate(cins)]
    cs = prep.ceil8(max(cins))
    p = prep.pack_conv(w, b, cins, cs)
    out = emu.conv2d([nhwc(x, cs) for x in xs], p, stride=stride, pad=pad)
    ref = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=pad)
    got = nchw(out, w.shape[0])
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 0.03 * max(1.0, ref.abs().max().item())
    if out.shape[-1] > w.shape[0]:
        assert np.abs(out[..., w.shape[0]:]).max() == 0.0       # pad channels stay zero
