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


@pytest.mark.parametrize("name,hw", [("gshift_deblur2", (20, 37)), ("gshift_deblur1", (33, 18)), ("gshift_deblur2", (3, 5))])
def test_shiftconv_on_the_matrix_cores_emulation_matches_oracle(name, hw):
    """K0 as a banded GEMM (round 6): the kernel's operand construction -- per-lane A selectors on the nine weights, the channel-planar window and its
    displaced 8-pixel B fragments, the padding masks of border tiles, the D layout -- replayed in numpy against the oracle's
    conv1(spatial_shift2(roll half)) (gshift_deblur1.py:470-503,223,251) and against the emulation of the VALU kernel, on ragged multi-tile maps."""
    V = O.VARIANTS[name]
    sd = synth_state_dict(name)
    C, T, (h, w) = V.c1, 2, hw
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=23))
    offs = np.array(shift_table(C), np.int8)
    for reverse, mode, unit in ((False, 1, "encoder_level1."), (True, 2, "encoder_level1_1.")):
        pre = "stage1.decoder_level1." + unit + "0."
        w1 = sd[pre + "conv1.weight"].reshape(C // 2, 9).numpy()
        got = emu.shiftconv_mfma(nhwc(x), offs, w1, mode, V.wrap)
        old = emu.shiftconv(nhwc(x), offs, w1, mode, V.wrap)
        _, half = O.temporal_roll(x, reverse, V.wrap)
        ref = torch.nn.functional.conv2d(O.spatial_shift(half.contiguous()), sd[pre + "conv1.weight"], padding=1, groups=C // 2)
        ref = ref.permute(0, 2, 3, 1).numpy()
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (name, mode, np.abs(got - ref).max())
        assert np.abs(got - old).max() <= 1e-5 * max(1.0, np.abs(ref).max())


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
            if V.grouped_rep:     # the grouped RepConv itself (pack_grouped_frag) is checked in test_grouped_frag_table; here: plain conv
                F = torch.nn.functional
                gin = torch.from_numpy(g1 if ca1 is None else g1 * ca1[:, None, None, :]).permute(0, 3, 1, 2)
                rr = (F.conv2d(gin, sd[f"{q}body.{i}.conv_1.weight"], padding=2, groups=C // 8)
                      + F.conv2d(gin, sd[f"{q}body.{i}.conv_2.weight"], padding=1, groups=C // 8) + gin)
                g1 = rr.permute(0, 2, 3, 1).contiguous().numpy()
                w5 = np.zeros((25, C), np.float32); w5[12] = 1.0; ca1 = None
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


def test_grouped_frag_table():
    """pack_grouped_frag (block-diagonal MFMA layout of the '+' RepConv) reproduces the oracle's RepConv when pushed through the
    kernel's documented slot conventions."""
    F = torch.nn.functional
    sd = synth_state_dict("gshift_deblur1")
    pre = "stage1.decoder_level1.encoder_level1.0.body.3."
    C, T, h, w = 80, 1, 6, 18
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=77))
    ref = F.conv2d(x, sd[pre + "conv_1.weight"], padding=2, groups=C // 8) + F.conv2d(x, sd[pre + "conv_2.weight"], padding=1, groups=C // 8) + x
    wg = emu.frag_to_np(prep.pack_grouped_frag(sd[pre + "conv_1.weight"], sd[pre + "conv_2.weight"]))     # [5][13][64][8]
    xn = np.zeros((h + 4, w + 4, C), np.float32); xn[2:-2, 2:-2] = nhwc(x)[0]
    out = np.zeros((h, w, C), np.float32)
    for oy in range(h):
        for ox0 in range(0, w, 16):
            for mt in range(C // 16):
                bfrag = np.zeros((13, 64, 8), np.float32)
                for lane in range(64):
                    g, p = lane >> 4, lane & 15
                    ox = min(ox0 + p, w - 1)
                    for s in range(13):
                        tap = 2 * s + (g >> 1)
                        tap = tap if tap < 25 else 0
                        dy, dx = divmod(tap, 5)
                        c0 = 16 * mt + (g & 1) * 8
                        bfrag[s, lane] = xn[oy + dy, ox + dx, c0:c0 + 8]
                regs = emu.mfma_tiles(wg[mt:mt + 1], bfrag)[0]
                for lane in range(64):
                    g, p = lane >> 4, lane & 15
                    if ox0 + p < w:
                        out[oy, ox0 + p, 16 * mt + g * 4: 16 * mt + g * 4 + 4] = regs[lane]
    got = torch.from_numpy(out).permute(2, 0, 1)[None]
    assert (got - ref).abs().max().item() < 0.03 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("k", [3, 5])
def test_toeplitz_band_table(k):
    """prep.pack_toeplitz: rebuild the 16 x 32 A operand of every (channel, kernel row) from the padded bands the way the
    lanes of sn_dw5m_gemm_gate address them (start s0 = k//2 - 1 - m + 8g, odd starts from the shifted copy, dword reads),
    then check that A @ (32 input columns starting 8 left of the tile) is the depthwise row convolution."""
    rng = np.random.default_rng(5)
    c = 8
    wk = torch.from_numpy(rng.standard_normal((k * k, c)).astype(np.float32)).bfloat16().float()
    tab = prep.pack_toeplitz(wk, k).float().numpy()                   # [c][k][2][20]
    assert tab.shape == (c, k, 2, 20)
    rec = tab.reshape(c, k, 40)
    A = np.zeros((c, k, 16, 32), np.float32)
    for m in range(16):
        for g in range(4):
            s0 = k // 2 - 1 - m + 8 * g
            if s0 < 0 or s0 > 11:
                s0 = 12
            tw = 10 + (s0 - 1) // 2 if s0 & 1 else s0 // 2            # dword offset in the 20-dword record
            A[:, :, m, 8 * g:8 * g + 8] = rec[:, :, 2 * tw:2 * tw + 8]
    x = rng.standard_normal((c, k + 3, 16 + 32)).astype(np.float32)    # rows, columns; tile starts at column 8
    w = wk.numpy().T.reshape(c, k, k)
    for ch in range(c):
        for oy in range(3):
            got = sum(A[ch, dy] @ x[ch, oy + dy, 0:32] for dy in range(k))
            ref = np.array([sum(w[ch, dy, dx] * x[ch, oy + dy, 8 + m + dx - k // 2] for dy in range(k) for dx in range(k))
                            for m in range(16)])
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_matrix_core_stencil_emulation_matches_valu_path():
    """sn_dw5m_gemm_gate's conventions (band records read as first/last dword + DPP neighbours, planar window starting 8
    columns left of the tile, natural-K gate GEMM) against the already oracle-checked emulation of the VALU kernel."""
    name = "gshift_deblur2"
    sd = synth_state_dict(name)
    C, T, h, w = 64, 1, 11, 70                                         # ragged in both directions, two tiles in x and y
    q = "stage1.decoder_level1.encoder_level1.0."
    rng = np.random.default_rng(3)
    g1 = torch.from_numpy(rng.standard_normal((T, h, w, C)).astype(np.float32)).bfloat16().float().numpy()
    w5 = prep.pack_dw5(sd[f"{q}body.3.conv_1.weight"], sd[f"{q}body.3.conv_2.weight"])
    wg = prep.pack_gate_gemm(sd[f"{q}body.4.weight"], C)
    ca = rng.uniform(0.5, 1.5, (T, C)).astype(np.float32)
    ref, ref_sums = emu.dw_gemm_gate(g1, w5.to(torch.bfloat16).float().numpy(), wg, ca)
    got, got_sums = emu.dw5m_gemm_gate(g1, prep.pack_toeplitz(w5, 5), wg, ca)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got_sums, ref_sums, rtol=1e-4, atol=1e-3)


def test_cab_pooled_mean_closed_form():
    """The algebra behind sn_cab_ca: the spatial mean of res = conv3x3(mid) (zero padding, no bias) follows from the channel
    sums of `mid`, its first/last rows and columns and its four corners -- so the CALayer scale of a CAB is known before the
    second conv runs.  Tap (ky, kx) reads mid(p + (ky-1, kx-1)): dy = +1 never reaches row 0, dy = -1 never reaches row h-1."""
    rng = np.random.default_rng(9)
    c, h, w = 6, 7, 9
    mid = torch.from_numpy(rng.standard_normal((1, c, h, w)).astype(np.float32))
    w2 = torch.from_numpy(rng.standard_normal((c, c, 3, 3)).astype(np.float32))
    ref = torch.nn.functional.conv2d(mid, w2, padding=1).mean((2, 3))[0].numpy()
    m = mid[0].numpy()
    tot, r0, r1, c0, c1 = m.sum((1, 2)), m[:, 0].sum(1), m[:, -1].sum(1), m[:, :, 0].sum(1), m[:, :, -1].sum(1)
    k00, k01, k10, k11 = m[:, 0, 0], m[:, 0, -1], m[:, -1, 0], m[:, -1, -1]
    rowex = [r1, 0.0, r0]                    # excluded row for ky = 0, 1, 2
    colex = [c1, 0.0, c0]
    cor = [[k11, 0.0, k10], [0.0, 0.0, 0.0], [k01, 0.0, k00]]
    got = np.zeros(c, np.float32)
    for ky in range(3):
        for kx in range(3):
            s_tap = tot - rowex[ky] - colex[kx] + cor[ky][kx]            # [cin]
            got += w2[:, :, ky, kx].numpy() @ s_tap
    np.testing.assert_allclose(got / (h * w), ref, rtol=1e-4, atol=1e-5)


def test_packed_fp16_weight_words():
    """Operand words of the packed stencil: pk_f16_words (K12: v_pk_fma_f16, word k = fp16 of elements 2k | 2k+1 << 16, round to nearest)."""
    import numpy as np
    import torch
    from shiftnet_amd import prep
    g = torch.Generator().manual_seed(5)
    w = torch.randn(9, 64, generator=g) * 0.3
    w[0, 0], w[0, 1] = 1.0 + 2.0 ** -12, -65504.0          # rounds to 1.0 in fp16; largest finite fp16
    words = prep.pk_f16_words(w)
    assert words.dtype == torch.int32 and words.shape == (9, 32)
    u = words.numpy().astype(np.int64) & 0xFFFFFFFF
    lo = (u & 0xFFFF).astype(np.uint16).view(np.float16)
    hi = (u >> 16).astype(np.uint16).view(np.float16)
    ref = w.to(torch.float16).numpy()
    assert np.array_equal(lo, ref[:, 0::2]) and np.array_equal(hi, ref[:, 1::2])
    assert lo[0, 0] == np.float16(1.0) and hi[0, 0] == np.float16(-65504.0)


def test_conv32_split_packing_reconstructs_the_weights():
    """prep.pack_conv32_split (A fragments of the split-precision fp32 conv, csrc/sn_f32.hip: conv32s_kernel): decoding the fragments exactly as
    the kernel addresses them -- M-tile, k-step, lane (row l & 15, k-slots 8 (l >> 4) ..), hi + lo -- gives the weights back to 2^-16, zeros in
    every padding slot and, for the grouped conv, in the other group's half of the block-diagonal."""
    g = torch.Generator().manual_seed(11)
    w = torch.randn(36, 36, 3, 3, generator=g)
    f = prep.pack_conv32_split(w).float().numpy()                      # [2][MT][KS][64][8]
    full = f[0] + f[1]
    co, ci, k = 36, 36, 3
    ncb = (ci + 31) // 32
    assert f.shape == (2, 3, 9 * ncb, 64, 8)
    rec = np.zeros((48, 64, 3, 3), np.float32)
    for m in range(3):
        for tap in range(9):
            for cb in range(ncb):
                for lane in range(64):
                    for j in range(8):
                        rec[16 * m + (lane & 15), 32 * cb + 8 * (lane >> 4) + j, tap // 3, tap % 3] = full[m, tap * ncb + cb, lane, j]
    wn = w.numpy()
    assert np.abs(rec[:co, :ci] - wn).max() <= 2.0 ** -15 * np.abs(wn).max()
    assert not rec[co:].any() and not rec[:, ci:].any()
    wg = torch.randn(32, 8, 5, 5, generator=g)                           # 4 groups of 8 -> 2 M-tiles
    fg = prep.pack_conv32_split(wg, 4).float().numpy()
    fullg = fg[0] + fg[1]
    assert fg.shape == (2, 2, 13, 64, 8)
    for m in range(2):
        for s_ in range(13):
            for lane in range(64):
                r, gq = lane & 15, lane >> 4
                tap = 2 * s_ + (gq >> 1)
                for j in range(8):
                    want = wg[16 * m + r, j, tap // 5, tap % 5].item() if (tap < 25 and (r >> 3) == (gq & 1)) else 0.0
                    assert abs(fullg[m, s_, lane, j] - want) <= 2.0 ** -15 * max(1.0, abs(want)), (m, s_, lane, j)


def test_split_precision_product_error_bound():
    """The arithmetic claim of the split-precision fp32 conv (csrc/sn_f32.hip: conv32s_kernel): with hi = bf16(v), lo = bf16(v - hi), the three
    kept products wh xh + wh xl + wl xh differ from w x by the dropped wl xl (~2^-16 |w x|) plus the residual of the two-term splits (~2^-17
    each): a dot product of 720 terms (3x3 conv over 80 channels) stays within 3e-5 of sum |w x| -- here against float64."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 720, generator=g)
    x = torch.randn(720, 256, generator=g) * 3.0

    def split(v):
        hi = v.to(torch.bfloat16).float()
        return hi, (v - hi).to(torch.bfloat16).float()
    wh, wl = split(w)
    xh, xl = split(x)
    assert ((w - wh - wl).abs() <= 2.0 ** -16 * w.abs() + 1e-30).all()          # a two-term bf16 split carries 16 significant bits
    got = (wh.double() @ xh.double()) + (wh.double() @ xl.double()) + (wl.double() @ xh.double())
    ref = w.double() @ x.double()
    bound = w.abs().double() @ x.abs().double()
    assert ((got - ref).abs() <= 3e-5 * bound).all()
    assert (got - ref).abs().max() > 0                                            # (it is an approximation, not an identity)


@pytest.mark.parametrize("name,reverse", [("gshift_deblur1", False), ("gshift_deblur1", True), ("gshift_deblur2", True)])
def test_phase1r_emulation_matches_oracle(name, reverse):
    """prep.pack_phase1r x the conventions of the role-split fused phase-1 kernel (csrc/sn_phase1r.hip, emu.cab_phase1r: normalised operands
    with the bias in constant-one k-slots, wave-paired rows, packed-fp16 3x3 words, RepConv -- grouped 8 -> 8 for the "+" model, depthwise
    for Shift-Net-s -- as the x-pair Toeplitz GEMM, fp16 second 1x1) == the reference's g2 for CAB2 and CAB1 (gshift_deblur1.py:183-255)."""
    V = O.VARIANTS[name]
    sd = synth_state_dict(name)
    C, T, h, w = V.c1, 2, 6, 13
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=29)).bfloat16().float()
    pre = "stage1.decoder_level1." + ("encoder_level1_1." if reverse else "encoder_level1.")
    mode = 2 if reverse else 1
    groups = C // 8 if V.grouped_rep else C

    def ref_g2(q, v):
        a = O._conv(sd, f"{q}body.0.", v)
        a = O._conv(sd, f"{q}body.1.conv_2.", a, groups=a.shape[1]) + a
        a1, a2 = a.chunk(2, dim=1)
        g = O._rep_conv(sd, f"{q}body.3.", a1 * a2, groups=groups)
        b1, b2 = O._conv(sd, f"{q}body.4.", g).chunk(2, dim=1)
        return b1 * torch.sigmoid(b2)

    def pk(q):
        return prep.pack_phase1r(sd[f"{q}body.0.weight"], sd[q + "norm.weight"], sd[q + "norm.bias"], sd[f"{q}body.1.conv_2.weight"],
                                 sd[f"{q}body.3.conv_1.weight"], sd[f"{q}body.3.conv_2.weight"], sd[f"{q}body.4.weight"], C)
    with torch.no_grad():
        q = pre + "0."
        u = O.gsts_gather(x, reverse, V.wrap)
        hw = O._conv(sd, q + "conv1.", u[:, C:], groups=C // 2)
        ref = ref_g2(q, O.layer_norm_2d(torch.cat((u[:, :C], hw), 1), sd[q + "norm.weight"], sd[q + "norm.bias"]))
        got, sums = emu.cab_phase1r(nhwc(x), nhwc(hw.bfloat16().float()), pk(q), mode, V.wrap)
        err = (nchw(got, C) - ref).abs().max().item()
        assert err < 0.02 * max(1.0, ref.abs().max().item()), ("cab2", reverse, err)
        q = pre + "1."
        ref = ref_g2(q, O.layer_norm_2d(x, sd[q + "norm.weight"], sd[q + "norm.bias"]))
        got, _ = emu.cab_phase1r(nhwc(x), None, pk(q), 0, V.wrap)
        err = (nchw(got, C) - ref).abs().max().item()
        assert err < 0.02 * max(1.0, ref.abs().max().item()), ("cab1", err)


@pytest.mark.parametrize("name", ["gshift_denoise1", "gshift_denoise2"])
def test_phase1r_emulation_of_the_denoisers_two_passes(name):
    """The denoisers' inner CALayer2 on g1 (gshift_denoise1.py:224,257) through the role-split kernel's two passes (sn_phase1_opts): pass 1 =
    channel sums of g1 -> the layer's scale (the reference's conv_du MLP on the mean); pass 2 = g1 times that scale -> RepConv -> 1x1 -> gate2.
    emu.cab_phase1r x prep.pack_phase1r against the reference's g2 for CAB1."""
    V = O.VARIANTS[name]
    sd = synth_state_dict(name)
    C, T, h, w = V.c1, 2, 6, 13
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=31)).bfloat16().float()
    q = "stage1.decoder_level1.encoder_level1.1."
    groups = C // 8 if V.grouped_rep else C
    pk = prep.pack_phase1r(sd[f"{q}body.0.weight"], sd[q + "norm.weight"], sd[q + "norm.bias"], sd[f"{q}body.1.conv_2.weight"],
                           sd[f"{q}body.4.conv_1.weight"], sd[f"{q}body.4.conv_2.weight"], sd[f"{q}body.5.weight"], C)
    with torch.no_grad():
        v = O.layer_norm_2d(x, sd[q + "norm.weight"], sd[q + "norm.bias"])
        a = O._conv(sd, f"{q}body.0.", v)
        a = O._conv(sd, f"{q}body.1.conv_2.", a, groups=a.shape[1]) + a
        a1, a2 = a.chunk(2, dim=1)
        g1 = a1 * a2
        g1s = O.channel_attention(sd, f"{q}body.3.", g1)
        b1, b2 = O._conv(sd, f"{q}body.5.", O._rep_conv(sd, f"{q}body.4.", g1s, groups=groups)).chunk(2, dim=1)
        ref = b1 * torch.sigmoid(b2)
        sums = emu.cab_phase1r(nhwc(x), None, pk, 0, V.wrap, want_g1_sums=True)
        rs = g1.sum((2, 3)).numpy()
        assert np.abs(sums - rs).max() <= 2e-2 * max(1.0, np.abs(rs).max())
        mean = torch.from_numpy(sums / (h * w)).float()
        hid = torch.relu(mean @ sd[f"{q}body.3.conv_du.0.weight"].reshape(-1, C).T)
        ca = torch.sigmoid(hid @ sd[f"{q}body.3.conv_du.2.weight"].reshape(C, -1).T)
        got, _ = emu.cab_phase1r(nhwc(x), None, pk, 0, V.wrap, ca_in=ca.numpy())
        err = (nchw(got, C) - ref).abs().max().item()
        assert err < 0.02 * max(1.0, ref.abs().max().item()), (name, err)


@pytest.mark.parametrize("name", ["gshift_deblur1", "gshift_deblur2", "gshift_denoise1", "gshift_denoise2"])
def test_fp32_plan_merged_repconv_weights(name):
    """Plan32.rep_merged (the fp32 engine's single 5x5 RepConv): conv(g, merged) [+ g] equals the reference's three terms conv_1(g) + g +
    conv_2(g) (gshift_deblur1.py:143-165) to fp32 round-off, for the grouped 8 -> 8 form (identity left to the kernel's residual) and the
    depthwise form (identity merged at the centre tap); and the denoisers' per-frame channel scale commutes the way the engine applies it:
    loader + residual scale (grouped), output scale (depthwise)."""
    import torch.nn.functional as F
    from shiftnet_amd.engine32 import Plan32
    V = VARIANTS[name]
    sd = synth_state_dict(name)
    P = Plan32(V, sd, torch.device("cpu"))
    pre = "stage1.decoder_level1.encoder_level1.0."
    rp = f"{pre}body.{P.units[pre]['rep']}."
    c = V.c1
    grp = c // 8 if V.grouped_rep else c
    g = torch.from_numpy(synth.unit_noise((2, c, 12, 20), seed=7)).double()
    w5, w3 = sd[rp + "conv_1.weight"].double(), sd[rp + "conv_2.weight"].double()
    ref = F.conv2d(g, w5, padding=2, groups=grp) + g + F.conv2d(g, w3, padding=1, groups=grp)
    key = P.rep_merged(rp, identity=not V.grouped_rep)
    m = P.sd[key]
    assert m.dtype == torch.float32 and m.shape == sd[rp + "conv_1.weight"].shape
    got = F.conv2d(g, m.double(), padding=2, groups=grp) + (g if V.grouped_rep else 0)
    assert (got - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    assert P.rep_merged(rp, identity=not V.grouped_rep) == key and torch.equal(P.dsd[key], m)            # cached under one key, host and device copies
    # RepConv of g * s[t, c] (CALayer2 of the denoisers, gshift_denoise1.py): what the engine computes instead of materialising g * s
    sc = torch.rand(2, c, 1, 1, dtype=torch.float64) + 0.5
    ref_s = F.conv2d(g * sc, w5, padding=2, groups=grp) + g * sc + F.conv2d(g * sc, w3, padding=1, groups=grp)
    if V.grouped_rep:       # iscale on the loader, rscale on the residual
        got_s = F.conv2d(g * sc, m.double(), padding=2, groups=grp) + g * sc
    else:                   # depthwise: one input channel per output channel, the scale is an output scale
        got_s = F.conv2d(g, m.double(), padding=2, groups=grp) * sc
    assert (got_s - ref_s).abs().max().item() <= 2e-6 * max(1.0, ref_s.abs().max().item())
