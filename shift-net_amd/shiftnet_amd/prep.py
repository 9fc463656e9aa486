"""Host-side weight preparation: checkpoint tensors -> the layouts the gfx950 kernels consume.

All channel permutations of the HIP path live HERE, the kernels only ever see "positions":

* MFMA A fragments ``[MT][KS][64][8]`` bf16 for ``v_mfma_f32_16x16x32_bf16`` with the weights as the A operand:
  lane ``l`` holds ``Wp[mt*16 + (l & 15)][ks*32 + (l >> 4)*8 + j]`` (csrc/sn_common.h).
* D registers: lane ``(g, p)`` reg ``r`` of M-tile ``mt`` is row ``mt*16 + g*4 + r``.  Rows are assigned to output
  channels so that a lane's registers are CONTIGUOUS channels of its pixel:
    - ``conv`` / ``out`` order:   channel = g*4*MT + mt*4 + r                       (natural NHWC store)
    - ``gate`` order (2C rows):   mt even -> channel c, mt odd -> its gate partner C + c, c = g*2*MT + (mt//2)*4 + r,
      so SimpleGate / SimpleGate2 are lane-local and the gated result comes out in natural order.
* identity / 3x3 branches of RepConv / RepConv2 are folded into one stencil (gshift_deblur1.py:157-174),
  LayerNorm affine into the following 1x1 (:27,190), beta (and the denoise bias) into the last 1x1 (:201,210).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def ceil8(c: int) -> int:
    return (c + 7) // 8 * 8


def pack_frag(wp: np.ndarray) -> torch.Tensor:
    """[16*MT, 32*KS] fp32 -> bf16 tensor [MT, KS, 64, 8] in A-fragment order."""
    m, k = wp.shape
    assert m % 16 == 0 and k % 32 == 0
    mt, ks = m // 16, k // 32
    a = wp.reshape(mt, 16, ks, 4, 8).transpose(0, 2, 3, 1, 4).reshape(mt, ks, 64, 8)
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)


def rows_natural(n_ch: int, mt: int) -> np.ndarray:
    """row index (into the 16*MT padded matrix) of output channel co, 'conv' order."""
    co = np.arange(n_ch)
    g, rem = co // (4 * mt), co % (4 * mt)
    return (rem // 4) * 16 + g * 4 + (rem % 4)


def rows_gate(c: int) -> np.ndarray:
    """row index of output channel o in [0, 2C) for the gate-paired order (MT = C/8 M-tiles)."""
    mt = c // 8
    o = np.arange(2 * c)
    half, cc = o // c, o % c
    g, rem = cc // (2 * mt), cc % (2 * mt)
    q, r = rem // 4, rem % 4
    return (2 * q + half) * 16 + g * 4 + r


def pos_gate(c: int) -> np.ndarray:
    """storage position (within a pixel's 2C values) of output channel o for the gate-paired order."""
    mt = c // 8
    o = np.arange(2 * c)
    half, cc = o // c, o % c
    g, rem = cc // (2 * mt), cc % (2 * mt)
    q, r = rem // 4, rem % 4
    return g * 4 * mt + (2 * q + half) * 4 + r


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], cins: Sequence[int], cs_in: int, shuffle: bool = False) -> Dict[str, object]:
    """Dense conv weight [Cout, sum(cins), k, k] -> implicit-GEMM fragments.

    K index = tap*(n_in*cs_in) + i*cs_in + c  (tap = ky*k + kx), rows in 'conv' order.
    shuffle: the conv of a PixelShufflePack (gshift_deblur1.py:335-350; sn_conv2d out_mode 1).  nn.PixelShuffle(2) sends conv channel 4 c + 2 i + j to
    output channel c of sub-pixel (i, j); the rows are re-ordered to [sub-pixel 2 i + j][c < cs_out] (zero rows for the storage padding), so the four
    consecutive rows a lane owns are four consecutive OUTPUT channels of ONE sub-pixel: one 8-byte store instead of four 2-byte ones ('cout' stays the
    reference's 4 x c_out, 'mt' covers the 4 x cs_out permuted rows).
    """
    w = weight.detach().float().cpu().numpy()
    if shuffle:
        c_log = w.shape[0] // 4
        cs_out = ceil8(c_log)
        ws = np.zeros((4 * cs_out,) + w.shape[1:], np.float32)
        bs = np.zeros(4 * cs_out, np.float32)
        bn = None if bias is None else bias.detach().float().cpu().numpy()
        for sub in range(4):
            ws[sub * cs_out:sub * cs_out + c_log] = w[sub::4]
            if bn is not None:
                bs[sub * cs_out:sub * cs_out + c_log] = bn[sub::4]
        p = pack_conv(torch.from_numpy(ws), None if bias is None else torch.from_numpy(bs), cins, cs_in)
        p["cout"] = w.shape[0]
        return p
    cout, cin_tot, k, _ = w.shape
    assert cin_tot == sum(cins)
    n_in = len(cins)
    cv = n_in * cs_in
    mt = (cout + 15) // 16
    kdim = k * k * cv
    ks = (kdim + 31) // 32
    wp = np.zeros((16 * mt, 32 * ks), np.float32)
    rows = rows_natural(cout, mt)
    base = 0
    for i, ci in enumerate(cins):
        for tap in range(k * k):
            ky, kx = divmod(tap, k)
            cols = tap * cv + i * cs_in + np.arange(ci)
            wp[np.ix_(rows, cols)] = w[:, base:base + ci, ky, kx]
        base += ci
    b = None
    if bias is not None:
        b = np.zeros(16 * mt, np.float32)
        b[:cout] = bias.detach().float().cpu().numpy()
    return {"wfrag": pack_frag(wp), "mt": mt, "ks": ks, "bias": None if b is None else torch.from_numpy(b),
            "k": k, "cout": cout, "n_in": n_in, "cs_in": cs_in}


def pack_ln_gemm(w1: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, c: int) -> Dict[str, object]:
    """body[0] (1x1, 2C x K) with the LayerNorm affine folded in; rows/bias in gate-paired order."""
    w = w1.detach().float().cpu().numpy().reshape(2 * c, -1)
    kdim = w.shape[1]
    lw = ln_w.detach().float().cpu().numpy()
    lb = ln_b.detach().float().cpu().numpy()
    wf = w * lw[None, :]
    bf = w @ lb
    ks = (kdim + 31) // 32
    mt = c // 8
    wp = np.zeros((16 * mt, 32 * ks), np.float32)
    wp[rows_gate(c), :kdim] = wf
    bias = np.zeros(2 * c, np.float32)
    bias[pos_gate(c)] = bf
    return {"wfrag": pack_frag(wp), "bias": torch.from_numpy(bias)}


def pack_dw3_gate(w: torch.Tensor, c: int) -> torch.Tensor:
    """RepConv2 depthwise 3x3 [2C,1,3,3] (+identity) -> fp32 [9][2C] in a's storage-position order."""
    wn = w.detach().float().cpu().numpy().reshape(2 * c, 9).copy()
    wn[:, 4] += 1.0
    out = np.zeros((9, 2 * c), np.float32)
    out[:, pos_gate(c)] = wn.T
    return torch.from_numpy(out)


def pack_dw5(w5: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """depthwise RepConv: conv_1 [C,1,5,5] + conv_2 [C,1,3,3] + identity -> fp32 [25][C]."""
    a = w5.detach().float().cpu().numpy()[:, 0].copy()
    a[:, 1:4, 1:4] += w3.detach().float().cpu().numpy()[:, 0]
    a[:, 2, 2] += 1.0
    return torch.from_numpy(np.ascontiguousarray(a.reshape(a.shape[0], 25).T))


def pack_gate_gemm(w2: torch.Tensor, c: int) -> torch.Tensor:
    """body 1x1 C -> 2C before SimpleGate2; rows gate-paired, K natural."""
    w = w2.detach().float().cpu().numpy().reshape(2 * c, c)
    ks = (c + 31) // 32
    wp = np.zeros((16 * (c // 8), 32 * ks), np.float32)
    wp[rows_gate(c), :c] = w
    return pack_frag(wp)


def pack_out_gemm(w3: torch.Tensor, beta: torch.Tensor, bias3: Optional[torch.Tensor], c: int) -> Dict[str, object]:
    """last 1x1 C -> C with beta (and beta*bias) folded; rows in natural 'conv' order (MT = C/16)."""
    w = w3.detach().float().cpu().numpy().reshape(c, c)
    b = beta.detach().float().cpu().numpy().reshape(c)
    mt = c // 16
    ks = (c + 31) // 32
    wp = np.zeros((16 * mt, 32 * ks), np.float32)
    wp[rows_natural(c, mt), :c] = w * b[:, None]
    bias = None
    if bias3 is not None:
        bias = torch.from_numpy((bias3.detach().float().cpu().numpy() * b).astype(np.float32))
    return {"wfrag": pack_frag(wp), "bias": bias}


def shift_offsets_i8(table: List[Tuple[int, int]]) -> torch.Tensor:
    return torch.tensor(table, dtype=torch.int8).reshape(-1, 2).contiguous()


def pk_f16_words(w: torch.Tensor) -> torch.Tensor:
    """fp32 [..., N] (N even) -> int32 [..., N/2] words for v_pk_fma_f16: fp16(w[..., 2k]) in the low half, fp16(w[..., 2k+1]) in the
    high half of word k (round to nearest)."""
    h = w.to(torch.float16).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    return (h[..., 0::2] | (h[..., 1::2] << 16)).to(torch.int32).contiguous()


def pack_grouped_frag(w5: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """grouped RepConv ("+", 8 in / 8 out per group): fold 3x3 + identity into the 5x5 and lay it out for the
    block-diagonal MFMA of sn_grp5_gemm_gate: M-tile mt = output channels [16mt, 16mt+16) (two groups), k-step s =
    taps 2s, 2s+1, slot (g, j) = tap 2s + (g >> 1), input channel 16mt + (g & 1)*8 + j; a row only sees its own group."""
    a = w5.detach().float().cpu().numpy().copy()          # [C, 8, 5, 5]
    c = a.shape[0]
    a[:, :, 1:4, 1:4] += w3.detach().float().cpu().numpy()
    for o in range(c):
        a[o, o % 8, 2, 2] += 1.0
    mtg, ksg = c // 16, 13
    wp = np.zeros((16 * mtg, 32 * ksg), np.float32)
    for mt in range(mtg):
        for m16 in range(16):
            o = 16 * mt + m16
            for s in range(ksg):
                for g in range(4):
                    tap = 2 * s + (g >> 1)
                    if tap >= 25 or (g & 1) != (m16 >> 3):
                        continue
                    wp[16 * mt + m16, s * 32 + g * 8: s * 32 + g * 8 + 8] = a[o, :, tap // 5, tap % 5]
    return pack_frag(wp)


def pack_toeplitz(wk: torch.Tensor, k: int) -> torch.Tensor:
    """depthwise k x k stencil [k*k][C] fp32 -> padded bands bf16 [C][k][2][20] for the Toeplitz MFMAs (csrc/sn_gsts3.hip).

    The A operand of kernel row dy is A[m][kk] = w[dy][kk - m - (8 - k//2)] (16 outputs x 32 input columns, the window
    starts 8 columns left of the tile).  With the band padded to P = [0]*7 + w[dy] + [0]*(13-k) (20 values), lane (m, g)
    needs A[m][8g .. 8g+7] = P[s0 .. s0+7], s0 = k//2 - 1 - m + 8g (an all-zero window, s0 = 12, when out of range).
    Copy 0 is P, copy 1 is P shifted left by one so that odd starts are dword-aligned too."""
    w = wk.detach().float().cpu().numpy()
    c = w.shape[1]
    w = w.T.reshape(c, k, k)
    tab = np.zeros((c, k, 2, 20), np.float32)
    tab[:, :, 0, 7:7 + k] = w
    tab[:, :, 1, 6:6 + k] = w
    return torch.from_numpy(tab).to(torch.bfloat16)


# ---- fused phase 1 (csrc/sn_phase1.hip): LayerNorm after the GEMM, depthwise stencils on the MFMA accumulator layout -----------------
P1_G1_SCALE = 2.0 ** -4      # g1 = a1' * a2' is carried in fp16: the first factor is scaled down, the second 1x1 scaled up (exact powers of two)


def pack_frag_f16(wp: np.ndarray) -> torch.Tensor:
    """pack_frag for v_mfma_f32_16x16x32_f16: the same A-fragment order, fp16 elements."""
    m, k = wp.shape
    assert m % 16 == 0 and k % 32 == 0
    mt, ks = m // 16, k // 32
    a = wp.reshape(mt, 16, ks, 4, 8).transpose(0, 2, 3, 1, 4).reshape(mt, ks, 64, 8)
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float16)


def _h2_words(lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """fp32 arrays -> uint32 words: fp16(lo) in bits 0..15, fp16(hi) in bits 16..31 (round to nearest)."""
    l16 = lo.astype(np.float16).view(np.uint16).astype(np.uint32)
    h16 = hi.astype(np.float16).view(np.uint16).astype(np.uint32)
    return l16 | (h16 << 16)


def pack_conv32_split(w: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """fp32 conv weight [co][ci/groups][k][k] -> bf16 A fragments [2 (hi | lo)][MT][KS][64][8] of the split-precision fp32 conv
    (csrc/sn_f32.hip: conv32s_kernel): hi = bf16(w), lo = bf16(w - hi); rows in natural order (M-tile m = output channels 16 m ..).
      dense   (groups == 1): k-step s = tap * ceil(ci / 32) + block, slot kk = input channel 32 block + kk (zero beyond ci);
      grouped (8 in / 8 out per group, k = 5 / 3): KS = 13 / 5, k-step s, slot 8 g + j = tap 2 s + (g >> 1), input channel 16 m + (g & 1) * 8 + j of the
              M-tile's own two groups; a row only sees its own group (block-diagonal), tap 25 does not exist (zeros)."""
    wn = w.detach().float().cpu().numpy()
    co, cig, k, _ = wn.shape
    ntap = k * k
    mt = (co + 15) // 16
    if groups == 1:
        ncb = (cig + 31) // 32
        ks = ntap * ncb
        wp = np.zeros((16 * mt, 32 * ks), np.float32)
        for tap in range(ntap):
            ty, tx = divmod(tap, k)
            for cb in range(ncb):
                c0, c1 = 32 * cb, min(32 * cb + 32, cig)
                s0 = (tap * ncb + cb) * 32
                wp[:co, s0:s0 + (c1 - c0)] = wn[:, c0:c1, ty, tx]
    else:
        assert cig == 8 and co % 16 == 0 and co // groups == 8 and k in (3, 5)
        ks = (ntap + 1) // 2
        wp = np.zeros((16 * mt, 32 * ks), np.float32)
        for m in range(mt):
            for r in range(16):
                o = 16 * m + r
                for s_ in range(ks):
                    for g in range(4):
                        tap = 2 * s_ + (g >> 1)
                        if tap >= ntap or (r >> 3) != (g & 1):
                            continue
                        ty, tx = divmod(tap, k)
                        wp[o, 32 * s_ + 8 * g: 32 * s_ + 8 * g + 8] = wn[o, :, ty, tx]
    hi = torch.from_numpy(wp).to(torch.bfloat16).float().numpy()
    lo = wp - hi
    return torch.stack([pack_frag(hi), pack_frag(lo)], 0).contiguous()


# ---- role-split fused phase 1 (csrc/sn_phase1r.hip): C = 64 / 80, depthwise or grouped RepConv on the matrix cores --------------------------
def rows_pair(c: int) -> np.ndarray:
    """row index (into the 2C-row padded matrix, M-tile 2q + half) of output channel o in [0, 2C) for the WAVE-paired order of the
    role-split kernel: wave q owns M-tiles (2q, 2q+1) = channels 16q .. 16q+15 and their gate partners C + 16q .., so that a wave's
    16 gated channels are two whole RepConv groups of 8.  Row inside the tile = channel - 16q (lane group g = 4 consecutive channels)."""
    o = np.arange(2 * c)
    half, cc = o // c, o % c
    return (2 * (cc // 16) + half) * 16 + cc % 16


def p1r_ks1(c: int, with_hw: bool) -> int:
    """k-steps of the first 1x1: K data slots + 2 bias slots (a constant-one operand, bias as bf16 hi + lo), rounded up to 32."""
    return ((c + c // 2 if with_hw else c) + 2 + 31) // 32


def rep_dense_group(w5: torch.Tensor, w3: torch.Tensor, c: int) -> np.ndarray:
    """RepConv (conv_1 5x5 + conv_2 3x3 + identity; gshift_deblur1.py:157-165 grouped [C,8,5,5], gshift_deblur2.py:159-168 depthwise [C,1,5,5])
    as per-group dense kernels [C/8 groups][8 oc][8 ic][5][5] (depthwise: diagonal)."""
    a5 = w5.detach().float().cpu().numpy()
    a3 = w3.detach().float().cpu().numpy()
    ng = c // 8
    out = np.zeros((ng, 8, 8, 5, 5), np.float32)
    if a5.shape[1] == 8:
        k = a5.copy()
        k[:, :, 1:4, 1:4] += a3
        out[:] = k.reshape(ng, 8, 8, 5, 5)
        for o in range(8):
            out[:, o, o, 2, 2] += 1.0
    else:
        assert a5.shape[1] == 1
        k = a5[:, 0].copy()
        k[:, 1:4, 1:4] += a3[:, 0]
        k[:, 2, 2] += 1.0
        k = k.reshape(ng, 8, 5, 5)
        for o in range(8):
            out[:, o, o] = k[:, o]
    return out


def p1r_tap(s: int, gq: int) -> Tuple[int, int]:
    """(dy, dx6) of k-step s, lane group gq of the x-pair Toeplitz RepConv, or (-1, -1) for the two unused slots.  dx6 in 0..5 is the input
    column relative to (pair's first pixel - 2): output position xp in {0, 1} of the pair sees tap dx = dx6 - xp.  Steps 0..5 hold rows
    dy = 0..3 (lane group = dy: the four groups read the SAME columns of four ring rows, whose pitch is a multiple of the 64 LDS banks, so
    the reads are conflict free), steps 6 / 7 row dy = 4 (lane group = dx6 resp. dx6 - 4)."""
    if s < 6:
        return gq, s
    if s == 6:
        return 4, gq
    return (4, 4 + gq) if gq < 2 else (-1, -1)


def pack_phase1r(w1: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, w_dw3: torch.Tensor, w_rep5: torch.Tensor, w_rep3: torch.Tensor,
                 w2: torch.Tensor, c: int) -> Dict[str, torch.Tensor]:
    """Operands of the role-split fused phase 1 (csrc/sn_phase1r.hip) for one CAB1 / CAB2, C in {64, 80}, RepConv grouped or depthwise.

    * wfrag1 bf16 [2 NGP][KS1][64][8] (NGP = C/16 wave pairs): body[0] with the LayerNorm scale folded, rows in rows_pair order; the kernel
      feeds the NORMALISED input (two-pass statistics by the stager waves) and a constant 1 in k-slots K, K+1, whose A columns carry the folded
      bias W ln_b as bf16 hi + lo.  Out-of-image pixels are all-zero operands (incl. the constant), so `a` is exactly 0 there: the zero
      padding of the 3x3.
    * w3 uint32 [NGP q][4 g][9 taps][4]: packed-fp16 taps of RepConv2 (+identity) for the lane's four packed registers: word k = channels
      (16q + 4g + 2k, +1) for k = 0, 1 (scaled by P1_G1_SCALE), their gate partners for k = 2, 3.
    * wgrp fp16 [NGP q][2 G][8 s][64][8]: RepConv of group 2q + G as the x-pair Toeplitz GEMM: row m = oc + 8 xp (output channel oc of the
      group at pixel xp of a pair), k-slot (gq, j) of step s = tap p1r_tap(s, gq), input channel j; value = w[oc][j][dy][dx6 - xp].
    * wfrag2 fp16 [2 NGP][KS2][64][8]: body[4] / P1_G1_SCALE, rows_pair order, natural K, the sigmoid rows times -log2(e)."""
    assert c in (64, 80)
    ngp = c // 16
    w = w1.detach().float().cpu().numpy().reshape(2 * c, -1)
    kdim = w.shape[1]
    ks1 = (kdim + 2 + 31) // 32
    wf = w * ln_w.detach().float().cpu().numpy()[None, :]
    b = w @ ln_b.detach().float().cpu().numpy()
    b_hi = torch.from_numpy(b.astype(np.float32)).to(torch.bfloat16).float().numpy()
    wp = np.zeros((32 * ngp, 32 * ks1), np.float32)
    rp = rows_pair(c)
    wp[rp, :kdim] = wf
    wp[rp, kdim] = b_hi
    wp[rp, kdim + 1] = b - b_hi
    d3 = w_dw3.detach().float().cpu().numpy().reshape(2 * c, 9).copy()
    d3[:, 4] += 1.0
    d3[:c] *= P1_G1_SCALE
    t3 = np.zeros((ngp, 4, 9, 4), np.uint32)
    for q in range(ngp):
        for g in range(4):
            c0 = 16 * q + 4 * g
            for k in range(4):
                o = (k >> 1) * c + c0 + 2 * (k & 1)
                t3[q, g, :, k] = _h2_words(d3[o], d3[o + 1])
    dg = rep_dense_group(w_rep5, w_rep3, c)                   # [C/8][oc][ic][dy][dx]
    wg = np.zeros((ngp, 2, 16, 8 * 32), np.float32)
    for q in range(ngp):
        for G in range(2):
            k5 = dg[2 * q + G]
            for s in range(8):
                for gq in range(4):
                    dy, dx6 = p1r_tap(s, gq)
                    if dy < 0:
                        continue
                    for xp in range(2):
                        dx = dx6 - xp
                        if 0 <= dx <= 4:
                            wg[q, G, 8 * xp: 8 * xp + 8, 32 * s + 8 * gq: 32 * s + 8 * gq + 8] = k5[:, :, dy, dx]
    wgf = torch.stack([torch.stack([pack_frag_f16(wg[q, G])[0] for G in range(2)]) for q in range(ngp)])     # [NGP][2][8][64][8]
    w2n = w2.detach().float().cpu().numpy().reshape(2 * c, c) / P1_G1_SCALE
    w2n[c:] *= -np.log2(np.e)
    ks2 = (c + 31) // 32
    wp2 = np.zeros((32 * ngp, 32 * ks2), np.float32)
    wp2[rp, :c] = w2n
    return {"wfrag1": pack_frag(wp), "w3": torch.from_numpy(t3.view(np.int32).copy()), "wgrp": wgf.contiguous(), "wfrag2": pack_frag_f16(wp2)}
