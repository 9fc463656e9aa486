"""numpy emulation of the gfx950 kernels' data-flow conventions (runs on CPU, no GPU needed).

Each function follows the indexing of the HIP kernel it names (csrc/*.hip) step by step -- MFMA lane/slot layout,
tap-offset tables, D-register -> channel maps, storage positions -- but in plain numpy on small inputs.  Feeding it the
host-prepared weights (shiftnet_amd/prep.py) and comparing with the oracle proves that the HOST packing and the KERNEL
conventions agree, independently of a GPU.  (bf16 rounding is not emulated: everything here is fp32.)
"""
import numpy as np
import torch


def frag_to_np(wfrag):
    return wfrag.float().numpy()


def mfma_tiles(wfrag, bfrag):
    """wfrag [MT,KS,64,8], bfrag [KS,64,8] (one N-tile) -> D regs [MT, 64 lanes, 4]."""
    MT, KS = wfrag.shape[:2]
    A = wfrag.reshape(MT, KS, 4, 16, 8)          # [mt, s, g, m, j]   lane = g*16 + m
    B = bfrag.reshape(KS, 4, 16, 8)              # [s, g, n, j]       lane = g*16 + n
    D = np.einsum("tsgmj,sgnj->tmn", A, B)       # [mt, m, n]
    regs = np.zeros((MT, 64, 4), np.float32)
    for lane in range(64):
        gD, n = lane >> 4, lane & 15
        for r in range(4):
            regs[:, lane, r] = D[:, gD * 4 + r, n]
    return regs


def conv2d(ins, p, stride=1, pad=None, bias=True, prelu=None, res=None, in_up=False):
    """sn_conv2d: ins list of [T,H,W,Cs] fp32 arrays (NHWC, padded), p = prep.pack_conv dict -> [T,Ho,Wo,16*MT]."""
    wfrag = frag_to_np(p["wfrag"]); MT, KS = wfrag.shape[:2]
    k = p["k"]; cs = p["cs_in"]; n_in = len(ins); cv = n_in * cs
    pad = k // 2 if pad is None else pad
    x = np.concatenate(ins, axis=-1)              # per-pixel interleave [.., n_in*cs]
    if in_up:
        raise NotImplementedError
    T, H, W, _ = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xp = np.zeros((T, H + 2 * pad + 64, W + 2 * pad + 64, cv), np.float32)
    xp[:, pad:pad + H, pad:pad + W] = x
    K = k * k * cv
    out = np.zeros((T, Ho, Wo, 16 * MT), np.float32)
    bvec = p["bias"].numpy() if (bias and p["bias"] is not None) else np.zeros(16 * MT, np.float32)
    for t in range(T):
        for oy in range(Ho):
            for ox0 in range(0, Wo, 16):
                bfrag = np.zeros((KS, 64, 8), np.float32)
                for lane in range(64):
                    g, pp = lane >> 4, lane & 15
                    ox = min(ox0 + pp, Wo - 1)
                    for s in range(KS):
                        kk0 = s * 32 + g * 8
                        if kk0 < K:
                            tap, cc0 = divmod(kk0, cv); dy, dx = divmod(tap, k)
                        else:
                            dy = dx = cc0 = 0
                        bfrag[s, lane] = xp[t, oy * stride + dy, ox * stride + dx, cc0:cc0 + 8]
                regs = mfma_tiles(wfrag, bfrag)
                for lane in range(64):
                    g, pp = lane >> 4, lane & 15
                    if ox0 + pp >= Wo:
                        continue
                    for mt in range(MT):
                        co0 = g * 4 * MT + mt * 4
                        out[t, oy, ox0 + pp, co0:co0 + 4] = regs[mt, lane] + bvec[co0:co0 + 4]
    if prelu is not None:
        out = np.where(out >= 0, out, out * prelu)
    if res is not None:
        out[..., :res.shape[-1]] += res
    return out


def unit_slabs(T, C, t, mode, wrap):
    Ch = C // 2
    f0, o0, f1, o1, fb, ob = t, 0, t, Ch, t, 0
    if mode == 1:
        if t > 0 or wrap:
            f0, o0, f1, o1 = (t - 1) % T, Ch, t, 0; fb, ob = f0, Ch
        else:
            fb, ob = t, 0
    elif mode == 2:
        if t < T - 1 or wrap:
            f0, o0, f1, o1 = t, Ch, (t + 1) % T, 0; fb, ob = f1, 0
        else:
            fb, ob = t, Ch
    return f0, o0, f1, o1, fb, ob


def shiftconv(x, offs, w1, mode, wrap):
    """sn_gsts_shiftconv: x [T,h,w,C] -> hw [T,h,w,C/2]."""
    T, h, w, C = x.shape; Ch = C // 2
    out = np.zeros((T, h, w, Ch), np.float32)
    for t in range(T):
        *_, fb, ob = unit_slabs(T, C, t, mode, wrap)
        src = np.zeros((h + 20, w + 20, Ch), np.float32)
        src[10:10 + h, 10:10 + w] = x[fb, :, :, ob:ob + Ch]
        for k in range(Ch):
            dy, dx = int(offs[k][0]), int(offs[k][1])
            for ty in range(3):
                for tx in range(3):
                    m = np.zeros((h, w), np.float32)
                    ys = np.arange(h) + ty - 1; xs = np.arange(w) + tx - 1
                    m[np.ix_((ys >= 0) & (ys < h), (xs >= 0) & (xs < w))] = 1
                    sh = src[10 + dy + ty - 1:10 + dy + ty - 1 + h, 10 + dx + tx - 1:10 + dx + tx - 1 + w, k]
                    out[t, :, :, k] += w1[k, ty * 3 + tx] * m * sh
    return out


def shiftconv_mfma(x, offs, w1, mode, wrap):
    """sn_gsts_shiftconv_mfma (csrc/sn_gsts.hip: shiftconv_mfma_kernel) as the kernel addresses it: per 16 x 16 tile a channel-planar 34 x 34 window
    (origin -9, zero outside the image), per channel three k-steps ty of a 16 x 16 x 32 MFMA whose A fragment a lane builds from the nine weights with
    its selector (element j = 8 g + i of row m = lane & 15: w[ty][j - m] for 0 <= j - m <= 2) and whose B fragment is 8 consecutive window pixels
    of row n + ty + 8 + dy from column 8 g + 8 + dx (g = 2: two pixels, g = 3: none), masked where the UNSHIFTED tap position lies outside the image;
    D[4 g + r][n] = out[y0 + n][x0 + 4 g + r].  x [T,h,w,C] -> hw [T,h,w,C/2]."""
    T, h, w, C = x.shape; Ch = C // 2
    out = np.zeros((T, h, w, Ch), np.float32)
    for t in range(T):
        *_, fb, ob = unit_slabs(T, C, t, mode, wrap)
        for y0 in range(0, h, 16):
            for x0 in range(0, w, 16):
                win = np.zeros((Ch, 34, 48), np.float32)                      # planar window (+ slack columns the masked lanes may touch)
                for ry in range(34):
                    for rx in range(34):
                        gy, gx = y0 - 9 + ry, x0 - 9 + rx
                        if 0 <= gy < h and 0 <= gx < w:
                            win[:, ry, rx] = x[fb, gy, gx, ob:ob + Ch]
                for k in range(Ch):
                    dy, dx = int(offs[k][0]), int(offs[k][1])
                    assert dx % 4 == 0 and abs(dy) <= 8 and abs(dx) <= 8
                    A = np.zeros((3, 16, 32), np.float32); B = np.zeros((3, 32, 16), np.float32)
                    for ty in range(3):
                        for lane in range(64):
                            g, m = lane >> 4, lane & 15                        # A: row m; B: column n = m
                            for i in range(8):
                                j = 8 * g + i
                                tx = j - m
                                A[ty, m, j] = w1[k, ty * 3 + tx] if 0 <= tx <= 2 else 0.0
                                if g == 3 or (g == 2 and i >= 2):
                                    continue                                   # bmask: no such slots
                                qy, qx = y0 + m + ty - 1, x0 + j - 1          # the tap's position before the shift: the conv's zero padding
                                if 0 <= qy < h and 0 <= qx < w:
                                    B[ty, j, m] = win[k, m + ty + 8 + dy, 8 * g + 8 + dx + i]
                    D = sum(A[ty] @ B[ty] for ty in range(3))                   # D[row = output column][col = output row]
                    for n in range(16):
                        for mm in range(16):
                            if y0 + n < h and x0 + mm < w:
                                out[t, y0 + n, x0 + mm, k] = D[mm, n]
    return out


def ln_gemm(x, hwb, wfrag, bias, mode, wrap):
    """sn_ln_gemm: -> a [T,h,w,2C] in storage-position order."""
    wfrag = frag_to_np(wfrag); MT, KS = wfrag.shape[:2]
    T, h, w, C = x.shape; Ch = C // 2
    K = C + Ch if hwb is not None else C
    a = np.zeros((T, h * w, 2 * C), np.float32)
    xf = x.reshape(T, h * w, C)
    hf = hwb.reshape(T, h * w, Ch) if hwb is not None else None
    for t in range(T):
        f0, o0, f1, o1, _, _ = unit_slabs(T, C, t, mode, wrap)
        for i0 in range(0, h * w, 16):
            bfrag = np.zeros((KS, 64, 8), np.float32)
            for p in range(16):
                i = min(i0 + p, h * w - 1)
                u = np.concatenate([xf[f0, i, o0:o0 + Ch], xf[f1, i, o1:o1 + Ch]] + ([hf[t, i]] if hf is not None else []))
                mean = u.mean(); var = ((u - mean) ** 2).mean()
                v = (u - mean) / np.sqrt(var + 1e-6)
                v = np.concatenate([v, np.zeros(KS * 32 - K, np.float32)])
                for g in range(4):
                    for s in range(KS):
                        bfrag[s, g * 16 + p] = v[s * 32 + g * 8: s * 32 + g * 8 + 8]
            regs = mfma_tiles(wfrag, bfrag)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                if i0 + p >= h * w:
                    continue
                for mt in range(MT):
                    pos = g * 4 * MT + mt * 4
                    a[t, i0 + p, pos:pos + 4] = regs[mt, lane] + bias[pos:pos + 4]
    return a.reshape(T, h, w, 2 * C)


def dw_gate(a, wdw):
    """sn_dw_gate: a [T,h,w,2C] positions, wdw [9][2C] -> g1 [T,h,w,C] natural."""
    T, h, w, C2 = a.shape
    ap = np.zeros((T, h + 2, w + 2, C2), np.float32); ap[:, 1:-1, 1:-1] = a
    o = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            o += wdw[dy * 3 + dx][None, None, None, :] * ap[:, dy:dy + h, dx:dx + w]
    o = o.reshape(T, h, w, C2 // 8, 8)
    return (o[..., :4] * o[..., 4:]).reshape(T, h, w, C2 // 2)


def dw_gemm_gate(g1, w5, wfrag, ca_in=None):
    """sn_dw_gemm_gate -> (g2 [T,h,w,C], channel sums [T,C])."""
    wfrag = frag_to_np(wfrag); MT, KS = wfrag.shape[:2]
    T, h, w, C = g1.shape
    if ca_in is not None:
        g1 = g1 * ca_in[:, None, None, :]
    gp = np.zeros((T, h + 4, w + 4, C), np.float32); gp[:, 2:-2, 2:-2] = g1
    r = np.zeros_like(g1)
    for dy in range(5):
        for dx in range(5):
            r += w5[dy * 5 + dx][None, None, None, :] * gp[:, dy:dy + h, dx:dx + w]
    rf = r.reshape(T, h * w, C)
    g2 = np.zeros((T, h * w, C), np.float32)
    for t in range(T):
        for i0 in range(0, h * w, 16):
            bfrag = np.zeros((KS, 64, 8), np.float32)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                i = min(i0 + p, h * w - 1)
                for s in range(KS):
                    kk0 = s * 32 + g * 8
                    if kk0 < C:
                        bfrag[s, lane] = rf[t, i, kk0:kk0 + 8]
            regs = mfma_tiles(wfrag, bfrag)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                if i0 + p >= h * w:
                    continue
                for mp in range(MT // 2):
                    b1, b2 = regs[2 * mp, lane], regs[2 * mp + 1, lane]
                    c0 = g * 2 * MT + mp * 4
                    g2[t, i0 + p, c0:c0 + 4] = b1 / (1.0 + np.exp(-b2))
    return g2.reshape(T, h, w, C), g2.sum(1)


def scale_gemm_res(x, g2, ca, wfrag, bias, mode, wrap):
    """sn_scale_gemm_res -> y [T,h,w,C]."""
    wfrag = frag_to_np(wfrag); MT, KS = wfrag.shape[:2]
    T, h, w, C = x.shape; Ch = C // 2
    xf = x.reshape(T, h * w, C); gf = g2.reshape(T, h * w, C)
    y = np.zeros((T, h * w, C), np.float32)
    for t in range(T):
        f0, o0, f1, o1, _, _ = unit_slabs(T, C, t, mode, wrap)
        for i0 in range(0, h * w, 16):
            bfrag = np.zeros((KS, 64, 8), np.float32)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                i = min(i0 + p, h * w - 1)
                for s in range(KS):
                    kk0 = s * 32 + g * 8
                    if kk0 < C:
                        bfrag[s, lane] = gf[t, i, kk0:kk0 + 8] * ca[t, kk0:kk0 + 8]
            regs = mfma_tiles(wfrag, bfrag)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                i = i0 + p
                if i >= h * w:
                    continue
                for mt in range(MT):
                    co0 = g * 4 * MT + mt * 4
                    sc = xf[f0, i, o0 + co0:o0 + co0 + 4] if co0 < Ch else xf[f1, i, o1 + co0 - Ch:o1 + co0 - Ch + 4]
                    b = bias[co0:co0 + 4] if bias is not None else 0.0
                    y[t, i, co0:co0 + 4] = sc + regs[mt, lane] + b
    return y.reshape(T, h, w, C)


def toeplitz_frag(rec, m, g, k):
    """A fragment (8 values) of lane (m, g) from one band record [2][20], built the way dw5m_gemm_gate_kernel / ln_gemm_gate_m
    do it: only the first (W) and last (X) dword of the lane's window are read, dwords 1 and 2 come from neighbouring lanes
    (DPP row shifts with the opposite-edge fallback).  k = stencil size (window start s0 = k//2 - 1 - m + 8g)."""
    flat = rec.reshape(40)

    def s0_of(mm):
        return k // 2 - 1 - mm + 8 * g

    def X(mm):                                   # elements (s0+6, s0+7) of lane mm, zero when out of the band's reach
        s0 = s0_of(mm)
        if s0 < 0 or s0 > 5:
            return flat[0:2]
        tw = 10 + (s0 + 5) // 2 if s0 & 1 else (s0 + 6) // 2
        return flat[2 * tw:2 * tw + 2]

    def W(mm):                                   # elements (s0, s0+1)
        s0 = s0_of(mm)
        if s0 < 6 or s0 > 11:
            return flat[0:2]
        tw = 10 + (s0 - 1) // 2 if s0 & 1 else s0 // 2
        return flat[2 * tw:2 * tw + 2]
    d0, d3 = W(m), X(m)
    d1 = W(m - 2) if m >= 2 else X(m + 4)        # row_shr:2, lanes 0..1 keep row_shl:4 of X
    d2 = X(m + 2) if m <= 13 else W(m - 4)       # row_shl:2, lanes 14..15 keep row_shr:4 of W
    return np.concatenate([d0, d1, d2, d3])


def dw5m_gemm_gate(g1, ttab, wfrag, ca_in=None):
    """sn_dw5m_gemm_gate (TH = 8 shape) -> (g2 [T,h,w,C], channel sums [T,C]).  g1 is given NHWC here; the planar image
    [channel][row][x] of one 64 x 8 tile (+2 rows, columns x0-8 .. x0+71) is what the Toeplitz MFMAs read."""
    wfrag = frag_to_np(wfrag); MT, KS = wfrag.shape[:2]
    tab = ttab.float().numpy()                                       # [C][5][2][20]
    T, h, w, C = g1.shape
    TW, TH = 64, 8
    g2 = np.zeros((T, h, w, C), np.float32)
    A = np.zeros((C, 5, 16, 32), np.float32)
    for c in range(C):
        for dy in range(5):
            for m in range(16):
                for g in range(4):
                    A[c, dy, m, 8 * g:8 * g + 8] = toeplitz_frag(tab[c, dy], m, g, 5)
    for t in range(T):
        for y0 in range(0, h, TH):
            for x0 in range(0, w, TW):
                img = np.zeros((C, TH + 4, 80), np.float32)              # rows y0-2.., columns x0-8..
                for ry in range(TH + 4):
                    gy = y0 - 2 + ry
                    if gy < 0 or gy >= h:
                        continue
                    for rx in range(80):
                        gx = x0 - 8 + rx
                        if 0 <= gx < w:
                            img[:, ry, rx] = g1[t, gy, gx]
                r = np.zeros((TH * TW, C), np.float32)                   # pair planes in the kernel; plain [px][ch] here
                for c in range(C):
                    sc = 1.0 if ca_in is None else ca_in[t, c]
                    for row in range(TH):
                        for xt in range(4):
                            acc = sum(A[c, dy] @ img[c, row + dy, 16 * xt:16 * xt + 32] for dy in range(5))
                            r[row * TW + 16 * xt:row * TW + 16 * xt + 16, c] = acc * sc
                for nt in range(TH * TW // 16):
                    bfrag = np.zeros((KS, 64, 8), np.float32)
                    for lane in range(64):
                        g, p = lane >> 4, lane & 15
                        for s in range(KS):
                            bfrag[s, lane] = r[nt * 16 + p, 32 * s + 8 * g:32 * s + 8 * g + 8]
                    regs = mfma_tiles(wfrag, bfrag)
                    for lane in range(64):
                        g, p = lane >> 4, lane & 15
                        tp = nt * 16 + p
                        oy, ox = y0 + tp // TW, x0 + tp % TW
                        if oy >= h or ox >= w:
                            continue
                        for mp in range(MT // 2):
                            b1, b2 = regs[2 * mp, lane], regs[2 * mp + 1, lane]
                            c0 = g * 2 * MT + mp * 4
                            g2[t, oy, ox, c0:c0 + 4] = b1 / (1.0 + np.exp(-b2))
    return g2, g2.reshape(T, h * w, C).sum(1)


def _h2(words):
    """uint32 words -> (lo, hi) fp16 values as float32."""
    w = np.asarray(words).astype(np.uint32)
    lo = (w & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
    hi = (w >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    return lo, hi


def cab_phase1r(x, hwb, pk, mode, wrap, ca_in=None, want_g1_sums=False):
    """Role-split fused phase 1 (csrc/sn_phase1r.hip) on whole frames, C = 64 / 80: the stager's two-pass LayerNorm + constant-one bias slots,
    first 1x1 with wave-paired rows (wave q, lane group g, register r <-> channel 16 q + 4 g + r), packed-fp16 3x3 table addressed by
    (wave, lane group, tap, word), RepConv as the x-pair Toeplitz GEMM (row = oc + 8 xp, k-slot -> tap through prep.p1r_tap), fp16 second
    1x1.  Strips, rings and lane <-> pixel maps are index arithmetic of the kernel and not emulated; everything the HOST prepares
    (prep.pack_phase1r) is decoded exactly as the kernel addresses it.  x [T,h,w,C], hwb [T,h,w,C/2] or None -> (g2 [T,h,w,C], sums [T,C]).
    Denoisers (sn_phase1_opts): want_g1_sums -> the channel sums [T,C] of g1 (pass 1); ca_in [T,C] -> g1 scaled by it before the RepConv (pass 2)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shift-net_amd"))
    from shiftnet_amd import prep
    w1 = frag_to_np(pk["wfrag1"]); MT, KS = w1.shape[:2]
    w2 = pk["wfrag2"].float().numpy(); KS2 = w2.shape[1]
    t3 = pk["w3"].numpy().view(np.uint32)
    wg = pk["wgrp"].float().numpy()                       # [NGP][2][8][64][8]
    T, h, w, C = x.shape; Ch = C // 2; NGP = C // 16
    K = C + Ch if hwb is not None else C
    assert MT == 2 * NGP and KS == (K + 2 + 31) // 32
    bf = lambda v: torch.tensor(v, dtype=torch.float32).to(torch.bfloat16).float().numpy()      # noqa: E731
    xf = x.reshape(T, h * w, C)
    hf = hwb.reshape(T, h * w, Ch) if hwb is not None else None
    a = np.zeros((T, h * w, 2 * C), np.float32)
    for t in range(T):
        f0, o0, f1, o1, _, _ = unit_slabs(T, C, t, mode, wrap)
        for i0 in range(0, h * w, 16):
            bfrag = np.zeros((KS, 64, 8), np.float32)
            for p in range(16):
                i = min(i0 + p, h * w - 1)
                u = np.concatenate([xf[f0, i, o0:o0 + Ch], xf[f1, i, o1:o1 + Ch]] + ([hf[t, i]] if hf is not None else []))
                mean = np.float32(u.sum() / K); d = (u - mean).astype(np.float32)
                rstd = np.float32(1.0) / np.sqrt(np.float32((d * d).sum() / K) + np.float32(1e-6))
                kv = np.zeros(KS * 32, np.float32)
                kv[:K] = bf(d * rstd); kv[K] = 1.0; kv[K + 1] = 1.0
                for g in range(4):
                    for s in range(KS):
                        bfrag[s, g * 16 + p] = kv[s * 32 + g * 8: s * 32 + g * 8 + 8]
            regs = mfma_tiles(w1, bfrag)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                if i0 + p >= h * w:
                    continue
                for q in range(NGP):
                    for half in range(2):
                        c0 = half * C + 16 * q + 4 * g
                        a[t, i0 + p, c0:c0 + 4] = regs[2 * q + half, lane].astype(np.float16).astype(np.float32)
    a = a.reshape(T, h, w, 2 * C)
    k3 = np.zeros((2 * C, 9), np.float32)
    for q in range(NGP):
        for g in range(4):
            for k in range(4):
                o = (k >> 1) * C + 16 * q + 4 * g + 2 * (k & 1)
                lo, hi = _h2(t3[q, g, :, k])
                k3[o] = lo; k3[o + 1] = hi
    ap = np.zeros((T, h + 2, w + 2, 2 * C), np.float32); ap[:, 1:-1, 1:-1] = a
    o = np.zeros_like(a)
    for ty in range(3):
        for tx in range(3):
            o += k3[:, ty * 3 + tx][None, None, None, :] * ap[:, ty:ty + h, tx:tx + w]
    o = o.astype(np.float16).astype(np.float32)
    g1 = (o[..., :C] * o[..., C:]).astype(np.float16).astype(np.float32)          # carries P1_G1_SCALE
    if want_g1_sums:
        return g1.reshape(T, h * w, C).sum(1) * 16.0
    if ca_in is not None:
        g1 = (g1 * np.asarray(ca_in, np.float32).astype(np.float16).astype(np.float32)[:, None, None, :]).astype(np.float16).astype(np.float32)
    # RepConv: pairs of pixels (cx, cx + 1), cx = -3, -1, 1, ... (region column 0 is image column x0 - 3)
    npair = (w + 3 + 1) // 2 + 1
    gp = np.zeros((T, h + 4, 2 * npair + 8, C), np.float32)
    X0 = 5                                                    # gp column of image column 0:  gp col = gx + X0, pair i starts at gx = 2 i - 3 -> gp col 2 i + 2
    gp[:, 2:2 + h, X0:X0 + w] = g1
    r = np.zeros((T, h, 2 * npair + 8, C), np.float32)
    for grp in range(C // 8):
        fr = wg[grp // 2, grp % 2].reshape(8, 4, 16, 8)       # [s][gq][m][j]
        for s in range(8):
            for gq in range(4):
                dy, dx6 = prep.p1r_tap(s, gq)
                if dy < 0:
                    assert not fr[s, gq].any()
                    continue
                for i in range(npair):
                    cx = 2 * i + 2                             # gp column of the pair's first pixel
                    src = gp[:, dy:dy + h, cx + dx6 - 2, 8 * grp:8 * grp + 8]           # [T,h,8 j]
                    contrib = np.einsum("mj,thj->thm", fr[s, gq], src)                  # [T,h,16 m]
                    r[:, :, cx, 8 * grp:8 * grp + 8] += contrib[..., :8]
                    r[:, :, cx + 1, 8 * grp:8 * grp + 8] += contrib[..., 8:]
    rf = r[:, :, X0:X0 + w].astype(np.float16).astype(np.float32).reshape(T, h * w, C)
    g2 = np.zeros((T, h * w, C), np.float32)
    for t in range(T):
        for i0 in range(0, h * w, 16):
            bfrag = np.zeros((KS2, 64, 8), np.float32)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                i = min(i0 + p, h * w - 1)
                for s in range(KS2):
                    kk0 = s * 32 + g * 8
                    if kk0 < C:
                        bfrag[s, lane] = rf[t, i, kk0:kk0 + 8]
                    else:
                        bfrag[s, lane] = 7.0                  # the kernel reads the next pixel's values there: the weights must be zero
            regs = mfma_tiles(w2, bfrag)
            for lane in range(64):
                g, p = lane >> 4, lane & 15
                if i0 + p >= h * w:
                    continue
                for q in range(NGP):
                    b1, b2 = regs[2 * q, lane], regs[2 * q + 1, lane]
                    c0 = 16 * q + 4 * g
                    g2[t, i0 + p, c0:c0 + 4] = b1 / (1.0 + np.exp2(b2))
    return g2.reshape(T, h, w, C), g2.sum(1)
