// Third-generation GSTS kernels (gfx950): depthwise stencils on the matrix cores.
//
// A depthwise k x k convolution has no channel contraction, so it cannot be a GEMM over channels.  It IS a GEMM over
// the x axis: for one channel c and one kernel row dy
//     out[c][y][x0 + m] += sum_k  A_{c,dy}[m][k] * in[c][y + dy - 2][x0 - 8 + k],   A_{c,dy}[m][k] = w[c][dy][k - m - 6]
// a 16 x 32 banded (Toeplitz) matrix times 32 consecutive input columns, and the 16 columns of the MFMA's N dimension
// are 16 different (x-tile, row-strip) places of the SAME channel.  One v_mfma_f32_16x16x32_bf16 therefore does 16 outputs
// x 16 places x 5 taps of real work (16 % of its flops), which is still ~6x the rate of the packed-VALU stencil
// (v_dot2c at 4.8 cycles per 64 taps), and the 5 x 5 window costs 5 MFMAs.  Requirements that shape the kernels:
//   * B operand = 8 consecutive x of one channel (16 B): the stencil input is channel-PLANAR, [T][h][C][wr] in HBM
//     (wr = w rounded up to 8, pad columns zero) and [channel][row][x] in LDS;
//   * A operand: lane (m, g) needs 8 consecutive values of the zero-padded band [0 x7, w0..w4, 0 x8] starting at element
//     1 - m + 8g; the band (and a copy shifted by one element, for odd starts) lives ONCE in LDS, 80 B per (c, dy)
//     (25 KB for 64 channels), and a fragment is four aligned ds_read_b32.
//
//   sn_dw5m_gemm_gate (K3m): g2 = SimpleGate2(body[4](RepConv(g1))) for C = 64, depthwise RepConv.  Persistent
//     1024-thread workgroups (one per CU, 4 waves per SIMD) walk 64 x 8 pixel tiles on an XCD-aware schedule.  Phase 1,
//     four steps of 16 channels: HBM -> registers (three steps ahead) -> LDS planar image, double buffered (one step
//     ahead, ONE barrier per step) -> Toeplitz MFMAs (wave = channel pair x half tile) -> r in LDS as channel-pair
//     planes [32][px] (one 16-byte store per lane and step).  Phase 2: the 1x1 C -> 2C on MFMA (wave = 64 pixels x half
//     of the M-tiles, A fragments resident in registers), SimpleGate2, NHWC stores, channel sums.
//   sn_nhwc_to_planar: layout change for A/B tests and for producers that still write NHWC.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

namespace {

constexpr int K3M_C = 64, K3M_TW = 64, K3M_RX = 80, K3M_CHK = 16;
// (Round 2 measured a conflict-free variant of the staged image -- 12-slot row pitch with the place map xt = p >> 2, rr = p & 3, the
// only combination in reach for which the 16 lanes of every ds_read_b128 group hit 16 distinct slots, band records streamed per chunk
// to make room: the Toeplitz phase got 27 % shorter, the barrier waits grew by the same amount, 464.3 vs 460.6 us at level 1 and
// 112.4 vs 110.3 us at level 2 on the same box.  The kernel is bound by the skew of its 225 lock-steps, not by the B reads; kept as is.)
constexpr int K3M_TAB_BYTES = K3M_C * 5 * 80;                          // 25600: [c][dy]{band padded to 20, same shifted by one}
// Tile 64 x 8: 1024 threads, one workgroup per CU, double-buffered staging image (one barrier per step).  (A 64 x 4 shape with two
// 512-thread workgroups per CU measured 432 vs 381 us in round 2: more halo rows and barriers than the interleaving wins back.)
template <int TH> struct K3mShape {
    static_assert(TH == 8, "one shape");
    static constexpr int RH = TH + 4, NWV = 2 * TH, NTHR = 64 * NWV, NBUF = 2;
    static constexpr int GIMG_BYTES = K3M_CHK * RH * K3M_RX * 2;       // 30720 per buffer
    static constexpr int RPITCH = K3M_TW * TH + 4;                     // dwords per channel-pair plane of r (lane groups g land 16 banks apart)
    static constexpr int R_BYTES = (K3M_C / 2) * RPITCH * 4;           // 66048
    static constexpr int RED_BYTES = NWV * 32 * 4;
    static constexpr int LDS = K3M_TAB_BYTES + NBUF * GIMG_BYTES + R_BYTES + RED_BYTES;      // 155136
};

template <int TH>
__global__ __launch_bounds__(K3mShape<TH>::NTHR, 1)
void dw5m_gemm_gate_kernel(const bf16_t* __restrict__ g1p, const float* __restrict__ ca_in, const uint32_t* __restrict__ ttab,
                           const uint4* __restrict__ wfrag, bf16_t* g2, float* pool, int T, int h, int w, int wr) {
    using SH = K3mShape<TH>;
    constexpr int C = K3M_C, TW = K3M_TW, RH = SH::RH, RX = K3M_RX, CHK = K3M_CHK, KS = 2, RP = SH::RPITCH;
    constexpr int NWV = SH::NWV, NTHR = SH::NTHR, NBUF = SH::NBUF, GIMG_BYTES = SH::GIMG_BYTES;
    constexpr int NITEMS = CHK * RH * (RX / 8), NIT = (NITEMS + NTHR - 1) / NTHR;     // staging items of 16 B per chunk: 1920 -> 2, 1280 -> 3 per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t* tab = (const uint32_t*)smem;
    char* gimg = smem + K3M_TAB_BYTES;                                 // [NBUF][CHK][RH][RX] bf16
    uint32_t* lds_r = (uint32_t*)(gimg + NBUF * GIMG_BYTES);           // [32 channel pairs][RP]: dword = (channel 2q, 2q+1) of one pixel
    float* red = (float*)((char*)lds_r + SH::R_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH, tpf = tiles_x * tiles_y, ntiles = T * tpf;
    const size_t frame = (size_t)h * C * wr;                 // elements per frame of the planar tensor

    for (int e = tid; e < K3M_TAB_BYTES / 4; e += NTHR) ((uint32_t*)smem)[e] = ttab[e];
    // phase 2 role of this wave: pixels of N-tiles 4 ng .. 4 ng + 3, M-tiles 4 mh .. 4 mh + 3 (gate pairs 2 mh, 2 mh + 1)
    const int ng = wv >> 1, mh = wv & 1;
    bf16x8_t A2[4][KS];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < KS; ++s) A2[m][s] = as_frag(wfrag[((4 * mh + m) * KS + s) * 64 + lane]);

    // ---- staging: item = (channel of the chunk, region row, 8-column piece) ----
    auto item = [&](int k, int& cl, int& row, int& xc) {               // cheap (mul-shift divisions), recomputed instead of kept live
        int idx = tid + k * NTHR;
        asm volatile("" : "+v"(idx));                                  // opaque: keeps LICM from hoisting (and then spilling) the results
        cl = idx / (RH * (RX / 8));
        const int rem = idx - cl * (RH * (RX / 8));
        row = rem / (RX / 8); xc = rem - row * (RX / 8);
        return idx < NITEMS;
    };
    int lofs[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        int cl, row, xc;
        lofs[k] = item(k, cl, row, xc) ? ((cl * RH + row) * RX + xc * 8) * 2 : -1;
    }
    auto plan_tile = [&](int tile, int* gofs) {         // in-frame element offset of each item for chunk 0, or -1 (outside); returns t
        const int t = tile / tpf, rem = tile - t * tpf, ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            int cl, row, xc;
            const bool live = item(k, cl, row, xc);
            const int gy = ty * TH - 2 + row, gx = tx * TW - 8 + xc * 8;
            gofs[k] = (live && gy >= 0 && gy < h && gx >= 0 && gx < wr) ? (gy * C + cl) * wr + gx : -1;
        }
        return t;
    };
    // UNCONDITIONAL loads (clamped offset, masked at the LDS write): a branch around a load makes the compiler wait for it at once
    auto issue_loads = [&](uint4* stg, const int* gofs, int t, int q) {
        const bf16_t* gt = g1p + (size_t)t * frame + (size_t)q * CHK * wr;
#pragma unroll
        for (int k = 0; k < NIT; ++k) stg[k] = *(const uint4*)(gt + (gofs[k] < 0 ? 0 : gofs[k]));
    };
    auto write_gimg = [&](int buf, const uint4* stg, const int* gofs) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (lofs[k] >= 0) *(uint4*)(gimg + buf * GIMG_BYTES + lofs[k]) = gofs[k] < 0 ? make_uint4(0, 0, 0, 0) : stg[k];
    };

    // ---- phase 1 role: channel pair pr of the chunk, rows 4 hf .. 4 hf + 3; lane column n = p -> (x-tile xt, row rr) ----
    // A row m = p, k-block g: the lane's 8 band values start at element s0 = 1 - p + 8 g of the padded band (all-zero window: 12)
    const int pr = wv / (TH / 4), hf = wv % (TH / 4), xt = p & 3, rr = p >> 2;
    // Only the FIRST and LAST dword of the window are read from LDS (W: elements s0, s0+1; X: s0+6, s0+7).  The windows of
    // neighbouring lanes are the same band shifted by one element, so dword 1 = W of lane m-2 (or X of lane m+4 at the row's
    // low edge) and dword 2 = X of lane m+2 (or W of lane m-4 at the high edge): DPP row shifts, 8 bytes of LDS per fragment.
    const int s0 = 1 - p + 8 * g;
    const int twx = (s0 < 0 || s0 > 5) ? 0 : ((s0 & 1) ? 10 + (s0 + 5) / 2 : (s0 + 6) / 2);    // dword inside the 20-dword record
    const int tww = (s0 < 6 || s0 > 11) ? 0 : ((s0 & 1) ? 10 + (s0 - 1) / 2 : s0 / 2);        // (dword 0 of a record is zero)
    const int boff = ((4 * hf + rr) * RX + 16 * xt + 8 * g) * 2;                // + (channel * RH + i) * RX * 2
    const int px0 = (4 * hf + rr) * TW + 16 * xt + 4 * g;

    // XCD-aware persistent schedule (workgroup b runs on XCD b % 8, each XCD has its own L2): XCD x walks the x-th contiguous
    // eighth of the tile list with its gridDim/8 workgroups side by side, so tiles that share halo rows and edge lines meet in one L2.
    const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1, wpx = gridDim.x / nxcd, seg = (ntiles + nxcd - 1) / nxcd;
    const int seg0 = (blockIdx.x % nxcd) * seg, seg1 = seg0 + seg < ntiles ? seg0 + seg : ntiles;
    int tile = seg0 + blockIdx.x / nxcd;

    // software pipeline over steps (tile, chunk q): HBM -> registers three steps ahead, registers -> LDS one step ahead, one barrier per step
    uint4 stgA[NIT], stgB[NIT];                 // data of even / odd steps
    int gofs[NIT], gofs_n[NIT];
    int t = plan_tile(tile < seg1 ? tile : 0, gofs), tn = t;
    issue_loads(stgA, gofs, t, 0);
    issue_loads(stgB, gofs, t, 1);
    write_gimg(0, stgA, gofs);
    issue_loads(stgA, gofs, t, 2);
    __syncthreads();                                          // table (and chunk 0) ready
    for (; tile < seg1; tile += wpx) {
        const int rem = tile - t * tpf, tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        const int ntile = tile + wpx < seg1 ? tile + wpx : tile;      // past the end: re-read this tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 0) tn = plan_tile(ntile, gofs_n);
            {   // (a) data of the next step: registers -> the other LDS buffer; (b) refill those registers for three steps ahead
                uint4* stg = (q & 1) ? stgA : stgB;               // step q + 1 has the opposite parity
                write_gimg((q + 1) & 1, stg, q == 3 ? gofs_n : gofs);
                if (q == 0) issue_loads(stg, gofs, t, 3); else issue_loads(stg, gofs_n, tn, q - 1);
            }
            // (c) Toeplitz MFMAs of this step from buffer q & 1
            const char* gb = gimg + (q & 1) * GIMG_BYTES + boff;
            f32x4_t D[2];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int cl = 2 * pr + ci, c = q * CHK + cl;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                {
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy) {
                        const int x3 = (int)tab[(c * 5 + dy) * 20 + twx], w0 = (int)tab[(c * 5 + dy) * 20 + tww];
                        const int d1 = __builtin_amdgcn_update_dpp(dpp_movi<0x104>(x3), w0, 0x112, 0xf, 0xf, false);   // row_shr:2 | row_shl:4
                        const int d2 = __builtin_amdgcn_update_dpp(dpp_movi<0x114>(w0), x3, 0x102, 0xf, 0xf, false);   // row_shl:2 | row_shr:4
                        const bf16x8_t A = as_frag(make_uint4((uint32_t)w0, (uint32_t)d1, (uint32_t)d2, (uint32_t)x3));
                        const bf16x8_t B = as_frag(*(const uint4*)(gb + (cl * RH + dy) * RX * 2));
                        acc = mfma16(A, B, acc);
                    }
                }
                D[ci] = acc * (ca_in ? ca_in[(size_t)t * C + c] : 1.f);
            }
            // channel pair complete: 4 consecutive pixels of one row per lane -> one 16-byte LDS store into the pair's plane
            *(uint4*)(lds_r + (q * (CHK / 2) + pr) * RP + px0) = make_uint4(pack_bf2(D[0][0], D[1][0]), pack_bf2(D[0][1], D[1][1]),
                                                                            pack_bf2(D[0][2], D[1][2]), pack_bf2(D[0][3], D[1][3]));
            __syncthreads();                  // next step's buffer complete, this step's buffer free; after q = 3: r complete
        }

        // ---- phase 2: 1x1 C -> 2C (gate-paired rows) on the finished r tile, SimpleGate2, NHWC stores, channel sums ----
        float ps[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) ps[j][r4] = 0.f;
#pragma unroll 1
        for (int n = 0; n < 4; ++n) {
            const int tp = (ng * 4 + n) * 16 + p;
            bf16x8_t Bf[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const uint32_t* rp = lds_r + (16 * s + 4 * g) * RP + tp;
                Bf[s] = as_frag(make_uint4(rp[0], rp[RP], rp[2 * RP], rp[3 * RP]));
            }
            f32x4_t acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc[m] = mfma16(A2[m][s], Bf[s], acc[m]);
            }
            const int oy = y0 + tp / TW, ox = x0 + (tp % TW);
            if (oy < h && ox < w) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) { v[r4] = acc[2 * j][r4] * sigmoidf_(acc[2 * j + 1][r4]); ps[j][r4] += v[r4]; }
                    o[2 * j] = pack_bf2(v[0], v[1]); o[2 * j + 1] = pack_bf2(v[2], v[3]);
                }
                *(uint4*)(g2 + (((size_t)t * h + oy) * w + ox) * C + g * 16 + mh * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float sm = row_sum16(ps[j][r4]);
                if (p == 0) red[wv * 32 + g * 8 + j * 4 + r4] = sm;
            }
        __syncthreads();                                      // every wave is done reading r; red complete
        if (pool && tid < C) {                                // channel tid = g*16 + mh*8 + j*4 + r4, summed over the 8 pixel groups
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NWV / 2; ++k) sm += red[(2 * k + ((tid >> 3) & 1)) * 32 + (tid >> 4) * 8 + (tid & 7)];
            pool[((size_t)t * tpf + rem) * C + tid] = sm;
        }
        // red is rewritten only after the next tile's first __syncthreads()
#pragma unroll
        for (int k = 0; k < NIT; ++k) gofs[k] = gofs_n[k];
        t = tn;
    }
}

// NHWC [T][h][w][C] -> planar [T][h][C][wr] (pad columns zero).  64-pixel row segments through LDS.
__global__ __launch_bounds__(256) void nhwc_to_planar_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ xp, int h, int w, int C, int wr) {
    __shared__ bf16_t tile[64][136];              // [px][channel], C <= 128; row pitch 272 B
    const int tid = threadIdx.x, t = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 64;
    const int npc = C / 8;
    for (int it = tid; it < 64 * npc; it += 256) {
        const int px = it / npc, pc = it - px * npc, gx = x0 + px;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gx < w) v = *(const uint4*)(x + (((size_t)t * h + y) * w + gx) * C + pc * 8);
        *(uint4*)(&tile[px][pc * 8]) = v;
    }
    __syncthreads();
    for (int it = tid; it < C * 8; it += 256) {
        const int c = it >> 3, xc = it & 7, gx = x0 + xc * 8;
        if (gx >= wr) continue;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (uint32_t)tile[xc * 8 + 2 * j][c] | ((uint32_t)tile[xc * 8 + 2 * j + 1][c] << 16);
        *(uint4*)(xp + (((size_t)t * h + y) * C + c) * wr + gx) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

extern "C" {
int sn_planar_pitch(int w);
}

template <int TH>
static int launch_k3m(const void* g1p, const float* ca_in, const void* ttab, const void* wfrag, void* g2, float* pool, int T, int h, int w,
                      void* stream) {
    using SH = K3mShape<TH>;
    const int ntiles = T * ((h + TH - 1) / TH) * ((w + K3M_TW - 1) / K3M_TW);
    int dev = 0, ncu = 0;                                   // persistent: one workgroup per CU of THIS device (LDS-limited)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1)
        return SN_ELAUNCH;
    const int nwg = ntiles < ncu ? ntiles : ncu;
    if (hipFuncSetAttribute((const void*)dw5m_gemm_gate_kernel<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::LDS) != hipSuccess)
        return SN_ELAUNCH;
    sn_clear_error();
    hipLaunchKernelGGL(dw5m_gemm_gate_kernel<TH>, dim3(nwg), dim3(SH::NTHR), SH::LDS, (hipStream_t)stream, (const bf16_t*)g1p, ca_in,
                       (const uint32_t*)ttab, (const uint4*)wfrag, (bf16_t*)g2, pool, T, h, w, sn_planar_pitch(w));
    return sn_check_launch();
}

extern "C" {

int sn_planar_pitch(int w) { return (w + 7) & ~7; }

int sn_nhwc_to_planar(const void* x, void* xp, int T, int h, int w, int C, void* stream) {
    sn_clear_error();
    if (!x || !xp || C < 8 || C > 128 || (C & 7) || T < 1 || h < 1 || w < 1) return SN_EINVAL;
    hipLaunchKernelGGL(nhwc_to_planar_kernel, dim3((w + 63) / 64, h, T), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)xp, h, w, C, sn_planar_pitch(w));
    return sn_check_launch();
}


// Tile of sn_dw5m_gemm_gate: 64 x 8 pixels, one 1024-thread workgroup per CU (381 us at level 1 of config 2).
#define SN_K3M_TH 8

int sn_dw5m_blocks(int h, int w) { return ((h + SN_K3M_TH - 1) / SN_K3M_TH) * ((w + K3M_TW - 1) / K3M_TW); }

int sn_dw5m_gemm_gate(const void* g1p, const float* ca_in, const void* ttab, const void* wfrag, void* g2, float* pool,
                      int T, int h, int w, int C, void* stream) {
    sn_clear_error();
    if (!g1p || !ttab || !wfrag || !g2 || C != 64 || T < 1 || h < 1 || w < 1) return SN_EINVAL;
    return launch_k3m<SN_K3M_TH>(g1p, ca_in, ttab, wfrag, g2, pool, T, h, w, stream);
}

}  // extern "C"
