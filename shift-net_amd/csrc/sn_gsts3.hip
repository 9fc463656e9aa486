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
// Two shapes of the same kernel.  TH = 8 (default): 1024 threads, one workgroup per CU, double-buffered staging image (one
// barrier per step).  TH = 4 (SN_K3M_TH=4, kept for A/B): 512 threads, two independent workgroups per CU (80 KB of LDS
// each, single staging buffer, two barriers per step).
template <int TH> struct K3mShape {
    static constexpr int RH = TH + 4, NWV = 2 * TH, NTHR = 64 * NWV, NBUF = TH == 8 ? 2 : 1;
    static constexpr int GIMG_BYTES = K3M_CHK * RH * K3M_RX * 2;       // 30720 / 20480 per buffer
    static constexpr int RPITCH = K3M_TW * TH + 4;                     // dwords per channel-pair plane of r (lane groups g land 16 banks apart)
    static constexpr int R_BYTES = (K3M_C / 2) * RPITCH * 4;           // 66048 / 33280
    static constexpr int RED_BYTES = NWV * 32 * 4;
    static constexpr int LDS = K3M_TAB_BYTES + NBUF * GIMG_BYTES + R_BYTES + RED_BYTES;      // 155136 / 80384
};

template <int TH>
__global__ __launch_bounds__(K3mShape<TH>::NTHR, TH == 8 ? 1 : 4)
void dw5m_gemm_gate_kernel(const bf16_t* __restrict__ g1p, const float* __restrict__ ca_in, const uint32_t* __restrict__ ttab,
                           const uint4* __restrict__ wfrag, bf16_t* g2, float* pool, int T, int h, int w, int wr, const int dbg_,
                           unsigned long long* prof_) {
#ifdef SN_EXPERIMENTAL
    const int dbg = dbg_; unsigned long long* const prof = prof_;                 // ablation / phase-clock hooks (tools/prof_k3m.py)
#else
    constexpr int dbg = 0; constexpr unsigned long long* prof = nullptr;          // production: the hooks fold away
#endif
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

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    auto tick = [&](int slot) {
        if (prof) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tacc[slot] += now - tlast; tlast = now; }
    };

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
            gofs[k] = (live && gy >= 0 && gy < h && gx >= 0 && gx < wr && !(dbg & 1)) ? (gy * C + cl) * wr + gx : -1;
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

    // software pipeline over steps (tile, chunk q).  NBUF = 2: HBM -> registers three steps ahead, registers -> LDS one step
    // ahead, one barrier per step.  NBUF = 1: HBM -> registers two steps ahead, registers -> LDS at the start of the step.
    uint4 stgA[NIT], stgB[NIT];                 // data of even / odd steps
    int gofs[NIT], gofs_n[NIT];
    int t = plan_tile(tile < seg1 ? tile : 0, gofs), tn = t;
    issue_loads(stgA, gofs, t, 0);
    issue_loads(stgB, gofs, t, 1);
    if (NBUF == 2) {
        write_gimg(0, stgA, gofs);
        issue_loads(stgA, gofs, t, 2);
    }
    __syncthreads();                                          // table (and chunk 0) ready
    for (; tile < seg1; tile += wpx) {
        const int rem = tile - t * tpf, tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        const int ntile = tile + wpx < seg1 ? tile + wpx : tile;      // past the end: re-read this tile
        tick(7);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 0) tn = plan_tile(ntile, gofs_n);
            if (NBUF == 2) {
                // (a) data of the next step: registers -> the other LDS buffer; (b) refill those registers for three steps ahead
                uint4* stg = (q & 1) ? stgA : stgB;               // step q + 1 has the opposite parity
                write_gimg((q + 1) & 1, stg, q == 3 ? gofs_n : gofs);
                if (q == 0) issue_loads(stg, gofs, t, 3); else issue_loads(stg, gofs_n, tn, q - 1);
            } else {
                // (a) data of THIS step: registers -> the LDS buffer (free since the barrier that ended the previous step);
                // (b) refill those registers for two steps ahead
                uint4* stg = (q & 1) ? stgB : stgA;
                write_gimg(0, stg, gofs);
                __syncthreads();
                if (q < 2) issue_loads(stg, gofs, t, q + 2); else issue_loads(stg, gofs_n, tn, q - 2);
            }
            tick(0);
            // (c) Toeplitz MFMAs of this step from buffer q & 1
            const char* gb = gimg + (NBUF == 2 ? (q & 1) : 0) * GIMG_BYTES + boff;
            f32x4_t D[2];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int cl = 2 * pr + ci, c = q * CHK + cl;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                if (!(dbg & 2)) {
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
            if (!(dbg & 8))
                *(uint4*)(lds_r + (q * (CHK / 2) + pr) * RP + px0) = make_uint4(pack_bf2(D[0][0], D[1][0]), pack_bf2(D[0][1], D[1][1]),
                                                                            pack_bf2(D[0][2], D[1][2]), pack_bf2(D[0][3], D[1][3]));
            tick(3);
            __syncthreads();                  // next step's buffer complete, this step's buffer free; after q = 3: r complete
            tick(5);
        }

        // ---- phase 2: 1x1 C -> 2C (gate-paired rows) on the finished r tile, SimpleGate2, NHWC stores, channel sums ----
        float ps[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) ps[j][r4] = 0.f;
#pragma unroll 1
        for (int n = (dbg & 4) ? 4 : 0; n < 4; ++n) {
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
        tick(6);
        __syncthreads();                                      // every wave is done reading r; red complete
        tick(1);
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
    if (prof && lane == 0 && wv < 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) prof[((size_t)blockIdx.x * 8 + wv) * 8 + k] = tacc[k];
    }
}

// ------------------------------------------------------------------------------------------------------------
// K3w: the same operator and tile (64 x 8 pixels, 1024 persistent threads, XCD-aware walk), restructured around the finding
// that dw5m_gemm_gate_kernel is bound by the skew of its five workgroup barriers per tile (42 % of wave cycles in s_barrier).
// Here every wave stages ITS OWN two channels of the step (12 rows x 80 columns, 3.84 KB) into a wave-private LDS image: LDS
// operations of one wave execute in order, so the staging needs no barrier at all and the 16 waves drift freely through the
// Toeplitz phase (loads, LDS writes, B reads and MFMAs of different waves overlap instead of running in lock-step).  A step
// is 32 channels (wave w = channel pair 16 s + w, both row halves, the band fragment of a (channel, dy) shared by the halves:
// half the table reads), so phase 1 is two steps and the tile has TWO barriers: r complete / r consumed.  Same LDS budget
// (table 25.6 KB + 16 x 3.84 KB images + r 66 KB), same r layout and phase 2 as dw5m_gemm_gate_kernel, bit-identical results
// (tools/ab_k3m.py: 9 shapes incl. ragged ones, g2 and pool equal bit for bit).
// MEASURED: standalone on a cold 590 MB input (20 x 360 x 640) 419 vs 460 us, 100 vs 100 us at level 2; inside the network, where
// g1 was written by K12 a moment earlier, 24.2 vs 23.0 ms per window (config 2) -- slower.  The lock-step kernel profits from the
// warm L2 / Infinity Cache, the free-running one from hiding HBM latency; the product path keeps dw5m_gemm_gate_kernel and this
// shape is a compile-time option (-DSN_K3M_WAVE=1).
constexpr int K3W_WIMG = 2 * 12 * K3M_RX * 2;                                   // 3840 B per wave
constexpr int K3W_LDS = K3M_TAB_BYTES + 16 * K3W_WIMG + (K3M_C / 2) * (K3M_TW * 8 + 4) * 4 + 16 * 32 * 4;      // 155136

__global__ __launch_bounds__(1024, 1)
void dw5w_gemm_gate_kernel(const bf16_t* __restrict__ g1p, const float* __restrict__ ca_in, const uint32_t* __restrict__ ttab,
                           const uint4* __restrict__ wfrag, bf16_t* g2, float* pool, int T, int h, int w, int wr) {
    constexpr int C = K3M_C, TW = K3M_TW, TH = 8, RH = 12, RX = K3M_RX, KS = 2, RP = TW * TH + 4, NWV = 16, NTHR = 1024, NIT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t* tab = (const uint32_t*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    char* wimg = smem + K3M_TAB_BYTES + wv * K3W_WIMG;                          // [2 channels][RH][RX] bf16, private to this wave
    uint32_t* lds_r = (uint32_t*)(smem + K3M_TAB_BYTES + NWV * K3W_WIMG);       // [32 channel pairs][RP]
    float* red = (float*)((char*)lds_r + (C / 2) * RP * 4);
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH, tpf = tiles_x * tiles_y, ntiles = T * tpf;
    const size_t frame = (size_t)h * C * wr;

    for (int e = tid; e < K3M_TAB_BYTES / 4; e += NTHR) ((uint32_t*)smem)[e] = ttab[e];
    const int ng = wv >> 1, mh = wv & 1;                                        // phase-2 role, as in dw5m_gemm_gate_kernel
    bf16x8_t A2[4][KS];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < KS; ++s) A2[m][s] = as_frag(wfrag[((4 * mh + m) * KS + s) * 64 + lane]);

    // staging item k of this lane: 16 B = 8 columns of (channel cl of the pair, region row, piece xc); 240 items per wave and step
    int lofs[NIT], irow[NIT], ixc[NIT], icl[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = lane + 64 * k;
        const int cl = idx >= RH * (RX / 8) ? 1 : 0, rem = idx - cl * (RH * (RX / 8)), row = rem / (RX / 8), xc = rem - row * (RX / 8);
        const bool live = idx < 2 * RH * (RX / 8);
        lofs[k] = live ? ((cl * RH + row) * RX + xc * 8) * 2 : -1;
        irow[k] = row; ixc[k] = xc; icl[k] = cl;
    }
    auto plan_tile = [&](int tile, int* gofs) {          // in-frame element offset of each item relative to the pair's first channel, or -1
        const int t = tile / tpf, rem = tile - t * tpf, ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int gy = ty * TH - 2 + irow[k], gx = tx * TW - 8 + ixc[k] * 8;
            gofs[k] = (lofs[k] >= 0 && gy >= 0 && gy < h && gx >= 0 && gx < wr) ? (gy * C + icl[k]) * wr + gx : -1;
        }
        return t;
    };
    auto issue_loads = [&](uint4* stg, const int* gofs, int t, int s) {          // unconditional, clamped; masked at the LDS write
        const bf16_t* gt = g1p + (size_t)t * frame + (size_t)(32 * s + 2 * wv) * wr;
#pragma unroll
        for (int k = 0; k < NIT; ++k) stg[k] = *(const uint4*)(gt + (gofs[k] < 0 ? 0 : gofs[k]));
    };
    auto write_img = [&](const uint4* stg, const int* gofs) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (lofs[k] >= 0) *(uint4*)(wimg + lofs[k]) = gofs[k] < 0 ? make_uint4(0, 0, 0, 0) : stg[k];
    };

    const int xt = p & 3, rr = p >> 2;
    const int s0 = 1 - p + 8 * g;                                               // band window of this lane, see dw5m_gemm_gate_kernel
    const int twx = (s0 < 0 || s0 > 5) ? 0 : ((s0 & 1) ? 10 + (s0 + 5) / 2 : (s0 + 6) / 2);
    const int tww = (s0 < 6 || s0 > 11) ? 0 : ((s0 & 1) ? 10 + (s0 - 1) / 2 : s0 / 2);
    const int boff = (rr * RX + 16 * xt + 8 * g) * 2;                           // + ((ci * RH + 4 hf + dy) * RX) * 2
    const int px0 = rr * TW + 16 * xt + 4 * g;                                  // + 4 hf * TW

    // Toeplitz MFMAs of step s (channels 32 s + 2 wv, + 1) from the private image -> the pair's r plane
    auto toeplitz = [&](int s, int t) {
        f32x4_t D[2][2];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            const int c = 32 * s + 2 * wv + ci;
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const int x3 = (int)tab[(c * 5 + dy) * 20 + twx], w0 = (int)tab[(c * 5 + dy) * 20 + tww];
                const int d1 = __builtin_amdgcn_update_dpp(dpp_movi<0x104>(x3), w0, 0x112, 0xf, 0xf, false);
                const int d2 = __builtin_amdgcn_update_dpp(dpp_movi<0x114>(w0), x3, 0x102, 0xf, 0xf, false);
                const bf16x8_t A = as_frag(make_uint4((uint32_t)w0, (uint32_t)d1, (uint32_t)d2, (uint32_t)x3));
                const char* gb = wimg + boff + ((ci * RH + dy) * RX) * 2;
                a0 = mfma16(A, as_frag(*(const uint4*)gb), a0);
                a1 = mfma16(A, as_frag(*(const uint4*)(gb + 4 * RX * 2)), a1);
            }
            const float sc = ca_in ? ca_in[(size_t)t * C + c] : 1.f;
            D[0][ci] = a0 * sc; D[1][ci] = a1 * sc;
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            *(uint4*)(lds_r + (16 * s + wv) * RP + px0 + 4 * hf * TW) =
                make_uint4(pack_bf2(D[hf][0][0], D[hf][1][0]), pack_bf2(D[hf][0][1], D[hf][1][1]),
                           pack_bf2(D[hf][0][2], D[hf][1][2]), pack_bf2(D[hf][0][3], D[hf][1][3]));
    };

    const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1, wpx = gridDim.x / nxcd, seg = (ntiles + nxcd - 1) / nxcd;
    const int seg0 = (blockIdx.x % nxcd) * seg, seg1 = seg0 + seg < ntiles ? seg0 + seg : ntiles;
    int tile = seg0 + blockIdx.x / nxcd;

    uint4 stg[NIT];
    int gofs[NIT], gofs_n[NIT];
    int t = plan_tile(tile < seg1 ? tile : 0, gofs), tn = t;
    issue_loads(stg, gofs, t, 0);
    write_img(stg, gofs);                                   // step 0 of the first tile
    issue_loads(stg, gofs, t, 1);                           // step 1 in flight
    __syncthreads();                                        // band table ready
    for (; tile < seg1; tile += wpx) {
        const int rem = tile - t * tpf, tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        const int ntile = tile + wpx < seg1 ? tile + wpx : tile;         // past the end: re-read this tile
        tn = plan_tile(ntile, gofs_n);
        toeplitz(0, t);
        write_img(stg, gofs);                               // step 1 -> the image (in order behind step 0's reads of this wave)
        issue_loads(stg, gofs_n, tn, 0);
        toeplitz(1, t);
        write_img(stg, gofs_n);                             // next tile's step 0
        issue_loads(stg, gofs_n, tn, 1);
        __syncthreads();                                    // r complete

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
        __syncthreads();                                    // r consumed, red complete
        if (pool && tid < C) {
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NWV / 2; ++k) sm += red[(2 * k + ((tid >> 3) & 1)) * 32 + (tid >> 4) * 8 + (tid & 7)];
            pool[((size_t)t * tpf + rem) * C + tid] = sm;
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) gofs[k] = gofs_n[k];
        t = tn;
    }
}

#ifdef SN_EXPERIMENTAL   // K12m: parity green, not faster than sn_ln_gemm_gate yet; off the production path
// ------------------------------------------------------------------------------------------------------------
// K12m: g1 = SimpleGate(RepConv2(body[0](norm(u)))) for C = 64 with the depthwise 3x3 on the matrix cores, g1 written
// channel-planar.  Same chunk pipeline as sn_ln_gemm_gate (LayerNorm'd operands resident in registers, the 2C-channel
// tensor `a` only ever exists 32 channels at a time in LDS, next chunk's GEMM overlaps this chunk's stencil), but
//   * the GEMM runs with the PIXELS on M (A = normalised activations, B = weights: the same prepacked fragments, operands
//     swapped), so a lane's 4 accumulators are 4 consecutive columns of one channel: `a` goes to LDS channel-planar with
//     ds_write_b64 -- exactly the layout the Toeplitz MFMAs read their B operands from;
//   * region = 10 rows x 48 columns (3 M-tiles per row, the tile's 32 columns start at column 8): windows stay 16-byte
//     aligned; 30 M-tiles per workgroup instead of 22 (the price of planar rows);
//   * stencil: per gate pair two channels x 3 kernel rows = 6 MFMAs for the whole 8 x 32 tile (N = 2 x-tiles x 8 rows),
//     band fragments from a per-chunk table in LDS (7.5 KB, double buffered, prefetched through registers);
//   * SimpleGate is lane-local (both halves of a pair come out in the same lane layout); the product is 4 consecutive
//     columns of one g1 channel = one 8-byte store into the planar tensor, no transpose epilogue.
struct UnitK3 {
    const bf16_t* x;
    int T, h, w, C, mode, wrap;
};
struct Slabs3 { int f0, o0, f1, o1; };
__device__ __forceinline__ Slabs3 unit_slabs3(const UnitK3& U, int t) {      // same table as sn_gsts2.hip::unit_slabs2
    const int Ch = U.C >> 1;
    Slabs3 s; s.f0 = t; s.o0 = 0; s.f1 = t; s.o1 = Ch;
    if (U.mode == 1) { if (t > 0 || U.wrap) { s.f0 = sn_prev_frame(t, U.T, U.wrap); s.o0 = Ch; s.f1 = t; s.o1 = 0; } }
    else if (U.mode == 2) { if (t < U.T - 1 || U.wrap) { s.f0 = t; s.o0 = Ch; s.f1 = sn_next_frame(t, U.T, U.wrap); s.o1 = 0; } }
    return s;
}

constexpr int K12M_TH = 8, K12M_TW = 32, K12M_RH = 10, K12M_RC = 48;
constexpr int K12M_PLANE = K12M_RH * K12M_RC + 8;           // elements per channel plane; +8: the 16 planes a ds_write_b64 group touches spread over the banks (2-way instead of 8-way)
constexpr int K12M_ABUF = 32 * K12M_PLANE * 2;              // 30720 B: 32 channel planes of one chunk
constexpr int K12M_TDW = 32 * 3 * 20;                       // 1920 dwords: band records of one chunk
constexpr int K12M_LDS = 2 * K12M_ABUF + 2 * K12M_TDW * 4;  // 77824 B -> two workgroups per CU

template <bool WITH_HW>
__global__ __launch_bounds__(512) void ln_gemm_gate_m_kernel(const UnitK3 U, const bf16_t* __restrict__ hwb, const uint4* __restrict__ wfrag,
                                                           const float* __restrict__ bias, const uint32_t* __restrict__ ttab3,
                                                           bf16_t* g1p, float* pool, const int wr, unsigned long long* prof) {
    constexpr int C = 64, CH = 32, K = WITH_HW ? C + CH : C, KS = K / 32, MT = 8, NCHK = 4;
    constexpr int TH = K12M_TH, TW = K12M_TW, RH = K12M_RH, RC = K12M_RC, PLANE = K12M_PLANE;
    constexpr int NWV = 8, NMT = RH * 3, NTW = (NMT + NWV - 1) / NWV;            // 30 M-tiles of 16 region pixels, <= 4 per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_a = smem;                                                  // [2][32 planes][RH][RC] bf16
    uint32_t* lds_t = (uint32_t*)(smem + 2 * K12M_ABUF);                 // [2][32 planes][3][20]
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    const int t = blockIdx.z, oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int hw = U.h * U.w;
    const Slabs3 sl = unit_slabs3(U, t);

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    auto tick = [&](int slot) {
        if (prof) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tacc[slot] += now - tlast; tlast = now; }
    };
    // weight fragments, bias and band table of one chunk, fetched one chunk AHEAD through registers
    bf16x8_t Wf[2][KS];
    float Wb[2];
    uint32_t Tq[4];
    auto load_w = [&](int q) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Wf[0][s] = as_frag(wfrag[((2 * q) * KS + s) * 64 + lane]);
            Wf[1][s] = as_frag(wfrag[((2 * q + 1) * KS + s) * 64 + lane]);
        }
        // as the B operand lane (g, n = p) carries weight row n of the 16-row block: its bias sits at (n>>2)*4*MT + mt*4 + (n&3)
        Wb[0] = bias[(p >> 2) * 4 * MT + (2 * q) * 4 + (p & 3)];
        Wb[1] = bias[(p >> 2) * 4 * MT + (2 * q + 1) * 4 + (p & 3)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 512;
            Tq[k] = ttab3[q * K12M_TDW + (idx < K12M_TDW ? idx : 0)];
        }
    };
    load_w(0);

    // ---- LayerNorm of the 30 M-tiles (16 consecutive region columns of one region row); operands stay in registers ----
    bf16x8_t Xf[NTW][KS];
    int ry_[NTW], cb_[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int mt = wv + NWV * n;                          // wave-uniform
        const int ry = mt / 3, cb = mt - ry * 3;
        ry_[n] = ry; cb_[n] = cb;
        const int gy = oy0 - 1 + ry, gx = ox0 - 8 + 16 * cb + p;
        const bool in = mt < NMT && gy >= 0 && gy < U.h && gx >= 0 && gx < U.w;
        const int ii = in ? gy * U.w + gx : 0;
        float xv[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk0 = s * 32 + g * 8;                   // K is a multiple of 32 here: every slab is real
            const bf16_t* s0 = U.x + ((ptrdiff_t)sl.f0 * hw + ii) * C + sl.o0 + (kk0 < CH ? kk0 : 0);
            const bf16_t* s1 = U.x + ((ptrdiff_t)sl.f1 * hw + ii) * C + sl.o1 + (kk0 >= CH && kk0 < C ? kk0 - CH : 0);
            const bf16_t* src = kk0 < CH ? s0 : s1;
            if (WITH_HW) {
                const bf16_t* s2 = hwb + ((size_t)t * hw + ii) * CH + (kk0 >= C ? kk0 - C : 0);
                src = kk0 >= C ? s2 : src;
            }
            unpack8(*(const uint4*)src, xv[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += xv[s][j];
        }
        sum = sum_rows4(sum);
        const float mean = sum * (1.0f / K);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = xv[s][j] - mean; xv[s][j] = d; sq += d * d; }
        sq = sum_rows4(sq);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / K) + 1e-6f);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[s][j] *= rstd;
            Xf[n][s] = as_frag(pack8(xv[s]));
        }
    }

    // GEMM chunk q -> LDS buffer q & 1 (planes 0..15: first-half channels, 16..31: their gate partners), zero outside the image;
    // the chunk's band table (prefetched into Tq) goes to the table buffer of the same parity.
    auto gemm_chunk = [&](int q) {
        char* ab = lds_a + (q & 1) * K12M_ABUF;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            if (wv + NWV * n >= NMT) continue;                // wave-uniform
            f32x4_t acc0 = {Wb[0], Wb[0], Wb[0], Wb[0]}, acc1 = {Wb[1], Wb[1], Wb[1], Wb[1]};
#pragma unroll
            for (int s = 0; s < KS; ++s) { acc0 = mfma16(Xf[n][s], Wf[0][s], acc0); acc1 = mfma16(Xf[n][s], Wf[1][s], acc1); }
            // lane (g, p): pixels = region columns 16 cb + 4g + r of row ry, channel row p
            const int gy = oy0 - 1 + ry_[n], gx0 = ox0 - 8 + 16 * cb_[n] + 4 * g;
            const bool yok = gy >= 0 && gy < U.h;
            float v0[4], v1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = yok && gx0 + r >= 0 && gx0 + r < U.w;
                v0[r] = in ? acc0[r] : 0.f; v1[r] = in ? acc1[r] : 0.f;
            }
            const int off = (ry_[n] * RC + 16 * cb_[n] + 4 * g) * 2;
            *(uint2*)(ab + (p * PLANE) * 2 + off) = make_uint2(pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]));
            *(uint2*)(ab + ((16 + p) * PLANE) * 2 + off) = make_uint2(pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3]));
        }
        uint32_t* tb = lds_t + (q & 1) * K12M_TDW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 512;
            if (idx < K12M_TDW) tb[idx] = Tq[k];
        }
    };
    tick(0);
    gemm_chunk(0);
    load_w(1);
    tick(1);

    // Toeplitz constants (3-tap band, prep.pack_toeplitz(k=3)): window start s0 = -m + 8g; see dw5m_gemm_gate_kernel
    const int s0 = -p + 8 * g;
    const int twx = (s0 < 0 || s0 > 5) ? 0 : ((s0 & 1) ? 10 + (s0 + 5) / 2 : (s0 + 6) / 2);
    const int tww = (s0 < 6 || s0 > 11) ? 0 : ((s0 & 1) ? 10 + (s0 - 1) / 2 : s0 / 2);
    const int xt = p & 1, row = p >> 1;                                   // B column n = p: x-tile (16 columns) and tile row
    const int boff = (row * RC + 16 * xt + 8 * g) * 2;
    const int oy = oy0 + row, ox = ox0 + 16 * xt + 4 * g;                 // this lane's 4 output columns ox .. ox + 3
    const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;

#pragma unroll 1
    for (int q = 0; q < NCHK; ++q) {
        __syncthreads();          // chunk q (and its table) complete in buffer q&1; everybody is done with buffer (q+1)&1
        tick(2);
        if (q + 1 < NCHK) gemm_chunk(q + 1);
        load_w(q + 2 < NCHK ? q + 2 : NCHK - 1);
        tick(1);
        const char* ab = lds_a + (q & 1) * K12M_ABUF + boff;
        const uint32_t* tb = lds_t + (q & 1) * K12M_TDW;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int nn = 2 * wv + pi;                                   // gate pair of the chunk (wave-uniform)
            f32x4_t D[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int plane = half * 16 + nn;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int x3 = (int)tb[(plane * 3 + dy) * 20 + twx], w0 = (int)tb[(plane * 3 + dy) * 20 + tww];
                    const int d1 = __builtin_amdgcn_update_dpp(dpp_movi<0x104>(x3), w0, 0x112, 0xf, 0xf, false);
                    const int d2 = __builtin_amdgcn_update_dpp(dpp_movi<0x114>(w0), x3, 0x102, 0xf, 0xf, false);
                    const bf16x8_t A = as_frag(make_uint4((uint32_t)w0, (uint32_t)d1, (uint32_t)d2, (uint32_t)x3));
                    const bf16x8_t B = as_frag(*(const uint4*)(ab + (plane * PLANE + dy * RC) * 2));
                    acc = mfma16(A, B, acc);
                }
                D[half] = acc;
            }
            const int ch = (nn >> 2) * 2 * MT + q * 4 + (nn & 3);       // natural g1 channel of the pair
            float v[4];
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = oy < U.h && ox + r < U.w;
                v[r] = in ? D[0][r] * D[1][r] : 0.f;                     // zeros in the pad columns w .. wr-1
                ps += v[r];
            }
            if (oy < U.h && ox < wr)
                *(uint2*)(g1p + (((size_t)t * U.h + oy) * C + ch) * wr + ox) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            if (pool) {                       // denoise CALayer2 on g1: the whole wave holds ONE channel -> wave sum
                ps = sum_rows4(row_sum16(ps));
                if (lane == 0) pool[((size_t)t * nblk + blk) * C + ch] = ps;
            }
        }
        tick(3);
    }
    if (prof && lane == 0 && blockIdx.z == 0 && blk < 256) {
#pragma unroll
        for (int k = 0; k < 8; ++k) prof[((size_t)blk * 8 + wv) * 8 + k] = tacc[k];
    }
}

#endif  // SN_EXPERIMENTAL

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
#ifdef SN_EXPERIMENTAL
int sn_debug_get(void);
void* sn_debug_buf_get(void);
#endif
}
#ifdef SN_EXPERIMENTAL
#define SN3_DBG_MASK sn_debug_get()
#define SN3_DBG_BUF(bit) ((unsigned long long*)((sn_debug_get() & (bit)) ? sn_debug_buf_get() : nullptr))
#else
#define SN3_DBG_MASK 0
#define SN3_DBG_BUF(bit) ((unsigned long long*)nullptr)
#endif

template <int TH>
static int launch_k3m(const void* g1p, const float* ca_in, const void* ttab, const void* wfrag, void* g2, float* pool, int T, int h, int w,
                      void* stream) {
    using SH = K3mShape<TH>;
    const int ntiles = T * ((h + TH - 1) / TH) * ((w + K3M_TW - 1) / K3M_TW);
    int dev = 0, ncu = 0;                                   // persistent: one / two workgroups per CU of THIS device (LDS-limited)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1)
        return SN_ELAUNCH;
    const int maxwg = ncu * (TH == 8 ? 1 : 2);
    const int nwg = ntiles < maxwg ? ntiles : maxwg;
    if (hipFuncSetAttribute((const void*)dw5m_gemm_gate_kernel<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::LDS) != hipSuccess)
        return SN_ELAUNCH;
    sn_clear_error();
    hipLaunchKernelGGL(dw5m_gemm_gate_kernel<TH>, dim3(nwg), dim3(SH::NTHR), SH::LDS, (hipStream_t)stream, (const bf16_t*)g1p, ca_in,
                       (const uint32_t*)ttab, (const uint4*)wfrag, (bf16_t*)g2, pool, T, h, w, sn_planar_pitch(w), SN3_DBG_MASK, SN3_DBG_BUF(512));
    return sn_check_launch();
}

static int launch_k3w(const void* g1p, const float* ca_in, const void* ttab, const void* wfrag, void* g2, float* pool, int T, int h, int w,
                      void* stream) {
    const int ntiles = T * ((h + 7) / 8) * ((w + K3M_TW - 1) / K3M_TW);
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1)
        return SN_ELAUNCH;
    const int nwg = ntiles < ncu ? ntiles : ncu;
    if (hipFuncSetAttribute((const void*)dw5w_gemm_gate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, K3W_LDS) != hipSuccess)
        return SN_ELAUNCH;
    sn_clear_error();
    hipLaunchKernelGGL(dw5w_gemm_gate_kernel, dim3(nwg), dim3(1024), K3W_LDS, (hipStream_t)stream, (const bf16_t*)g1p, ca_in,
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

#ifdef SN_EXPERIMENTAL
int sn_lngatem_blocks(int h, int w) { return ((h + K12M_TH - 1) / K12M_TH) * ((w + K12M_TW - 1) / K12M_TW); }

int sn_ln_gemm_gate_m(const sn_unit_src* s, const void* hw, const void* wfrag, const float* bias, const void* ttab3,
                      void* g1p, float* pool, void* stream) {
    sn_clear_error();
    if (!s || !s->x || s->C != 64 || s->mode < 0 || s->mode > 2 || !wfrag || !bias || !ttab3 || !g1p || (s->mode != 0 && !hw))
        return SN_EINVAL;
    UnitK3 u; u.x = (const bf16_t*)s->x; u.T = s->T; u.h = s->h; u.w = s->w; u.C = s->C; u.mode = s->mode; u.wrap = s->wrap;
    dim3 grid((s->w + K12M_TW - 1) / K12M_TW, (s->h + K12M_TH - 1) / K12M_TH, s->T);
    hipStream_t st = (hipStream_t)stream;
    const int wr = sn_planar_pitch(s->w);
    if (s->mode) {
        if (hipFuncSetAttribute((const void*)ln_gemm_gate_m_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, K12M_LDS) != hipSuccess) return SN_ELAUNCH;
        sn_clear_error();
        hipLaunchKernelGGL(ln_gemm_gate_m_kernel<true>, grid, dim3(512), K12M_LDS, st, u, (const bf16_t*)hw, (const uint4*)wfrag, bias,
                           (const uint32_t*)ttab3, (bf16_t*)g1p, pool, wr, SN3_DBG_BUF(256));
    } else {
        if (hipFuncSetAttribute((const void*)ln_gemm_gate_m_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, K12M_LDS) != hipSuccess) return SN_ELAUNCH;
        sn_clear_error();
        hipLaunchKernelGGL(ln_gemm_gate_m_kernel<false>, grid, dim3(512), K12M_LDS, st, u, (const bf16_t*)hw, (const uint4*)wfrag, bias,
                           (const uint32_t*)ttab3, (bf16_t*)g1p, pool, wr, SN3_DBG_BUF(256));
    }
    return sn_check_launch();
}
#endif  // SN_EXPERIMENTAL

// Tile height of sn_dw5m_gemm_gate: 8 = one 1024-thread workgroup per CU (381 us at level 1).  The experimental build can
// also compile the 4-row shape (two 512-thread workgroups per CU: 432 us, more halo rows and barriers than the interleaving
// wins back) with -DSN_K3M_TH=4; it is a compile-time choice, the library reads no environment and keeps no state.
#ifndef SN_K3M_TH
#define SN_K3M_TH 8
#endif
#ifndef SN_K3M_WAVE          // 1: wave-private staging, two barriers per tile (dw5w_gemm_gate_kernel; slower inside the network, see there)
#define SN_K3M_WAVE 0
#endif

int sn_dw5m_blocks(int h, int w) { return ((h + SN_K3M_TH - 1) / SN_K3M_TH) * ((w + K3M_TW - 1) / K3M_TW); }

int sn_dw5m_gemm_gate(const void* g1p, const float* ca_in, const void* ttab, const void* wfrag, void* g2, float* pool,
                      int T, int h, int w, int C, void* stream) {
    sn_clear_error();
    if (!g1p || !ttab || !wfrag || !g2 || C != 64 || T < 1 || h < 1 || w < 1) return SN_EINVAL;
#if SN_K3M_WAVE && SN_K3M_TH == 8
    return launch_k3w(g1p, ca_in, ttab, wfrag, g2, pool, T, h, w, stream);
#else
    return launch_k3m<SN_K3M_TH>(g1p, ca_in, ttab, wfrag, g2, pool, T, h, w, stream);
#endif
}

}  // extern "C"
