// Grouped spatial-temporal shift unit (channel_shift -> CAB2 -> CAB1) for gfx950.
//
// Production kernels of this file:
//   sn_gsts_shiftconv (K0): hw = dw3x3(spatial_shift(borrowed half))     LDS-staged gather, never materialises the shift
//   sn_scale_gemm_res (K4): y  = roll(x) + beta * W3 . (ca * g2)         per-pixel MFMA, rolled shortcut
//   sn_gsts_gather / sn_temporal_roll: validation op / Shift_CAB roll (pure index work)
// -DSN_EXPERIMENTAL additionally builds the round-1 chain (one stencil or one GEMM per kernel, every intermediate in HBM):
//   sn_ln_gemm, sn_dw_gate, sn_dw_gemm_gate (include/shiftnet_hip_experimental.h)
// The temporal roll is only ever an address computation (frame/channel-offset pairs below).
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

namespace {

struct UnitK {
    const bf16_t* x;
    int T, h, w, C, mode, wrap;
};

struct Slabs {
    int f0, o0;   // u[:, :C/2]  = x[f0][o0 : o0 + C/2]
    int f1, o1;   // u[:, C/2:C] = x[f1][o1 : o1 + C/2]
    int fb, ob;   // borrowed half (input of the spatial shift) = x[fb][ob : ob + C/2]
};

// SURVEY.md 8a-1 table; gshift_deblur1.py:504-528 (keep) / gshift_deblur2.py:499-519 (wrap)
__device__ __forceinline__ Slabs unit_slabs(const UnitK& U, int t) {
    const int Ch = U.C >> 1;
    Slabs s;
    s.f0 = t; s.o0 = 0; s.f1 = t; s.o1 = Ch; s.fb = t; s.ob = 0;
    if (U.mode == 1) {
        if (t > 0 || U.wrap) { s.f0 = sn_prev_frame(t, U.T, U.wrap); s.o0 = Ch; s.f1 = t; s.o1 = 0; s.fb = s.f0; s.ob = Ch; }
        else { s.fb = t; s.ob = 0; }
    } else if (U.mode == 2) {
        if (t < U.T - 1 || U.wrap) { s.f0 = t; s.o0 = Ch; s.f1 = sn_next_frame(t, U.T, U.wrap); s.o1 = 0; s.fb = s.f1; s.ob = 0; }
        else { s.fb = t; s.ob = Ch; }
    }
    return s;
}

// ------------------------------------------------------------------------------------------------------------
// validation op: u = cat(y, shift(hw))
__global__ void gather_kernel(const UnitK U, const int8_t* offs, bf16_t* u, const int CU) {
    const int t = blockIdx.y, Ch = U.C >> 1, hw = U.h * U.w;
    const Slabs s = unit_slabs(U, t);
    const size_t n = (size_t)hw * CU;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / CU), c = (int)(e - (size_t)i * CU);
        bf16_t v = 0;
        if (c < Ch) v = U.x[((ptrdiff_t)s.f0 * hw + i) * U.C + s.o0 + c];
        else if (c < U.C) v = U.x[((ptrdiff_t)s.f1 * hw + i) * U.C + s.o1 + c - Ch];
        else {
            const int k = c - U.C, y = i / U.w, x = i - y * U.w;
            const int sy = y + offs[2 * k], sx = x + offs[2 * k + 1];
            if (sy >= 0 && sy < U.h && sx >= 0 && sx < U.w) v = U.x[(((ptrdiff_t)s.fb * U.h + sy) * U.w + sx) * U.C + s.ob + k];
        }
        u[(size_t)t * n + e] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K0: hw[t][p][k] = sum_tap w1[k][tap] * [p+tap in image] * shifted_k(p+tap),  shifted_k(q) = x[fb][q + off_k][ob + k] or 0
// PP = 8-channel chunks of the 34 x 34 window staged per pass.  Default (SN_K0_PP = 0): the whole borrowed half at once, 79 / 97 KB
// of LDS = two / one 4-wave workgroups per CU.  -DSN_K0_PP=2 stages two chunks per pass (42 KB, three workgroups per CU): MEASURED
// slower, 12.4 vs 10.1 ms per window (config 2) and 51.8 vs 48.9 ms (config 3) -- the extra barrier pair and staging prologue per
// pass cost more than the added occupancy returns; kept as a compile-time shape.
#ifndef SN_K0_PP
#define SN_K0_PP 0
#endif
template <int CH, int PP>
__global__ __launch_bounds__(256) void shiftconv_kernel(const UnitK U, const XcdTiles G, const int8_t* __restrict__ offs,
                                                      const uint32_t* __restrict__ w1d, bf16_t* hw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RW = 34, PSB = PP * 16 + 4;      // odd number of dwords per pixel: lanes = pixels hit distinct banks
    constexpr int PCS = CH / 8;
    int t, ty_, tx_;
    if (!sn_xcd_tile(G, t, ty_, tx_)) return;      // XCD-aware walk: the 34-wide windows of x-neighbours overlap by 18 columns
    const int tid = threadIdx.x, y0 = ty_ * 16, x0 = tx_ * 16;
    const Slabs s = unit_slabs(U, t);
    const bf16_t* src = U.x + (ptrdiff_t)s.fb * U.h * U.w * U.C + s.ob;
    const int px = tid & 15, py = tid >> 4, oy = y0 + py, ox = x0 + px;
    const bool valid = oy < U.h && ox < U.w;
    // interior tiles (every tap position p+tap of every output pixel is inside the image): no conv-padding masks at all,
    // and each tap is ONE v_dot2c on the zero-extended bf16 value with a packed weight word (bf16 weight in the low half)
    const bool interior = y0 >= 1 && x0 >= 1 && y0 + 16 < U.h && x0 + 16 < U.w;
#pragma unroll
    for (int pc0 = 0; pc0 < PCS; pc0 += PP) {
        const int np = PCS - pc0 < PP ? PCS - pc0 : PP;           // chunks of this pass (compile-time after unrolling)
        {   // staging: issue ALL global loads first (branch-free, clamped addresses), then mask + write to LDS: one memory
            // round trip per pass instead of one per loop iteration
            constexpr int NIT = (RW * RW * PP + 255) / 256;
            uint4 v[NIT];
            int lo[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256;
                const int pix = idx / np, pc = idx - pix * np;
                const int ry = pix / RW, rx = pix - ry * RW;
                const int gy = y0 - 9 + ry, gx = x0 - 9 + rx;
                const bool in = idx < RW * RW * np && gy >= 0 && gy < U.h && gx >= 0 && gx < U.w;
                lo[k] = idx < RW * RW * np ? (in ? pix * PSB + pc * 16 : -(pix * PSB + pc * 16) - 1) : 0x7fffffff;
                v[k] = *(const uint4*)(src + (in ? ((size_t)gy * U.w + gx) * U.C + (pc0 + pc) * 8 : 0));
            }
            if (pc0) __syncthreads();                             // every wave is done reading the previous pass
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                if (lo[k] == 0x7fffffff) continue;
                const bool in = lo[k] >= 0;
                uint32_t* d = (uint32_t*)(smem + (in ? lo[k] : -(lo[k] + 1)));
                d[0] = in ? v[k].x : 0u; d[1] = in ? v[k].y : 0u; d[2] = in ? v[k].z : 0u; d[3] = in ? v[k].w : 0u;
            }
        }
        __syncthreads();
        if (interior) {
            for (int kc = pc0; kc < pc0 + np; ++kc) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kc * 8 + j;
                    const int dy = offs[2 * k], dx = offs[2 * k + 1];
                    const char* base = smem + ((py + 8 + dy) * RW + (px + 8 + dx)) * PSB + (k - pc0 * 8) * 2;
                    float acc = 0.f;
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx)
                            acc = dot2bf((uint32_t)(*(const bf16_t*)(base + (ty * RW + tx) * PSB)), w1d[k * 9 + ty * 3 + tx], acc);
                    o[j] = acc;
                }
                *(uint4*)(hw + (((size_t)t * U.h + oy) * U.w + ox) * CH + kc * 8) = pack8(o);
            }
        } else {
            float m[9];
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int qy = oy + ty - 1, qx = ox + tx - 1;
                    m[ty * 3 + tx] = (qy >= 0 && qy < U.h && qx >= 0 && qx < U.w) ? 1.f : 0.f;
                }
            for (int kc = pc0; kc < pc0 + np; ++kc) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kc * 8 + j;
                    const int dy = offs[2 * k], dx = offs[2 * k + 1];
                    const char* base = smem + ((py + 8 + dy) * RW + (px + 8 + dx)) * PSB + (k - pc0 * 8) * 2;
                    float acc = 0.f;
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx) {
                            const float v = m[ty * 3 + tx] * bf_to_f(*(const bf16_t*)(base + (ty * RW + tx) * PSB));
                            acc = dot2bf(__float_as_uint(v) >> 16, w1d[k * 9 + ty * 3 + tx], acc);   // same bf16 weights as the fast path
                        }
                    o[j] = acc;
                }
                if (valid) *(uint4*)(hw + (((size_t)t * U.h + oy) * U.w + ox) * CH + kc * 8) = pack8(o);
            }
        }
    }
}

#ifdef SN_EXPERIMENTAL   // round-1 five-kernel chain (K1, K2, K3): off the production path, kept for A/B measurements
// ------------------------------------------------------------------------------------------------------------
// K1: LayerNorm over K channels (affine folded into the weights) + 1x1 conv to 2C, operands straight from HBM.
template <int C, bool WITH_HW>
__global__ __launch_bounds__(256) void ln_gemm_kernel(const UnitK U, const bf16_t* __restrict__ hwb, const uint4* __restrict__ wfrag,
                                                    const float* __restrict__ bias, bf16_t* a) {
    constexpr int CH = C / 2, K = WITH_HW ? C + CH : C, KS = (K + 31) / 32, MT = C / 8, NT = 4;
    const int lane = threadIdx.x & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    const int t = blockIdx.y, hw = U.h * U.w;
    const Slabs sl = unit_slabs(U, t);
    const int ibase = blockIdx.x * 256 + wv * 64;

    bf16x8_t B[NT][KS];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int i = ibase + n * 16 + p, ii = i < hw ? i : hw - 1;
        float xv[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk0 = s * 32 + g * 8;
            const bf16_t* src = nullptr;
            if (kk0 < CH) src = U.x + ((ptrdiff_t)sl.f0 * hw + ii) * C + sl.o0 + kk0;
            else if (kk0 < C) src = U.x + ((ptrdiff_t)sl.f1 * hw + ii) * C + sl.o1 + kk0 - CH;
            else if (WITH_HW && kk0 < K) src = hwb + ((size_t)t * hw + ii) * CH + kk0 - C;
            if (src) {
                unpack8(*(const uint4*)src, xv[s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += xv[s][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[s][j] = 0.f;
            }
        }
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / K);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool has = (s * 32 + g * 8) < K;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = has ? xv[s][j] - mean : 0.f;
                xv[s][j] = d; sq += d * d;
            }
        }
        sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / K) + 1e-6f);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[s][j] *= rstd;
            B[n][s] = as_frag(pack8(xv[s]));
        }
    }

#pragma unroll 1
    for (int mp = 0; mp < MT / 2; ++mp) {
        f32x4_t acc0[NT], acc1[NT];
        const float4 b0 = *(const float4*)(bias + g * 4 * MT + (2 * mp) * 4);
        const float4 b1 = *(const float4*)(bias + g * 4 * MT + (2 * mp + 1) * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc0[n] = (f32x4_t){b0.x, b0.y, b0.z, b0.w}; acc1[n] = (f32x4_t){b1.x, b1.y, b1.z, b1.w}; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8_t a0 = as_frag(wfrag[((2 * mp) * KS + s) * 64 + lane]);
            const bf16x8_t a1 = as_frag(wfrag[((2 * mp + 1) * KS + s) * 64 + lane]);
#pragma unroll
            for (int n = 0; n < NT; ++n) { acc0[n] = mfma16(a0, B[n][s], acc0[n]); acc1[n] = mfma16(a1, B[n][s], acc1[n]); }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int i = ibase + n * 16 + p;
            if (i < hw) {
                uint4 o;
                o.x = pack_bf2(acc0[n][0], acc0[n][1]); o.y = pack_bf2(acc0[n][2], acc0[n][3]);
                o.z = pack_bf2(acc1[n][0], acc1[n][1]); o.w = pack_bf2(acc1[n][2], acc1[n][3]);
                *(uint4*)(a + ((size_t)t * hw + i) * (2 * C) + g * 4 * MT + mp * 8) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K2: g1 = (a1 + dw3x3(a1)) * (a2 + dw3x3(a2)); a is stored in 16 B chunks of [4 first-half | 4 partner] positions.
template <int C>
__global__ __launch_bounds__(256) void dw_gate_kernel(const bf16_t* __restrict__ a, const float* __restrict__ wdw, bf16_t* g1,
                                                    float* pool, int h, int w) {
    constexpr int NCH = C / 4, RY = 8, C2 = 2 * C;
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int t = blockIdx.z, y0 = blockIdx.y * RY, x = blockIdx.x * 64 + lane;
    const bf16_t* at = a + (size_t)t * h * w * C2;
    for (int c8 = wv; c8 < NCH; c8 += 4) {
        float wt[9][8];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int j = 0; j < 8; ++j) wt[tp][j] = wdw[tp * C2 + c8 * 8 + j];
        float rows[3][3][8];
        float psum[4] = {0.f, 0.f, 0.f, 0.f};
        auto load_row = [&](int slot, int gy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int gx = x + dx - 1;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = *(const uint4*)(at + ((size_t)gy * w + gx) * C2 + c8 * 8);
                unpack8(v, rows[slot][dx]);
            }
        };
        load_row(0, y0 - 1);
        load_row(1, y0);
#pragma unroll
        for (int yy = 0; yy < RY; ++yy) {
            load_row((yy + 2) % 3, y0 + yy + 1);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += wt[dy * 3 + dx][j] * rows[(yy + dy) % 3][dx][j];
            const int oy = y0 + yy;
            if (oy < h && x < w) {
                const float g0 = o[0] * o[4], g1v = o[1] * o[5], g2v = o[2] * o[6], g3v = o[3] * o[7];
                uint2 q; q.x = pack_bf2(g0, g1v); q.y = pack_bf2(g2v, g3v);
                *(uint2*)(g1 + (((size_t)t * h + oy) * w + x) * C + c8 * 4) = q;
                psum[0] += g0; psum[1] += g1v; psum[2] += g2v; psum[3] += g3v;
            }
        }
        if (pool) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = psum[j];
                s = row_sum16(s);
                s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
                if (lane == 0) {
                    const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
                    pool[((size_t)t * nblk + blk) * C + c8 * 4 + j] = s;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K3: r = dw5x5'(g1) (3x3 and identity folded in) -> LDS -> b = W2 . r on MFMA -> g2 = b1 * sigmoid(b2), channel sums
template <int C>
__global__ __launch_bounds__(256) void dw_gemm_gate_kernel(const bf16_t* __restrict__ g1, const float* __restrict__ ca_in,
                                                         const float* __restrict__ w5, const uint4* __restrict__ wfrag,
                                                         bf16_t* g2, float* pool, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = C / 8, PS = C * 2 + 16, KS = (C + 31) / 32, MT = C / 8, NT = 4, TY = 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    const int t = blockIdx.z, y0 = blockIdx.y * TY, x0 = blockIdx.x * 64;
    const bf16_t* gt = g1 + (size_t)t * h * w * C;
    float* red = (float*)(smem + 256 * PS);

    {   // ---- stencil: lane = pixel column, channel block uniform per wave -> weights are scalar loads ----
        const int x = x0 + lane;
        for (int cb = wv; cb < NB; cb += 4) {
            float sc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sc[j] = ca_in ? ca_in[(size_t)t * C + cb * 8 + j] : 1.f;
            float acc[TY][8];
#pragma unroll
            for (int oy = 0; oy < TY; ++oy)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[oy][j] = 0.f;
#pragma unroll
            for (int iy = 0; iy < TY + 4; ++iy) {
                const int gy = y0 - 2 + iy;
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) {
                    const int gx = x + dx - 2;
                    uint4 q = make_uint4(0, 0, 0, 0);
                    if (gy >= 0 && gy < h && gx >= 0 && gx < w) q = *(const uint4*)(gt + ((size_t)gy * w + gx) * C + cb * 8);
                    float v[8];
                    unpack8(q, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= sc[j];
#pragma unroll
                    for (int oy = 0; oy < TY; ++oy) {
                        const int dy = iy - oy;
                        if (dy >= 0 && dy < 5) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[oy][j] += w5[(dy * 5 + dx) * C + cb * 8 + j] * v[j];
                        }
                    }
                }
            }
#pragma unroll
            for (int oy = 0; oy < TY; ++oy) *(uint4*)(smem + (oy * 64 + lane) * PS + cb * 16) = pack8(acc[oy]);
        }
    }
    __syncthreads();

    // ---- GEMM: wave wv owns tile row wv (64 pixels = 4 N-tiles) ----
    bf16x8_t B[NT][KS];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk0 = s * 32 + g * 8;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (kk0 < C) q = *(const uint4*)(smem + (wv * 64 + n * 16 + p) * PS + kk0 * 2);
            B[n][s] = as_frag(q);
        }
    const int oy = y0 + wv;
#pragma unroll 1
    for (int mp = 0; mp < MT / 2; ++mp) {
        f32x4_t acc0[NT], acc1[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc0[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc1[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8_t a0 = as_frag(wfrag[((2 * mp) * KS + s) * 64 + lane]);
            const bf16x8_t a1 = as_frag(wfrag[((2 * mp + 1) * KS + s) * 64 + lane]);
#pragma unroll
            for (int n = 0; n < NT; ++n) { acc0[n] = mfma16(a0, B[n][s], acc0[n]); acc1[n] = mfma16(a1, B[n][s], acc1[n]); }
        }
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ox = x0 + n * 16 + p;
            if (oy < h && ox < w) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc0[n][r] * sigmoidf_(acc1[n][r]); ps[r] += v[r]; }
                uint2 q; q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]);
                *(uint2*)(g2 + (((size_t)t * h + oy) * w + ox) * C + g * 2 * MT + mp * 4) = q;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = ps[r];
            s = row_sum16(s);
            if (p == 0) red[wv * C + g * 2 * MT + mp * 4 + r] = s;
        }
    }
    __syncthreads();
    if (pool && tid < C) {
        const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
        pool[((size_t)t * nblk + blk) * C + tid] = red[tid] + red[C + tid] + red[2 * C + tid] + red[3 * C + tid];
    }
}

#endif  // SN_EXPERIMENTAL

// ------------------------------------------------------------------------------------------------------------
// K4: y = shortcut + W3' . (ca * g2) (+ bias'), beta folded into W3'/bias'; shortcut = rolled x (CAB2) or x (CAB1)
// waves-per-SIMD target: see the register-budget note in sn_conv.hip (91 VGPRs + 80 AGPRs = 2 waves without it; 148 / 152 = 3 waves)
#ifndef SN_OCC_AGGR
#define SN_OCC_AGGR 0
#endif
// NT = N-tiles (16 pixels) per wave.  NT = 4 (default): 148 / 152 registers, 3 waves per SIMD, the shortcut is loaded after the MFMAs.
// -DSN_K4_NT=2: half the accumulators and operands per wave, the shortcut fetched BEFORE the MFMAs, 6 / 4 waves per SIMD: MEASURED
// 22.6 vs 20.5 ms per window for C = 64 and 69.0 vs 71.2 ms for C = 80 -- twice the weight-fragment loads per pixel eat the gain.
#ifndef SN_K4_NT
#define SN_K4_NT 4
#endif
template <int C, int NT>
__global__ __launch_bounds__(256, NT == 4 ? ((C == 64 && SN_OCC_AGGR) ? 4 : 3) : (C == 64 ? 6 : 4))
void scale_gemm_res_kernel(const UnitK U, const bf16_t* __restrict__ g2, const float* __restrict__ ca,
                           const uint4* __restrict__ wfrag, const float* __restrict__ bias, bf16_t* y) {
    constexpr int CH = C / 2, KS = (C + 31) / 32, MT = C / 16;
    constexpr bool PRE = NT < 4;                 // prefetch the shortcut ahead of the MFMAs
    const int lane = threadIdx.x & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    const int t = blockIdx.y, hw = U.h * U.w;
    const Slabs sl = unit_slabs(U, t);
    const int ibase = blockIdx.x * (64 * NT) + wv * (16 * NT);
    const int c0 = g * 4 * MT;                   // lane (g,p) owns channels [c0, c0 + 4 MT): one contiguous 8*MT-byte run of the shortcut and of y
    const bf16_t* const sbase = c0 < CH ? U.x + (ptrdiff_t)sl.f0 * hw * C + sl.o0 + c0 : U.x + (ptrdiff_t)sl.f1 * hw * C + sl.o1 + c0 - CH;
    uint32_t scp[PRE ? NT : 1][2 * MT];
    if constexpr (PRE) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int i = ibase + n * 16 + p, ii = i < hw ? i : hw - 1;
            const bf16_t* sp = sbase + (size_t)ii * C;
#pragma unroll
            for (int m = 0; m + 1 < MT; m += 2) { const uint4 q = *(const uint4*)(sp + m * 4); scp[n][2 * m] = q.x; scp[n][2 * m + 1] = q.y; scp[n][2 * m + 2] = q.z; scp[n][2 * m + 3] = q.w; }
            if (MT & 1) { const uint2 q = *(const uint2*)(sp + (MT - 1) * 4); scp[n][2 * MT - 2] = q.x; scp[n][2 * MT - 1] = q.y; }
        }
    }

    bf16x8_t B[NT][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk0 = s * 32 + g * 8;
        float cs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] = kk0 < C ? ca[(size_t)t * C + kk0 + j] : 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int i = ibase + n * 16 + p, ii = i < hw ? i : hw - 1;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (kk0 < C) q = *(const uint4*)(g2 + ((size_t)t * hw + ii) * C + kk0);
            float v[8];
            unpack8(q, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= cs[j];
            B[n][s] = as_frag(pack8(v));
        }
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bb = *(const float4*)(bias + g * 4 * MT + m * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4_t){bb.x, bb.y, bb.z, bb.w};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf16x8_t a = as_frag(wfrag[(m * KS + s) * 64 + lane]);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = mfma16(a, B[n][s], acc[m][n]);
        }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int i = ibase + n * 16 + p;
        if (i >= hw) continue;
        uint32_t sc[2 * MT], o[2 * MT];
        if constexpr (PRE) {
#pragma unroll
            for (int k = 0; k < 2 * MT; ++k) sc[k] = scp[n][k];
        } else {
            const bf16_t* sp = sbase + (size_t)i * C;
#pragma unroll
            for (int m = 0; m + 1 < MT; m += 2) { const uint4 q = *(const uint4*)(sp + m * 4); sc[2 * m] = q.x; sc[2 * m + 1] = q.y; sc[2 * m + 2] = q.z; sc[2 * m + 3] = q.w; }
            if (MT & 1) { const uint2 q = *(const uint2*)(sp + (MT - 1) * 4); sc[2 * MT - 2] = q.x; sc[2 * MT - 1] = q.y; }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            o[2 * m] = pack_bf2(bf_lo(sc[2 * m]) + acc[m][n][0], bf_hi(sc[2 * m]) + acc[m][n][1]);
            o[2 * m + 1] = pack_bf2(bf_lo(sc[2 * m + 1]) + acc[m][n][2], bf_hi(sc[2 * m + 1]) + acc[m][n][3]);
        }
        bf16_t* yp = y + ((size_t)t * hw + i) * C + c0;
#pragma unroll
        for (int m = 0; m + 1 < MT; m += 2) *(uint4*)(yp + m * 4) = make_uint4(o[2 * m], o[2 * m + 1], o[2 * m + 2], o[2 * m + 3]);
        if (MT & 1) *(uint2*)(yp + (MT - 1) * 4) = make_uint2(o[2 * MT - 2], o[2 * MT - 1]);
    }
}

UnitK to_k(const sn_unit_src* s) {
    UnitK u; u.x = (const bf16_t*)s->x; u.T = s->T; u.h = s->h; u.w = s->w; u.C = s->C; u.mode = s->mode; u.wrap = s->wrap;
    return u;
}
bool unit_ok(const sn_unit_src* s) {
    return s && s->x && (s->C == 64 || s->C == 80) && s->T > 0 && s->h > 0 && s->w > 0 && s->mode >= 0 && s->mode <= 2;
}

}  // namespace

extern "C" {

int sn_gsts_gather(const sn_unit_src* s, const int8_t* offs, void* u, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !offs || !u || s->mode == 0) return SN_EINVAL;
    hipLaunchKernelGGL(gather_kernel, dim3(1024, s->T), dim3(256), 0, (hipStream_t)stream, to_k(s), offs, (bf16_t*)u, s->C + s->C / 2);
    return sn_check_launch();
}

int sn_temporal_roll(const sn_unit_src* s, void* y, void* stream) {
    sn_clear_error();
    if (!s || !s->x || !y || y == s->x || (s->C & 1) || s->mode < 1 || s->mode > 2 || s->T < 1) return SN_EINVAL;
    hipLaunchKernelGGL(gather_kernel, dim3(1024, s->T), dim3(256), 0, (hipStream_t)stream, to_k(s), (const int8_t*)nullptr, (bf16_t*)y, s->C);
    return sn_check_launch();
}

int sn_gsts_shiftconv(const sn_unit_src* s, const int8_t* offs, const uint32_t* w1, void* hw, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !offs || !w1 || !hw || s->mode == 0) return SN_EINVAL;
    const XcdTiles G = sn_xcd_tiles((s->w + 15) / 16, (s->h + 15) / 16, s->T);
    const dim3 grid = sn_xcd_grid(G);
    if (s->C == 64) {
        constexpr int PP = SN_K0_PP ? SN_K0_PP : 4;
        const size_t lds = 34 * 34 * (PP * 16 + 4);
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)shiftconv_kernel<32, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH;
        hipLaunchKernelGGL((shiftconv_kernel<32, PP>), grid, dim3(256), lds, (hipStream_t)stream, to_k(s), G, offs, w1, (bf16_t*)hw);
    } else {
        constexpr int PP = SN_K0_PP ? SN_K0_PP : 5;
        const size_t lds = 34 * 34 * (PP * 16 + 4);
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)shiftconv_kernel<40, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH;
        hipLaunchKernelGGL((shiftconv_kernel<40, PP>), grid, dim3(256), lds, (hipStream_t)stream, to_k(s), G, offs, w1, (bf16_t*)hw);
    }
    return sn_check_launch();
}

#ifdef SN_EXPERIMENTAL
int sn_ln_gemm(const sn_unit_src* s, const void* hw, const void* wfrag, const float* bias, void* a, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !wfrag || !bias || !a || (s->mode != 0 && !hw)) return SN_EINVAL;
    const int npx = s->h * s->w;
    dim3 grid((npx + 255) / 256, s->T);
    const UnitK u = to_k(s);
    hipStream_t st = (hipStream_t)stream;
    if (s->C == 64) {
        if (s->mode) hipLaunchKernelGGL((ln_gemm_kernel<64, true>), grid, dim3(256), 0, st, u, (const bf16_t*)hw, (const uint4*)wfrag, bias, (bf16_t*)a);
        else hipLaunchKernelGGL((ln_gemm_kernel<64, false>), grid, dim3(256), 0, st, u, (const bf16_t*)hw, (const uint4*)wfrag, bias, (bf16_t*)a);
    } else {
        if (s->mode) hipLaunchKernelGGL((ln_gemm_kernel<80, true>), grid, dim3(256), 0, st, u, (const bf16_t*)hw, (const uint4*)wfrag, bias, (bf16_t*)a);
        else hipLaunchKernelGGL((ln_gemm_kernel<80, false>), grid, dim3(256), 0, st, u, (const bf16_t*)hw, (const uint4*)wfrag, bias, (bf16_t*)a);
    }
    return sn_check_launch();
}

int sn_dwgate_blocks(int h, int w) { return ((h + 7) / 8) * ((w + 63) / 64); }

int sn_dw_gate(const void* a, const float* w, void* g1, float* pool, int T, int h, int w_, int C, void* stream) {
    sn_clear_error();
    if (!a || !w || !g1 || (C != 64 && C != 80)) return SN_EINVAL;
    dim3 grid((w_ + 63) / 64, (h + 7) / 8, T);
    if (C == 64) hipLaunchKernelGGL(dw_gate_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, w, (bf16_t*)g1, pool, h, w_);
    else hipLaunchKernelGGL(dw_gate_kernel<80>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, w, (bf16_t*)g1, pool, h, w_);
    return sn_check_launch();
}

int sn_dwgemm_blocks(int h, int w) { return ((h + 3) / 4) * ((w + 63) / 64); }

int sn_dw_gemm_gate(const void* g1, const float* ca_in, const float* w5, const void* wfrag, void* g2, float* pool,
                    int T, int h, int w, int C, void* stream) {
    sn_clear_error();
    if (!g1 || !w5 || !wfrag || !g2 || (C != 64 && C != 80)) return SN_EINVAL;
    dim3 grid((w + 63) / 64, (h + 3) / 4, T);
    const size_t lds = 256 * (C * 2 + 16) + 4 * C * sizeof(float);
    if (C == 64) hipLaunchKernelGGL(dw_gemm_gate_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)g1, ca_in, w5, (const uint4*)wfrag, (bf16_t*)g2, pool, h, w);
    else hipLaunchKernelGGL(dw_gemm_gate_kernel<80>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)g1, ca_in, w5, (const uint4*)wfrag, (bf16_t*)g2, pool, h, w);
    return sn_check_launch();
}

#endif  // SN_EXPERIMENTAL

int sn_scale_gemm_res(const sn_unit_src* s, const void* g2, const float* ca, const void* wfrag, const float* bias,
                      void* y, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !g2 || !ca || !wfrag || !y || y == s->x) return SN_EINVAL;
    const int npx = s->h * s->w;
    constexpr int PXWG = 64 * SN_K4_NT;
    dim3 grid((npx + PXWG - 1) / PXWG, s->T);
    if (s->C == 64) hipLaunchKernelGGL((scale_gemm_res_kernel<64, SN_K4_NT>), grid, dim3(256), 0, (hipStream_t)stream, to_k(s), (const bf16_t*)g2, ca, (const uint4*)wfrag, bias, (bf16_t*)y);
    else hipLaunchKernelGGL((scale_gemm_res_kernel<80, SN_K4_NT>), grid, dim3(256), 0, (hipStream_t)stream, to_k(s), (const bf16_t*)g2, ca, (const uint4*)wfrag, bias, (bf16_t*)y);
    return sn_check_launch();
}

}  // extern "C"
