// Dense convolutions of the Shift-Net encoder-decoder as implicit GEMMs on MFMA, plus the small kernels around
// them (ingest, channel-attention MLP, CAB tail).  gfx950 only.
//
// sn_conv2d: one workgroup (4 waves) owns a TH x TW output tile of one frame.
//   1. the input patch ((TH-1)*stride+k) x ((TW-1)*stride+k) x Cv is staged in LDS as [pixel][Cv] bf16 with the
//      pixel stride padded to an odd multiple of 16 B (conflict-free ds_read_b128 for 16 consecutive pixels);
//      up to three NHWC inputs are interleaved per pixel (torch.cat along channels is never materialised) and the
//      bilinear x2 upsampling of SkipUpSample is done by the loader;
//   2. K = k*k*Cv is walked in steps of 32: a lane's B operand is 8 consecutive channels of one tap of its pixel
//      (one ds_read_b128 at a precomputed tap offset), its A operand one 16 B load of prepacked weights;
//   3. the epilogue works on D registers directly: the host permutes weight rows so that lane (g,p) holds channels
//      [g*4*MT, (g+1)*4*MT) of pixel p -> contiguous NHWC stores; bias, PReLU, residual, pixel-shuffle or NCHW
//      stores and the per-workgroup channel sums for the following CALayer are all fused here.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

// the streaming form of the single-input 3x3 convs (sn_conv3p.hip); sn_conv2d routes to it unless the descriptor asks for the tile kernel
int sn_conv3p_key(const sn_conv_desc* d, bool want_pool);   // want_pool: the caller is about to attach a pool buffer (sn_conv_pool_blocks)
int sn_conv3p_pool_rows(const sn_conv_desc* d);
int sn_conv3p_launch(const sn_conv_desc* d, int lines_len, void* stream);

namespace {

struct ConvK {
    const bf16_t* in0; const bf16_t* in1; const bf16_t* in2;
    int n_in, cs, cv;
    int hin, win, in_mode, k, stride, pad, hout, wout;
    const uint4* wfrag; int ks;
    const float* bias; int act; float prelu;
    const bf16_t* res; bf16_t* out; int cs_out, out_mode, c_out, nchw_dtype, sc_dtype;
    const void* sc; float* pool; const float* oscale; int oscale_stride; const bf16_t* res2;
    int rh, rw, ps;
    int lines_len;                   // conv3_fast_kernel<.., STATS>: `out` is the border-line buffer [T][4][lines_len][cs_out] (sn_cab_stats)
    XcdTiles xg;                     // tile walk of conv_mfma_kernel / conv3_fast_kernel (sn_common.h)
    unsigned m_nblk8, m_rw, m_csb, m_cv, m_k;   // ceil(2^24/d) multipliers: integer division by runtime constants without v_div
};

// smallest k >= npb with k = 2 (mod 4): conflict-free pixel stride (in 16-byte slots) for stride-1 B-operand reads, see conv3_fast_kernel
// (when that costs more than one extra slot -- 24 channels: 6 slots instead of 3 -- the odd count is kept: the doubled LDS footprint
// halves the resident workgroups of a memory-bound kernel, measured 16.3 -> 17.2 ms per window)
constexpr __host__ __device__ int sn_lds_slots(int npb) {
    const int k = npb <= 2 ? 2 : 4 * ((npb - 2 + 3) / 4) + 2;
    return k <= npb + 1 ? k : ((npb & 1) ? npb : npb + 1);
}

__device__ __forceinline__ int fdiv(int x, unsigned magic) { return (int)(((unsigned)x * (unsigned long long)magic) >> 24); }

__device__ __forceinline__ uint4 ld_bilinear(const bf16_t* src, int t, int hs, int ws, int cs, int cb, int gy, int gx) {
    // nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False): src = dst*0.5 - 0.25 clamped at 0
    float sy = gy * 0.5f - 0.25f; if (sy < 0.f) sy = 0.f;
    float sx = gx * 0.5f - 0.25f; if (sx < 0.f) sx = 0.f;
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
    float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const bf16_t* b = src + (size_t)t * hs * ws * cs + cb * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    unpack8(*(const uint4*)(b + ((size_t)y0 * ws + x0) * cs), v00);
    unpack8(*(const uint4*)(b + ((size_t)y0 * ws + x1) * cs), v01);
    unpack8(*(const uint4*)(b + ((size_t)y1 * ws + x0) * cs), v10);
    unpack8(*(const uint4*)(b + ((size_t)y1 * ws + x1) * cs), v11);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
    return pack8(o);
}

// Register budgets.  Without a waves-per-SIMD target hipcc keeps ~64-100 architectural VGPRs and parks the MFMA accumulators in
// AGPRs on top of them: the unified file then holds 108 registers per lane for the 24-channel 3x3 conv (4 waves per SIMD) where
// 72 are needed (7 waves), 144 instead of 110 for the 40-channel one, 171 instead of 148 for K4.  These kernels hide their
// load -> barrier -> MFMA -> store latency ONLY through co-resident workgroups, so every one states the occupancy it can reach
// without spilling (probed per instantiation with -Rpass-analysis=kernel-resource-usage; accepting 2-6 spilled
// registers for one more wave lost time everywhere it was tried).
constexpr int sn_conv_waves(int mt, int th) {         // generic conv: 8x32 / 4x16 tiles; the 16x32 shape is left to the compiler
    return th == 16 ? 1 : (mt == 1 ? 6 : mt == 2 ? 5 : mt == 3 ? 4 : mt == 4 ? 3 : 2);
}
// (feat_extract.0 -- 8 input channels -- and the concatenating 3x3 convs rconcat / conv_hr0 stay on the generic kernel: routing them
// through conv3_fast_kernel measured neutral in round 2, 122.7 vs 122.0 ms and 673.7 vs 664.8 ms.)
// (Round 3 A/B: 4-row tiles for >= 3 M-tiles -- 2 N-tiles per wave: 40 / 48 channels 76-80 registers and 19.6 KB of LDS = 6 waves per SIMD
// instead of 4 / 3, 64 and 80 channels 4 instead of 2 -- measured SLOWER: config 3 679.8 -> 708.1 ms (the 40 / 48-channel convs 141.0 ->
// 160.6 ms, 80-channel 17.5 -> 20.8), config 5 window 464.6 -> 480.5 ms.  The wide convs are not short of occupancy any more; halving
// the tile raises the halo reads 1.33x -> 1.59x and doubles the weight-fragment fetches per pixel, 36-115 KB per workgroup and tile.)
constexpr int sn_conv3_waves(int mt, int cs) {
    return mt == 1 ? 7 : mt == 2 ? 6 : mt == 3 ? (cs == 48 ? 3 : 4) : mt == 4 ? 3 : 2;
}

template <int MT, int TH, int TW>
__global__ __launch_bounds__(256, sn_conv_waves(MT, TH)) void conv_mfma_kernel(const ConvK P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTW = (TH * TW) / 64;      // N-tiles (16 pixels) per wave
    constexpr int XB = TW / 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
    const int g = lane >> 4, p = lane & 15;
    int t, tyi, txi;
    if (!sn_xcd_tile(P.xg, t, tyi, txi)) return;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int tile_bytes = P.rh * P.rw * P.ps;
    int* tapoff = (int*)(smem + tile_bytes);
    float* red = (float*)(smem + tile_bytes + ((P.ks * 4 * 4 + 15) & ~15));

    for (int e = tid; e < P.ks * 4; e += 256) {
        const int kk0 = e * 8, K = P.k * P.k * P.cv;
        int off = 0;
        if (kk0 < K) {
            const int tap = fdiv(kk0, P.m_cv), cc0 = kk0 - tap * P.cv;
            const int dy = fdiv(tap, P.m_k), dx = tap - dy * P.k;
            off = (dy * P.rw + dx) * P.ps + cc0 * 2;
        }
        tapoff[e] = off;
    }
    {
        const int nblk8 = P.cv >> 3, csb = P.cs >> 3, total = P.rh * P.rw * nblk8;
        const int iy0 = oy0 * P.stride - P.pad, ix0 = ox0 * P.stride - P.pad;
        // loads are issued in batches of 4 before any LDS write (clamped addresses, masked afterwards): a load inside a
        // branch is waited for immediately, which serialises one memory round trip per item
        for (int idx0 = tid; idx0 < total; idx0 += 4 * 256) {
            uint4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * 256;
                const int idc = idx < total ? idx : tid;
                const int pix = fdiv(idc, P.m_nblk8), blk = idc - pix * nblk8;
                const int ry = fdiv(pix, P.m_rw), rx = pix - ry * P.rw;
                const int ii = fdiv(blk, P.m_csb), cb = blk - ii * csb;
                const bf16_t* src = ii == 0 ? P.in0 : (ii == 1 ? P.in1 : P.in2);
                const int gy = iy0 + ry, gx = ix0 + rx;
                const bool in = gy >= 0 && gy < P.hin && gx >= 0 && gx < P.win;
                dst[u] = idx < total ? (in ? pix * P.ps + blk * 16 : -(pix * P.ps + blk * 16) - 1) : 0x7fffffff;
                if (P.in_mode == 0) v[u] = *(const uint4*)(src + (in ? (((size_t)t * P.hin + gy) * P.win + gx) * P.cs + cb * 8 : 0));
                else v[u] = in ? ld_bilinear(src, t, P.hin >> 1, P.win >> 1, P.cs, cb, gy, gx) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (dst[u] == 0x7fffffff) continue;
                const bool in = dst[u] >= 0;
                *(uint4*)(smem + (in ? dst[u] : -(dst[u] + 1))) = in ? v[u] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    bf16x8_t a_cur[MT];                      // k-step 0 fragments: in flight across the barrier
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = as_frag(P.wfrag[(m * P.ks) * 64 + lane]);
    __syncthreads();

    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int pixbase[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        pixbase[n] = ((row * P.stride) * P.rw + (xb * 16 + p) * P.stride) * P.ps;
    }
    // Residual operand of the epilogue, fetched NOW (unconditional loads at clamped addresses: a load inside the epilogue's
    // divergent `valid` branch is waited for on the spot -- one HBM round trip per N-tile, four in a row per wave); it lands
    // while the MFMAs run.  Without a residual the lanes read one dummy 8-byte word of the weight buffer.
    // Only for MT = 1 (the full-resolution 16-channel convs): for wider outputs the extra registers cost more occupancy
    // than the hidden round trips win (measured: mt1 22.4 -> 21.0 ms, mt2 19.3 -> 19.8 ms per window).
    constexpr bool PRE_RES = MT == 1;
    uint2 rres[PRE_RES ? MT : 1][PRE_RES ? NTW : 1];
    if constexpr (PRE_RES) {
        const bf16_t* rb = P.res ? P.res : (const bf16_t*)P.wfrag;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
            const int oy = oy0 + row, ox = ox0 + xb * 16 + p;
            const bool valid = P.res && (oy < P.hout) && (ox < P.wout);
            const size_t opix = ((size_t)t * P.hout + oy) * P.wout + ox;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = g * 4 * MT + m * 4;
                rres[m][n] = *(const uint2*)(rb + ((valid && co0 < P.cs_out) ? opix * P.cs_out + co0 : 0));
            }
        }
    }
    // weight fragments one k-step AHEAD (they come from L1/L2; loaded at the point of use every k-step would pay the full
    // load latency, and the trip count is a run-time value, so the compiler cannot pipeline this itself)
    for (int s = 0; s < P.ks; ++s) {
        const int toff = tapoff[s * 4 + g];
        const int sn = s + 1 < P.ks ? s + 1 : s;
        bf16x8_t a_nxt[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nxt[m] = as_frag(P.wfrag[(m * P.ks + sn) * 64 + lane]);
        bf16x8_t b[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n) b[n] = as_frag(*(const uint4*)(smem + pixbase[n] + toff));
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[m][n] = mfma16(a_cur[m], b[n], acc[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    }

    // ---------------- epilogue: lane (g,p) owns channels [g*4*MT, (g+1)*4*MT) of pixel p of each N-tile ------------
    float psum[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) psum[m][r] = 0.f;

    // per-channel epilogue constants once per lane (the stores below may alias them as far as the compiler knows)
    float4 bia[MT], osc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co0 = g * 4 * MT + m * 4;
        bia[m] = P.bias ? *(const float4*)(P.bias + co0) : make_float4(0.f, 0.f, 0.f, 0.f);
        osc[m] = P.oscale ? *(const float4*)(P.oscale + (size_t)t * P.oscale_stride + co0) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        const int oy = oy0 + row, ox = ox0 + xb * 16 + p;
        const bool valid = (oy < P.hout) && (ox < P.wout);
        const size_t opix = ((size_t)t * P.hout + oy) * P.wout + ox;
        uint2 shuf[MT];                                      // out_mode 1: this lane's 4 MT values = all cs_out channels of ONE output pixel
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co0 = g * 4 * MT + m * 4;
            float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
            v[0] += bia[m].x; v[1] += bia[m].y; v[2] += bia[m].z; v[3] += bia[m].w;
            if (P.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * P.prelu;
            }
            v[0] *= osc[m].x; v[1] *= osc[m].y; v[2] *= osc[m].z; v[3] *= osc[m].w;
            if constexpr (PRE_RES) {
                if (P.res && valid && co0 < P.cs_out) {   // out-of-range lanes hold a dummy word: it must reach neither a store nor psum
                    const uint2 rr = rres[m][n];
                    v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
                }
            } else if (P.res && valid && co0 < P.cs_out) {
                const uint2 rr = *(const uint2*)(P.res + opix * P.cs_out + co0);
                v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
            }
            if (P.res2 && valid && co0 < P.cs_out) {       // wave-uniform pointer test; two launches per window use it
                const uint2 rr = *(const uint2*)(P.res2 + opix * P.cs_out + co0);
                v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
            }
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) psum[m][r] += v[r];
                if (P.out_mode == 0) {
                    if (co0 < P.cs_out) {
                        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                        *(uint2*)(P.out + opix * P.cs_out + co0) = o;
                    }
                } else if (P.out_mode == 1) {
                    shuf[m].x = pack_bf2(v[0], v[1]); shuf[m].y = pack_bf2(v[2], v[3]);      // stored below, all M-tiles together
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + r;
                        if (co < P.c_out) {
                            const size_t oi = (((size_t)t * P.c_out + co) * P.hout + oy) * P.wout + ox;
                            const float scv = P.sc_dtype == SN_F32 ? ((const float*)P.sc)[oi]
                                            : (P.sc_dtype == SN_F16 ? __half2float(((const __half*)P.sc)[oi]) : bf_to_f(((const bf16_t*)P.sc)[oi]));
                            if (P.nchw_dtype == SN_F32) ((float*)P.out)[oi] = v[r] + scv;
                            else if (P.nchw_dtype == SN_F16) ((__half*)P.out)[oi] = __float2half(v[r] + scv);
                            else ((bf16_t*)P.out)[oi] = f_to_bf(v[r] + scv);
                        }
                    }
                }
            }
        }
        if (P.out_mode == 1 && valid) {
            // pixel shuffle: the host ordered the rows [sub-pixel 2 i + j][output channel c < cs_out] with cs_out = 4 MT (prep.pack_conv, shuffle), so lane
            // group g holds ALL channels of sub-pixel g of its pixel: cs_out x 2 contiguous bytes, and lane groups (0, 1) / (2, 3) of 16 pixels fill
            // a contiguous run of 32 output pixels (round 5: four 2-byte stores per M-tile to four pixels, 0.6-0.9 TB/s of the conv's own bytes)
            bf16_t* dst = P.out + (((size_t)t * 2 * P.hout + 2 * oy + (g >> 1)) * (2 * P.wout) + 2 * ox + (g & 1)) * P.cs_out;
#pragma unroll
            for (int m = 0; m + 1 < MT; m += 2) *(uint4*)(dst + m * 4) = make_uint4(shuf[m].x, shuf[m].y, shuf[m + 1].x, shuf[m + 1].y);
            if constexpr (MT & 1) *(uint2*)(dst + (MT - 1) * 4) = shuf[MT - 1];
        }
    }

    if (P.pool) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = psum[m][r];
                s = row_sum16(s);
                if (p == 0) red[wv * 16 * MT + g * 4 * MT + m * 4 + r] = s;
            }
        __syncthreads();
        if (tid < 16 * MT) {
            const float s = red[tid] + red[16 * MT + tid] + red[2 * 16 * MT + tid] + red[3 * 16 * MT + tid];
            const int nblk = P.xg.ntx * P.xg.nty, blk = tyi * P.xg.ntx + txi;
            P.pool[((size_t)t * nblk + blk) * (16 * MT) + tid] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Specialisation for the workhorse of the encoder-decoder: 3x3, stride 1, ONE input, no upsampling, NHWC output
// (both convs of every CAB, conv_trans: ~95 % of the dense-conv time).  Same algorithm, operand layouts and epilogue
// arithmetic as conv_mfma_kernel (results are bit identical); what changes is the instruction count:
//   * channels per pixel (CS) and the tile are template constants: every division in the staging plan is by a
//     compile-time constant, the k-loop is fully unrolled (weight fragments are prefetched by the compiler's own
//     scheduling) and there are no in_mode / n_in branches in the staging loop;
//   * a region row is ONE contiguous run of (TW+2)*CS*2 bytes in memory: lane i of the staging loop loads the i-th 16-byte
//     piece of its row, so consecutive lanes read consecutive addresses and the only per-item math is idx -> (row, piece);
//   * tiles whose ring lies inside the image (all but the frame border) take a path without bounds masks.
// One workgroup per tile, deliberately: a PERSISTENT variant with the next tile's region prefetched into registers (round 2, like the
// round-1 attempt on the generic kernel) measured slower -- 16-channel convs 16.5 -> 20.4 ms, 24-channel 15.7 -> 18.2 ms per window of
// config 2: the prefetch and the loop-carried state take the kernel from 32-48 to 107-155 VGPRs, and at 3.8-5.1 TB/s this kernel
// hides its load latency through occupancy (6-8 resident workgroups per CU), not through software pipelining.
// STATS (sn_cab_stats, pass 1 of the fused CAB of csrc/sn_cabf.hip): same tile, same arithmetic, same per-workgroup channel sums, but the
// output is NOT stored -- only its first / last rows and columns, bf16-rounded exactly like the stored tensor, into the small line buffer
// `out` = [T][4][lines_len][cs_out] (row 0, row h-1, column 0, column w-1): everything sn_cab_ca's closed form reads of `mid`.
template <int MT, int CS, int TH, bool STATS = false>
__global__ __launch_bounds__(256, sn_conv3_waves(MT, CS)) void conv3_fast_kernel(const ConvK P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 32, RH = TH + 2, RW = TW + 2, NPB = CS / 8;
    // LDS bytes per pixel = k slots of 16 B with k the smallest value >= CS/8 that is 2 mod 4.  ds_read_b128 is serviced in the lane
    // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md): half of a group are pixels of lane block g, the other
    // half pixels of block g+1 at another tap / channel chunk.  Enumerating the k-steps of every width in use, k = 2 mod 4 costs
    // 1.0-1.4 LDS cycles per group, the odd k that suits 16 CONSECUTIVE lanes costs ~2.0 (tools/lds_stride_cost.py).
    constexpr int PS = 16 * sn_lds_slots(NPB);
    constexpr int KTOT = 9 * CS, KS = (KTOT + 31) / 32;
    constexpr int NTW = (TH * TW) / 64, XB = TW / 16;
    constexpr int ROWP = RW * NPB, NITEM = RH * ROWP, NIT = (NITEM + 255) / 256;
    constexpr int TILE_BYTES = RH * RW * PS;
    const int tid = threadIdx.x & 255, lane = tid & 63, wv = wave_id();     // (& 255: lets the compiler fold the idx < NITEM guards)
    const int g = lane >> 4, p = lane & 15;
    int t, tyi, txi;
    if (!sn_xcd_tile(P.xg, t, tyi, txi)) return;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    float* red = (float*)(smem + TILE_BYTES);

    {
        const int iy0 = oy0 - 1, ix0 = ox0 - 1;
        const bf16_t* inb = P.in0 + (size_t)t * P.hin * P.win * CS;
        const bool interior = iy0 >= 0 && iy0 + RH <= P.hin && ix0 >= 0 && ix0 + RW <= P.win;    // workgroup-uniform
        uint4 v[NIT];
        if (interior) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256, idc = idx < NITEM ? idx : NITEM - 1;
                const int r = idc / ROWP, i = idc - r * ROWP;
                v[k] = *(const uint4*)(inb + ((size_t)iy0 * P.win + ix0) * CS + (size_t)r * P.win * CS + i * 8);     // a region row is one contiguous run
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256;
                const int r = idx / ROWP, i = idx - r * ROWP, px = i / NPB, blk = i - px * NPB;
                if (idx < NITEM) *(uint4*)(smem + (r * RW + px) * PS + blk * 16) = v[k];
            }
        } else {
            bool in[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256, idc = idx < NITEM ? idx : NITEM - 1;
                const int r = idc / ROWP, i = idc - r * ROWP, px = i / NPB, blk = i - px * NPB;
                const int gy = iy0 + r, gx = ix0 + px;
                in[k] = gy >= 0 && gy < P.hin && gx >= 0 && gx < P.win;
                v[k] = *(const uint4*)(inb + (in[k] ? ((size_t)gy * P.win + gx) * CS + blk * 8 : 0));   // branch-free, clamped
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256;
                const int r = idx / ROWP, i = idx - r * ROWP, px = i / NPB, blk = i - px * NPB;
                if (idx < NITEM) *(uint4*)(smem + (r * RW + px) * PS + blk * 16) = in[k] ? v[k] : make_uint4(0, 0, 0, 0);
            }
        }
    }

    // Output / residual addressing: ONE 64-bit wave-uniform base per tensor (scalar registers) plus a 32-bit per-lane element
    // offset (a frame has < 2^31 elements), instead of a 64-bit multiply chain per N-tile.
    const bool full = (oy0 + TH <= P.hout) && (ox0 + TW <= P.wout);           // workgroup-uniform: no bounds masks at all
    const size_t tbase = (((size_t)t * P.hout + oy0) * P.wout + ox0) * P.cs_out;
    bf16_t* const outb = P.out + tbase;
    const bf16_t* const resb = P.res ? P.res + tbase : nullptr;
    const bf16_t* const res2b = P.res2 ? P.res2 + tbase : nullptr;
    const int c0 = g * 4 * MT;                                      // this lane's 4*MT consecutive channels of its pixel
    int loff[NTW];
    bool valid[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        loff[n] = (row * P.wout + xb * 16 + p) * P.cs_out + c0;
        valid[n] = full || ((oy0 + row < P.hout) && (ox0 + xb * 16 + p < P.wout));
    }
    // Residual operand of the epilogue, fetched before the MFMAs (see conv_mfma_kernel); only for the 16-channel convs
    constexpr bool PRE_RES = MT == 1;
    uint2 rres[PRE_RES ? NTW : 1];
    if constexpr (PRE_RES) {
        const bf16_t* rb = resb ? resb : (const bf16_t*)P.wfrag;
#pragma unroll
        for (int n = 0; n < NTW; ++n) rres[n] = *(const uint2*)(rb + ((resb && valid[n] && c0 < P.cs_out) ? loff[n] : 0));
    }
    __syncthreads();

    // accumulators start at the bias (the zero-initialisation they replace cost the same moves; saves the epilogue adds)
    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float4 b4 = P.bias ? *(const float4*)(P.bias + c0 + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4_t){b4.x, b4.y, b4.z, b4.w};
    }
    int pixbase[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        pixbase[n] = (row * RW + xb * 16 + p) * PS;
    }
    // K walk: lane group g reads k-slots [(4s+g)*8, +8) = 8 channels starting at cc0 of tap (dy,dx); compile-time per (s, g)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        int toff = 0;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int kk0 = (s * 4 + gg) * 8;
            const int tap = kk0 / CS, cc0 = kk0 - tap * CS, dy = tap / 3, dx = tap - dy * 3;
            const int o = kk0 < KTOT ? (dy * RW + dx) * PS + cc0 * 2 : 0;
            toff = g == gg ? o : toff;
        }
        bf16x8_t a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = as_frag(P.wfrag[(m * KS + s) * 64 + lane]);
        bf16x8_t b[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n) b[n] = as_frag(*(const uint4*)(smem + pixbase[n] + toff));
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[m][n] = mfma16(a[m], b[n], acc[m][n]);
    }

    // ---------------- epilogue (arithmetic identical to conv_mfma_kernel, out_mode 0 only) ----------------
    float psum[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) psum[m][r] = 0.f;
    float4 osc[MT];
    const bool has_osc = P.oscale != nullptr;                       // wave-uniform: the first conv of a CAB has no output scale
#pragma unroll
    for (int m = 0; m < MT; ++m)
        osc[m] = has_osc ? *(const float4*)(P.oscale + (size_t)t * P.oscale_stride + c0 + m * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float slope = P.prelu;
    // PReLU: for a slope in [0, 1] (every trained / synthetic checkpoint of this family) x >= 0 ? x : a*x == max(x, a*x):
    // two instructions per value; any other slope takes the general max(x,0) + a*min(x,0) form.  Wave-uniform choice.
    const int act = P.act != 1 ? 0 : ((slope >= 0.f && slope <= 1.f) ? 1 : 2);
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        float v[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            v[m][0] = acc[m][n][0]; v[m][1] = acc[m][n][1]; v[m][2] = acc[m][n][2]; v[m][3] = acc[m][n][3];
            if (act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = fmaxf(v[m][r], slope * v[m][r]);
            } else if (act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = fmaf(slope, fminf(v[m][r], 0.f), fmaxf(v[m][r], 0.f));
            }
            if (has_osc) { v[m][0] *= osc[m].x; v[m][1] *= osc[m].y; v[m][2] *= osc[m].z; v[m][3] *= osc[m].w; }
        }
        const bool ok = valid[n];
        if constexpr (PRE_RES) {
            if (resb && ok && c0 < P.cs_out) {
                const uint2 rr = rres[n];
                v[0][0] += bf_lo(rr.x); v[0][1] += bf_hi(rr.x); v[0][2] += bf_lo(rr.y); v[0][3] += bf_hi(rr.y);
            }
        }
        if (ok) {
            const bf16_t* rp[2] = {PRE_RES ? nullptr : resb, res2b};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!rp[q]) continue;                               // wave-uniform
                const bf16_t* src = rp[q] + loff[n];
                if constexpr (MT == 2 || MT == 4) {                 // the lane's run is 16-byte aligned: 16-byte loads
#pragma unroll
                    for (int m = 0; m < MT; m += 2) {
                        if (c0 + m * 4 < P.cs_out) {
                            const uint4 rr = *(const uint4*)(src + m * 4);
                            v[m][0] += bf_lo(rr.x); v[m][1] += bf_hi(rr.x); v[m][2] += bf_lo(rr.y); v[m][3] += bf_hi(rr.y);
                            v[m + 1][0] += bf_lo(rr.z); v[m + 1][1] += bf_hi(rr.z); v[m + 1][2] += bf_lo(rr.w); v[m + 1][3] += bf_hi(rr.w);
                        }
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (c0 + m * 4 < P.cs_out) {
                            const uint2 rr = *(const uint2*)(src + m * 4);
                            v[m][0] += bf_lo(rr.x); v[m][1] += bf_hi(rr.x); v[m][2] += bf_lo(rr.y); v[m][3] += bf_hi(rr.y);
                        }
                    }
                }
            }
            bf16_t* dst = outb + loff[n];
            if (P.pool) {                                           // wave-uniform
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) psum[m][r] += v[m][r];
            }
            if constexpr (STATS) {
                const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
                const int oy = oy0 + row, ox = ox0 + xb * 16 + p;
                bf16_t* const lb = P.out + (size_t)t * 4 * P.lines_len * P.cs_out + c0;
                bf16_t* tgt[4] = {oy == 0 ? lb + (size_t)ox * P.cs_out : nullptr,
                                  oy == P.hout - 1 ? lb + ((size_t)P.lines_len + ox) * P.cs_out : nullptr,
                                  ox == 0 ? lb + ((size_t)2 * P.lines_len + oy) * P.cs_out : nullptr,
                                  ox == P.wout - 1 ? lb + ((size_t)3 * P.lines_len + oy) * P.cs_out : nullptr};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!tgt[q]) continue;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        if (c0 + m * 4 < P.cs_out) {
                            uint2 o; o.x = pack_bf2(v[m][0], v[m][1]); o.y = pack_bf2(v[m][2], v[m][3]);
                            *(uint2*)(tgt[q] + m * 4) = o;
                        }
                }
                continue;
            }
            if constexpr (MT == 2 || MT == 4) {
#pragma unroll
                for (int m = 0; m < MT; m += 2)
                    if (c0 + m * 4 < P.cs_out)
                        *(uint4*)(dst + m * 4) = make_uint4(pack_bf2(v[m][0], v[m][1]), pack_bf2(v[m][2], v[m][3]),
                                                            pack_bf2(v[m + 1][0], v[m + 1][1]), pack_bf2(v[m + 1][2], v[m + 1][3]));
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (c0 + m * 4 < P.cs_out) {
                        uint2 o; o.x = pack_bf2(v[m][0], v[m][1]); o.y = pack_bf2(v[m][2], v[m][3]);
                        *(uint2*)(dst + m * 4) = o;
                    }
            }
        }
    }
    if (P.pool) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sm = psum[m][r];
                sm = row_sum16(sm);
                if (p == 0) red[wv * 16 * MT + g * 4 * MT + m * 4 + r] = sm;
            }
        __syncthreads();
        if (tid < 16 * MT) {
            const float sm = red[tid] + red[16 * MT + tid] + red[2 * 16 * MT + tid] + red[3 * 16 * MT + tid];
            const int nblk = P.xg.ntx * P.xg.nty, blk = tyi * P.xg.ntx + txi;
            P.pool[((size_t)t * nblk + blk) * (16 * MT) + tid] = sm;
        }
    }
}


template <int MT, int CS, bool STATS = false>
int launch_conv3_fast(const ConvK& K, int T, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NPB = CS / 8, PS = 16 * sn_lds_slots(NPB);
    ConvK P = K; P.xg = sn_xcd_tiles((K.wout + TW - 1) / TW, (K.hout + TH - 1) / TH, T);
    const dim3 grid = sn_xcd_grid(P.xg);
    const size_t lds = (size_t)(TH + 2) * (TW + 2) * PS + 4 * 16 * MT * sizeof(float);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)conv3_fast_kernel<MT, CS, TH, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return SN_ELAUNCH;
    }
    hipLaunchKernelGGL((conv3_fast_kernel<MT, CS, TH, STATS>), grid, dim3(256), lds, st, P);
    return sn_check_launch();
}

template <int TH, int TW>
int launch_conv(const ConvK& K, int mt, int T, hipStream_t st) {
    const int rh = (TH - 1) * K.stride + K.k, rw = (TW - 1) * K.stride + K.k;
    ConvK P = K; P.rh = rh; P.rw = rw;
    P.xg = sn_xcd_tiles((K.wout + TW - 1) / TW, (K.hout + TH - 1) / TH, T);
    const dim3 grid = sn_xcd_grid(P.xg);
    auto magic = [](int d) { return (unsigned)(((1u << 24) + d - 1) / d); };
    P.m_nblk8 = magic(P.cv >> 3); P.m_rw = magic(rw); P.m_csb = magic(P.cs >> 3); P.m_cv = magic(P.cv); P.m_k = magic(P.k);
    const size_t lds = (size_t)rh * rw * P.ps + ((P.ks * 16 + 15) & ~15) + 4 * 16 * mt * sizeof(float);
    if (lds > 160 * 1024) return SN_EINVAL;
#define SN_CONV_CASE(M) case M: \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_mfma_kernel<M, TH, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH; \
        hipLaunchKernelGGL((conv_mfma_kernel<M, TH, TW>), grid, dim3(256), lds, st, P); break;
    switch (mt) {
        SN_CONV_CASE(1) SN_CONV_CASE(2) SN_CONV_CASE(3) SN_CONV_CASE(4) SN_CONV_CASE(5) SN_CONV_CASE(6)
        default: return SN_EINVAL;
    }
#undef SN_CONV_CASE
    return sn_check_launch();
}

// ------------------------------------------------------------------------------------------------------------
__global__ void ingest_kernel(const void* src, int dt, const void* noise, uint4* dst, int C, int HW) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; ++c) {
        const size_t idx = ((size_t)t * C + c) * HW + i;
        v[c] = dt == SN_F32 ? ((const float*)src)[idx] : (dt == SN_F16 ? __half2float(((const __half*)src)[idx]) : bf_to_f(((const bf16_t*)src)[idx]));
    }
    if (noise) {
        const size_t idx = (size_t)t * HW + i;
        v[C] = dt == SN_F32 ? ((const float*)noise)[idx] : (dt == SN_F16 ? __half2float(((const __half*)noise)[idx]) : bf_to_f(((const bf16_t*)noise)[idx]));
    }
    dst[(size_t)t * HW + i] = pack8(v);
}

__global__ __launch_bounds__(1024) void ca_mlp_kernel(const float* partial, int nblk, int cpad, int c, int cr, float inv_hw,
                                                   const float* wa, const float* wb, float* ca, unsigned* bad) {
    __shared__ float acc[1024];
    __shared__ float mean[128];
    __shared__ float hid[128];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int nsplit = 1024 / cpad;          // the partial-sum rows of a frame are split over all 16 waves
    const int ch = tid % cpad, part = tid / cpad;
    float s = 0.f;
    if (part < nsplit) {
        const float* pp = partial + (size_t)t * nblk * cpad + ch;
        for (int b = part; b < nblk; b += nsplit) s += pp[(size_t)b * cpad];
    }
    acc[tid] = s;
    __syncthreads();
    if (tid < cpad) {
        float m = 0.f;
        for (int q = 0; q < nsplit; ++q) m += acc[q * cpad + tid];
        sn_flag_nonfinite(bad, m);
        mean[tid] = m * inv_hw;
    }
    __syncthreads();
    if (tid < cr) {
        float h = 0.f;
        for (int j = 0; j < c; ++j) h += wa[tid * c + j] * mean[j];
        hid[tid] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    if (tid < cpad) {
        float o = 0.f;
        if (tid < c) {
            for (int j = 0; j < cr; ++j) o += wb[tid * cr + j] * hid[j];
            o = sigmoidf_(o);
        }
        ca[(size_t)t * cpad + tid] = o;
    }
}

// CALayer of a CAB from the sums of `mid` (see shiftnet_hip.h::sn_cab_ca).  Two launches: the sums (total from conv1's
// per-workgroup partials, first/last row, first/last column) are split over SN_CABCA_NS workgroups per frame -- one
// workgroup per frame was a 40 us latency chain, 101 times per window -- then one workgroup per frame finishes.
#define SN_CABCA_NS 16
__device__ __forceinline__ float cab_ldf(const bf16_t* p) { return bf_to_f(*p); }
__device__ __forceinline__ float cab_ldf(const float* p) { return *p; }
// E = bf16_t (bf16 engine: cs = padded channel count = pixel stride) or float (fp32 engine: cs = channel count = pixel stride)
// L = 0: mid is the whole tensor [T][h][w][cs]; L > 0: mid is the border-line buffer of sn_cab_stats, [T][4][L][cs] = row 0, row h-1, column 0,
// column w-1 (the only pixels of mid these kernels read)
template <typename E>
__device__ __forceinline__ const E* cab_px(const E* mt, int L, int w, int cs, int line, int y, int x) {
    return L > 0 ? mt + ((size_t)line * L + (line < 2 ? x : y)) * cs : mt + ((size_t)y * w + x) * cs;
}
template <typename E>
__global__ __launch_bounds__(256) void cab_ca_part_kernel(const float* partial, int nblk, int cpad, const E* mid, int cs,
                                                        int h, int w, float* scratch, int L) {
    __shared__ float acc[256];
    const int t = blockIdx.y, sidx = blockIdx.x, tid = threadIdx.x;
    const E* mt = mid + (L > 0 ? (size_t)t * 4 * L * cs : (size_t)t * h * w * cs);
    float* out = scratch + ((size_t)t * SN_CABCA_NS + sidx) * 5 * 128;
    {
        const int nsplit = 256 / cpad, ch = tid % cpad, part = tid / cpad;
        float sm = 0.f;
        if (part < nsplit) {
            const float* pp = partial + (size_t)t * nblk * cpad + ch;
            for (int bb = sidx * nsplit + part; bb < nblk; bb += SN_CABCA_NS * nsplit) sm += pp[(size_t)bb * cpad];
        }
        acc[tid] = sm;
        __syncthreads();
        if (tid < cpad) {
            float m = 0.f;
            for (int q = 0; q < nsplit; ++q) m += acc[q * cpad + tid];
            if (tid < 128) out[tid] = m;
        }
        __syncthreads();
    }
    for (int line = 0; line < 4; ++line) {
        const int len = line < 2 ? w : h;
        const int nseg = 256 / cs, ch = tid % cs, seg = tid / cs;
        float sm = 0.f;
        if (seg < nseg)
            for (int i = sidx * nseg + seg; i < len; i += SN_CABCA_NS * nseg) {
                const int y = line == 0 ? 0 : (line == 1 ? h - 1 : i), x = line == 2 ? 0 : (line == 3 ? w - 1 : i);
                sm += cab_ldf(cab_px(mt, L, w, cs, line, y, x) + ch);
            }
        acc[tid] = sm;
        __syncthreads();
        if (tid < cs) {
            float m = 0.f;
            for (int q = 0; q < nseg; ++q) m += acc[q * cs + tid];
            out[(1 + line) * 128 + tid] = m;
        }
        __syncthreads();
    }
}

template <typename E>
__global__ __launch_bounds__(1024) void cab_ca_kernel(const float* scratch, int cpad, const E* mid, int cs, int c, int cr,
                                                     int h, int w, const float* w2, const float* wa, const float* wb, float* ca, int L) {
    __shared__ float acc[1024];
    __shared__ float S[9][128];      // 0 total, 1 row0, 2 row h-1, 3 col0, 4 col w-1, 5..8 corners (0,0) (0,w-1) (h-1,0) (h-1,w-1)
    __shared__ float mean[128];
    __shared__ float hid[128];
    const int t = blockIdx.x, tid = threadIdx.x;
    const E* mt = mid + (L > 0 ? (size_t)t * 4 * L * cs : (size_t)t * h * w * cs);
    if (tid < 5 * 128) {
        const int k = tid >> 7, ch = tid & 127;
        float m = 0.f;
        if (ch < (k == 0 ? cpad : cs)) {
            const float* sp = scratch + (size_t)t * SN_CABCA_NS * 5 * 128 + tid;
            for (int q = 0; q < SN_CABCA_NS; ++q) m += sp[q * 5 * 128];
        }
        S[k][ch] = m;
    }
    if (tid < cs) {
        S[5][tid] = cab_ldf(cab_px(mt, L, w, cs, 0, 0, 0) + tid);
        S[6][tid] = cab_ldf(cab_px(mt, L, w, cs, 0, 0, w - 1) + tid);
        S[7][tid] = cab_ldf(cab_px(mt, L, w, cs, 1, h - 1, 0) + tid);
        S[8][tid] = cab_ldf(cab_px(mt, L, w, cs, 1, h - 1, w - 1) + tid);
    }
    __syncthreads();
    {   // pooled res[co] = (1/hw) sum_ci sum_tap w2[ci][tap][co] * S_tap[ci]; thread = (slice of ci, co), then a tree over slices
        const int nsplit = 1024 / cpad, co = tid % cpad, part = tid / cpad;
        float r = 0.f;
        if (part < nsplit)
            for (int cin = part; cin < c; cin += nsplit) {
                const float tot = S[0][cin], r0 = S[1][cin], r1 = S[2][cin], c0 = S[3][cin], c1 = S[4][cin];
                const float k00 = S[5][cin], k01 = S[6][cin], k10 = S[7][cin], k11 = S[8][cin];
                const float* wk = w2 + (size_t)cin * 9 * cpad + co;
                // tap (ky,kx) reads mid(p + (ky-1, kx-1)): dy=+1 cannot reach row 0, dy=-1 cannot reach row h-1, same for columns
                const float rowex[3] = {r1, 0.f, r0}, colex[3] = {c1, 0.f, c0};
                const float cor[3][3] = {{k11, 0.f, k10}, {0.f, 0.f, 0.f}, {k01, 0.f, k00}};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) r += wk[(ky * 3 + kx) * cpad] * (tot - rowex[ky] - colex[kx] + cor[ky][kx]);
            }
        acc[tid] = r;
        __syncthreads();
        if (tid < cpad) {
            float m = 0.f;
            for (int q = 0; q < nsplit; ++q) m += acc[q * cpad + tid];
            if (tid < 128) mean[tid] = m / ((float)h * (float)w);
        }
    }
    __syncthreads();
    if (tid < cr) {
        float hh = 0.f;
        for (int j = 0; j < c; ++j) hh += wa[tid * c + j] * mean[j];
        hid[tid] = hh > 0.f ? hh : 0.f;
    }
    __syncthreads();
    if (tid < cpad) {
        float o = 0.f;
        if (tid < c) {
            for (int j = 0; j < cr; ++j) o += wb[tid * cr + j] * hid[j];
            o = sigmoidf_(o);
        }
        ca[(size_t)t * cpad + tid] = o;
    }
}

__global__ void selftest_mfma_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x, g = lane >> 4, m = lane & 15;
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = a[m * 32 + g * 8 + j];       // A[m][k], k-slot (g,j) := k = g*8+j
        bv[j] = b[(g * 8 + j) * 16 + m];     // B[k][n = lane&15]
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma16(as_frag(pack8(av)), as_frag(pack8(bv)), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[(g * 4 + r) * 16 + m] = acc[r];   // D[row = g*4+r][col = lane&15]
}

// ------------------------------------------------------------------------------------------------------------
// SkipUpSample after the 1x1: out = bilinear_x2(lo) + res (gshift_deblur1.py:341-350).  The 1x1 conv is linear and the interpolation weights sum
// to one, so conv(up(x)) = up(conv(x)): the conv runs at LOW resolution (a quarter of the pixels, sn_conv2d in_mode 0) and this streaming pass
// does the rest -- the loader of the in_mode-1 conv interpolated cin channels per OUTPUT pixel in front of the MFMAs and moved its bytes at
// 1.3-1.8 TB/s (round 6, tools/conv_labels.py).  One thread = one 8-channel piece of output column x of the row PAIR (2 i, 2 i + 1): six
// low-resolution pieces (rows i - 1, i, i + 1 clamped, the two source columns of x), two residual pieces, two stores, all 16 bytes and
// coalesced over consecutive pieces / columns.  nn.Upsample(scale_factor=2, bilinear, align_corners=False): src = dst / 2 - 0.25 clamped at 0.
__global__ __launch_bounds__(256) void upsample2_add_kernel(const bf16_t* __restrict__ lo, const bf16_t* __restrict__ res, bf16_t* __restrict__ out,
                                                            int hs, int ws, int npb) {
    // grid (pieces of an output row / 256, hs, T): block-uniform row pair, one 32-bit division per thread
    const int w2 = 2 * ws, cs = npb * 8, i = (int)blockIdx.y, t = (int)blockIdx.z;
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (e >= w2 * npb) return;
    const int x = e / npb, piece = e - x * npb;
    float sx = x * 0.5f - 0.25f; if (sx < 0.f) sx = 0.f;
    const int x0 = (int)sx, x1 = min(x0 + 1, ws - 1);
    const float lx = sx - x0, hx = 1.f - lx;
    const int ra = max(i - 1, 0), rc = min(i + 1, hs - 1);
    const bf16_t* b = lo + (size_t)t * hs * ws * cs + piece * 8;
    const uint4 qa0 = *(const uint4*)(b + ((size_t)ra * ws + x0) * cs), qa1 = *(const uint4*)(b + ((size_t)ra * ws + x1) * cs);
    const uint4 qb0 = *(const uint4*)(b + ((size_t)i * ws + x0) * cs), qb1 = *(const uint4*)(b + ((size_t)i * ws + x1) * cs);
    const uint4 qc0 = *(const uint4*)(b + ((size_t)rc * ws + x0) * cs), qc1 = *(const uint4*)(b + ((size_t)rc * ws + x1) * cs);
    const size_t o0 = (((size_t)t * 2 * hs + 2 * i) * w2 + x) * cs + piece * 8, o1 = o0 + (size_t)w2 * cs;
    const uint4 r0 = *(const uint4*)(res + o0), r1 = *(const uint4*)(res + o1);
    float a0[8], a1[8], b0[8], b1[8], c0[8], c1[8], f0[8], f1[8], u0[8], u1[8];
    unpack8(qa0, a0); unpack8(qa1, a1); unpack8(qb0, b0); unpack8(qb1, b1); unpack8(qc0, c0); unpack8(qc1, c1); unpack8(r0, f0); unpack8(r1, f1);
    // row 2 i: source row i - 0.25 -> rows (i - 1, i) with weights (0.25, 0.75), row 0 alone for i = 0 (ra == i: the same numbers);
    // row 2 i + 1: source row i + 0.25 -> rows (i, i + 1) with weights (0.75, 0.25)
    const float ly0 = i > 0 ? 0.75f : 0.f, hy0 = 1.f - ly0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float va = hx * a0[j] + lx * a1[j], vb = hx * b0[j] + lx * b1[j], vc = hx * c0[j] + lx * c1[j];
        u0[j] = (hy0 * va + ly0 * vb) + f0[j];
        u1[j] = (0.75f * vb + 0.25f * vc) + f1[j];
    }
    *(uint4*)(out + o0) = pack8(u0);
    *(uint4*)(out + o1) = pack8(u1);
}

}  // namespace

extern "C" {

int sn_abi_version(void) { return SN_ABI_VERSION; }

int sn_selftest_mfma(const float* a, const float* b, float* d, void* stream) {
    sn_clear_error();
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
    return sn_check_launch();
}

int sn_ingest(const void* src, int dt, const void* noise, void* dst, int T, int C, int H, int W, void* stream) {
    sn_clear_error();
    if (!src || !dst || C < 1 || C + (noise ? 1 : 0) > 8 || dt < 0 || dt > 2) return SN_EINVAL;
    const int hw = H * W;
    hipLaunchKernelGGL(ingest_kernel, dim3((hw + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, src, dt, noise, (uint4*)dst, C, hw);
    return sn_check_launch();
}

// key of the specialised single-input 3x3 path (M-tiles * 1000 + channels), 0: the generic kernel
static int conv3_key(const sn_conv_desc* d) {
    if (!(d->k == 3 && d->stride == 1 && d->pad == 1 && d->in_mode == 0 && d->out_mode == 0 && d->n_in == 1 && d->ks == (9 * d->cs_in + 31) / 32)) return 0;
    const int key = d->mt * 1000 + d->cs_in;
    return (key == 1016 || key == 2024 || key == 4064 || key == 3040 || key == 3048 || key == 5080) ? key : 0;
}
static void conv_tile(const sn_conv_desc* d, int* th, int* tw) {
    if (d->stride == 1) { *tw = 32; *th = 8; }   // (16x32 tiles for narrow convs measured slower: 47.4 vs 44.2 ms per window)
    // stride 2: 8 x 32 tiles while the (17 x 65)-pixel input region stays near 50 KB (<= 24 input channels: three workgroups per CU) -- a wave of a
    // 4 x 16 tile owns ONE N-tile and fetches every weight fragment for a single MFMA; wider inputs keep 4 x 16 (region 106-159 KB at 8 x 32).
    // Bit 15 of flags: 4 x 16 everywhere (round 5's choice; measurements)
    else if (d->n_in * d->cs_in <= 24 && !(d->flags & (1 << 15))) { *th = 8; *tw = 32; }
    else { *th = 4; *tw = 16; }
}

int sn_conv_pool_blocks(const sn_conv_desc* d) {
    if (!d) return SN_EINVAL;
    if (!(d->flags & SN_CONV_TILE_KERNEL)) {                 // the streaming kernel writes one row per (chunk of the tile list, wave) and frame
        const int rows = sn_conv3p_pool_rows(d);
        if (rows > 0) return rows;
    }
    int th, tw; conv_tile(d, &th, &tw);
    return ((d->h_out + th - 1) / th) * ((d->w_out + tw - 1) / tw);
}

int sn_conv2d(const sn_conv_desc* d, void* stream) {
    sn_clear_error();
    if (!d || d->n_in < 1 || d->n_in > 3 || (d->cs_in & 7) || (d->cs_out & 7) || !d->wfrag) return SN_EINVAL;
    if (!d->out) return SN_EINVAL;
    if (d->k < 1 || d->k > 5 || (d->stride != 1 && d->stride != 2) || d->mt < 1 || d->mt > 6 || d->ks < 1) return SN_EINVAL;
    if (d->in_mode == 1 && ((d->h_in | d->w_in) & 1)) return SN_EINVAL;
    if (d->out_mode == 2 && (!d->sc || d->c_out > 4 * d->mt || d->nchw_dtype < 0 || d->nchw_dtype > 2 || d->sc_dtype < 0 || d->sc_dtype > 2)) return SN_EINVAL;
    if (d->ks * 32 < d->k * d->k * d->n_in * d->cs_in) return SN_EINVAL;
    if (d->oscale && d->oscale_stride < 16 * d->mt) return SN_EINVAL;
    if ((d->res || d->res2) && d->out_mode != 0) return SN_EINVAL;
    if (d->out_mode == 1 && d->cs_out != 4 * d->mt) return SN_EINVAL;      // rows ordered [sub-pixel][cs_out channels]: a lane group = one sub-pixel
    ConvK K;
    K.in0 = (const bf16_t*)d->in[0]; K.in1 = (const bf16_t*)d->in[1]; K.in2 = (const bf16_t*)d->in[2];
    K.n_in = d->n_in; K.cs = d->cs_in; K.cv = d->n_in * d->cs_in;
    K.hin = d->h_in; K.win = d->w_in; K.in_mode = d->in_mode; K.k = d->k; K.stride = d->stride; K.pad = d->pad;
    K.hout = d->h_out; K.wout = d->w_out;
    K.wfrag = (const uint4*)d->wfrag; K.ks = d->ks; K.bias = d->bias; K.act = d->act; K.prelu = d->prelu;
    K.res = (const bf16_t*)d->res; K.out = (bf16_t*)d->out; K.cs_out = d->cs_out; K.out_mode = d->out_mode;
    K.c_out = d->c_out; K.nchw_dtype = d->nchw_dtype; K.sc_dtype = d->sc_dtype; K.sc = d->sc; K.pool = d->pool; K.oscale = d->oscale; K.oscale_stride = d->oscale_stride; K.res2 = (const bf16_t*)d->res2;
    const int blocks = K.cv >> 3;
    K.ps = d->stride == 1 ? 16 * sn_lds_slots(blocks) : ((blocks & 1) ? K.cv * 2 : K.cv * 2 + 16);     // stride 2: odd slot count (pixels 2 apart)
    K.rh = K.rw = 0; K.lines_len = 0;
    if (!(d->flags & SN_CONV_TILE_KERNEL) && sn_conv3p_key(d, false)) {
        const int rc = sn_conv3p_launch(d, 0, stream);
        if (rc != SN_EINVAL) return rc;                      // (SN_EINVAL: no device to plan for -- fall through to the tile kernel's own checks)
    }
    int th, tw; conv_tile(d, &th, &tw);
    {   // the specialised single-input 3x3 path; key = M-tiles, channels
        hipStream_t st = (hipStream_t)stream;
        switch (conv3_key(d)) {
            case 1016: return launch_conv3_fast<1, 16>(K, d->T, st);
            case 2024: return launch_conv3_fast<2, 24>(K, d->T, st);
            case 4064: return launch_conv3_fast<4, 64>(K, d->T, st);
            case 3040: return launch_conv3_fast<3, 40>(K, d->T, st);
            case 3048: return launch_conv3_fast<3, 48>(K, d->T, st);
            case 5080: return launch_conv3_fast<5, 80>(K, d->T, st);
            default: break;                       // any other width: the generic kernel below
        }
    }
    if (th == 16) return launch_conv<16, 32>(K, d->mt, d->T, (hipStream_t)stream);
    if (th == 8) return launch_conv<8, 32>(K, d->mt, d->T, (hipStream_t)stream);
    return launch_conv<4, 16>(K, d->mt, d->T, (hipStream_t)stream);
}


int sn_ca_mlp(const float* partial, int nblk, int cpad, int c, int cr, float inv_hw,
              const float* wa, const float* wb, float* ca, int T, unsigned* bad, void* stream) {
    sn_clear_error();
    if (!partial || !wa || !wb || !ca || cpad < 16 || cpad > 128 || c > cpad || cr > 128 || cr < 1 || nblk < 1) return SN_EINVAL;
    hipLaunchKernelGGL(ca_mlp_kernel, dim3(T), dim3(1024), 0, (hipStream_t)stream, partial, nblk, cpad, c, cr, inv_hw, wa, wb, ca, bad);
    return sn_check_launch();
}

int sn_cab_ca_scratch_floats(int T) { return T * SN_CABCA_NS * 5 * 128; }

static int cab_ca_launch(const float* partial, int nblk, int cpad, const void* mid, int lines_len, int cs, int c, int cr, int h, int w,
                         const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream) {
    sn_clear_error();
    if (!partial || !mid || !w2 || !wa || !wb || !ca || !scratch || cpad < 16 || cpad > 128 || cs > 128 || (cs & 7) || c > cs ||
        c > cpad || cr < 1 || cr > 128 || nblk < 1 || h < 2 || w < 2 || (lines_len != 0 && lines_len < (h > w ? h : w))) return SN_EINVAL;
    hipLaunchKernelGGL(cab_ca_part_kernel<bf16_t>, dim3(SN_CABCA_NS, T), dim3(256), 0, (hipStream_t)stream, partial, nblk, cpad,
                       (const bf16_t*)mid, cs, h, w, scratch, lines_len);
    hipLaunchKernelGGL(cab_ca_kernel<bf16_t>, dim3(T), dim3(1024), 0, (hipStream_t)stream, (const float*)scratch, cpad, (const bf16_t*)mid, cs,
                       c, cr, h, w, w2, wa, wb, ca, lines_len);
    return sn_check_launch();
}

int sn_cab_ca(const float* partial, int nblk, int cpad, const void* mid, int cs, int c, int cr, int h, int w,
              const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream) {
    return cab_ca_launch(partial, nblk, cpad, mid, 0, cs, c, cr, h, w, w2, wa, wb, scratch, ca, T, stream);
}

int sn_cab_ca_lines(const float* partial, int nblk, int cpad, const void* lines, int lines_len, int cs, int c, int cr, int h, int w,
                    const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream) {
    if (lines_len < 1) return SN_EINVAL;
    return cab_ca_launch(partial, nblk, cpad, lines, lines_len, cs, c, cr, h, w, w2, wa, wb, scratch, ca, T, stream);
}

// pass 1 of the fused CAB (csrc/sn_cabf.hip): d = the CAB's FIRST conv as sn_conv2d would run it with `pool`, except that d->out is the line buffer
int sn_cab_stats(const sn_conv_desc* d, int lines_len, void* stream) {
    sn_clear_error();
    if (!d || !d->out || !d->pool || !d->wfrag || !d->in[0] || d->res || d->res2 || d->oscale || d->cs_in != d->cs_out) return SN_EINVAL;
    if (lines_len < (d->h_out > d->w_out ? d->h_out : d->w_out) || d->h_out < 2 || d->w_out < 2) return SN_EINVAL;
    if (!(d->flags & SN_CONV_TILE_KERNEL) && sn_conv3p_key(d, false)) {      // the streaming kernel's statistics mode (pool rows: sn_conv_pool_blocks(d))
        const int rc = sn_conv3p_launch(d, lines_len, stream);
        if (rc != SN_EINVAL) return rc;
    }
    const int key = conv3_key(d);
    if (key != 1016 && key != 2024) return SN_EINVAL;
    ConvK K;
    K.in0 = (const bf16_t*)d->in[0]; K.in1 = K.in2 = nullptr;
    K.n_in = 1; K.cs = d->cs_in; K.cv = d->cs_in;
    K.hin = d->h_in; K.win = d->w_in; K.in_mode = 0; K.k = 3; K.stride = 1; K.pad = 1; K.hout = d->h_out; K.wout = d->w_out;
    K.wfrag = (const uint4*)d->wfrag; K.ks = d->ks; K.bias = d->bias; K.act = d->act; K.prelu = d->prelu;
    K.res = nullptr; K.out = (bf16_t*)d->out; K.cs_out = d->cs_out; K.out_mode = 0; K.c_out = d->c_out; K.nchw_dtype = 0; K.sc_dtype = 0; K.sc = nullptr;
    K.pool = d->pool; K.oscale = nullptr; K.oscale_stride = 0; K.res2 = nullptr;
    K.ps = 16 * sn_lds_slots(K.cv >> 3); K.rh = K.rw = 0; K.lines_len = lines_len;
    hipStream_t st = (hipStream_t)stream;
    return key == 1016 ? launch_conv3_fast<1, 16, true>(K, d->T, st) : launch_conv3_fast<2, 24, true>(K, d->T, st);
}

int sn32_cab_ca(const float* partial, int nblk, int cpad, const float* mid, int c, int cr, int h, int w,
                const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream) {
    sn_clear_error();
    if (!partial || !mid || !w2 || !wa || !wb || !ca || !scratch || cpad < 16 || cpad > 128 || c < 1 || c > cpad || cr < 1 || cr > 128 ||
        nblk < 1 || h < 2 || w < 2) return SN_EINVAL;
    hipLaunchKernelGGL(cab_ca_part_kernel<float>, dim3(SN_CABCA_NS, T), dim3(256), 0, (hipStream_t)stream, partial, nblk, cpad, mid, c, h, w, scratch, 0);
    hipLaunchKernelGGL(cab_ca_kernel<float>, dim3(T), dim3(1024), 0, (hipStream_t)stream, (const float*)scratch, cpad, mid, c, c, cr, h, w, w2, wa, wb, ca, 0);
    return sn_check_launch();
}

}  // extern "C"

// out[T][2 hs][2 ws][cs] = bilinear_x2(lo[T][hs][ws][cs]) + res[T][2 hs][2 ws][cs], bf16 NHWC, cs a multiple of 8 (SkipUpSample's tail, see the kernel)
int sn_upsample2_add(const void* lo, const void* res, void* out, int T, int hs, int ws, int cs, void* stream) {
    sn_clear_error();
    if (!lo || !res || !out || T < 1 || hs < 1 || ws < 1 || cs < 8 || (cs & 7)) return SN_EINVAL;
    if (hs > 65535 || T > 65535 || (long)2 * ws * (cs / 8) > 0x7fffffffL) return SN_EINVAL;
    hipLaunchKernelGGL(upsample2_add_kernel, dim3((unsigned)((2 * ws * (cs / 8) + 255) / 256), (unsigned)hs, (unsigned)T), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)lo, (const bf16_t*)res, (bf16_t*)out, hs, ws, cs / 8);
    return sn_check_launch();
}
