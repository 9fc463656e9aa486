// Fused dense CAB of the Shift-Net encoder-decoder (gshift_deblur1.py:141-156: conv3x3 -> PReLU -> conv3x3 -> CALayer -> + x) in TILE form.
// gfx950 only.
//
// The two-launch form (sn_conv2d twice, csrc/sn_conv.hip) moves five tensors per CAB: conv1 reads x and writes mid, conv2 reads mid and x
// and writes out.  Here `mid` never reaches HBM:
//   pass 1  sn_cab_stats   conv1 + PReLU of x, nothing stored but what the CALayer needs: the per-workgroup channel sums of mid and its
//                          first / last rows and columns (conv3_fast_kernel<.., STATS = true> in sn_conv.hip -- the very code that used to
//                          store mid, so the sums and the closed-form scale of sn_cab_ca are bit-identical to the two-launch form);
//   pass 2  sn_cab_fused   one workgroup per TH x 32 output tile: x region (TH+4) x 36 -> LDS, conv1 + PReLU on the (TH+2) x 34 ring the second
//                          conv needs -> bf16 `mid` tile in LDS (zero outside the image: conv2's zero padding), conv2 from that tile, CALayer
//                          scale, + x from the staged region, + the optional second residual, one store.
// Three tensor passes (x, x, out) instead of five.  Operand layouts, k-slot order, accumulation order and every rounding are those of
// conv3_fast_kernel, so the result is BIT-IDENTICAL to the two-launch form (tests/test_gpu_parity.py::test_fused_cab_*).
// The price is arithmetic: conv1 runs twice, the second time on (TH+2)*34 / (TH*32) of the pixels -- 1.33x at TH = 8 -- on matrix cores that
// the two-launch form leaves 80 % idle (profiles/r05_mfma_util_and_traffic_per_kernel_cfg2.json: 14-21 % busy).
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

int sn_cabp_launch(const sn_conv_desc* a, const sn_conv_desc* b, void* stream);      // sn_conv3p.hip

namespace {

struct CabK {
    const bf16_t* x; int h, w;
    const uint4* w1; const float* b1; float prelu; int act;
    const uint4* w2; const float* b2;
    const float* oscale; int oscale_stride;
    const bf16_t* res2; bf16_t* out;
    XcdTiles xg;
};

constexpr __host__ __device__ int cabf_lds_slots(int npb) {        // == sn_lds_slots of sn_conv.hip (pixel stride in 16-byte slots)
    const int k = npb <= 2 ? 2 : 4 * ((npb - 2 + 3) / 4) + 2;
    return k <= npb + 1 ? k : ((npb & 1) ? npb : npb + 1);
}
constexpr int cabf_waves(int mt, int th) {                         // waves per SIMD each instance can reach without spilling
    return mt == 1 ? (th == 8 ? 6 : 3) : mt == 2 ? 4 : mt == 3 ? 2 : 2;
}

// NC1 / NC2: N-tiles (16 pixels) a wave carries through one K walk of conv1 / conv2 (bounds the accumulator / fragment registers)
template <int MT, int CS, int TH, int NC1, int NC2>
__global__ __launch_bounds__(256, cabf_waves(MT, TH)) void cab_fused_kernel(const CabK P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 32, RH = TH + 4, RW = TW + 4, MH = TH + 2, MW = TW + 2, NPB = CS / 8;
    constexpr int PS = 16 * cabf_lds_slots(NPB);
    constexpr int KTOT = 9 * CS, KS = (KTOT + 31) / 32;
    constexpr int ROWP = RW * NPB, NITEM = RH * ROWP, NIT = (NITEM + 255) / 256;
    constexpr int X_BYTES = RH * RW * PS;
    constexpr int NM = MH * MW, NT1 = (NM + 15) / 16, PER1 = (NT1 + 3) / 4;      // conv1: N-tiles of the mid ring, per wave (interleaved)
    constexpr int NT2 = (TH * TW) / 64, XB = TW / 16;                           // conv2: N-tiles per wave
    char* const xs = smem;
    char* const ms = smem + X_BYTES;
    const int tid = threadIdx.x & 255, lane = tid & 63, wv = wave_id();
    const int g = lane >> 4, p = lane & 15;
    int t, tyi, txi;
    if (!sn_xcd_tile(P.xg, t, tyi, txi)) return;
    const int oy0 = tyi * TH, ox0 = txi * TW;

    // ---- x region -> LDS, [pixel][PS]; zero outside the image (conv1's zero padding) -------------------------------------------------
    {
        const int iy0 = oy0 - 2, ix0 = ox0 - 2;
        const bf16_t* inb = P.x + (size_t)t * P.h * P.w * CS;
        const bool interior = iy0 >= 0 && iy0 + RH <= P.h && ix0 >= 0 && ix0 + RW <= P.w;    // workgroup-uniform
        uint4 v[NIT];
        if (interior) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256, idc = idx < NITEM ? idx : NITEM - 1;
                const int r = idc / ROWP, i = idc - r * ROWP;
                v[k] = *(const uint4*)(inb + ((size_t)iy0 * P.w + ix0) * CS + (size_t)r * P.w * CS + i * 8);     // a region row is one contiguous run
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256;
                const int r = idx / ROWP, i = idx - r * ROWP, px = i / NPB, blk = i - px * NPB;
                if (idx < NITEM) *(uint4*)(xs + (r * RW + px) * PS + blk * 16) = v[k];
            }
        } else {
            bool in[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256, idc = idx < NITEM ? idx : NITEM - 1;
                const int r = idc / ROWP, i = idc - r * ROWP, px = i / NPB, blk = i - px * NPB;
                const int gy = iy0 + r, gx = ix0 + px;
                in[k] = gy >= 0 && gy < P.h && gx >= 0 && gx < P.w;
                v[k] = *(const uint4*)(inb + (in[k] ? ((size_t)gy * P.w + gx) * CS + blk * 8 : 0));   // branch-free, clamped
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256;
                const int r = idx / ROWP, i = idx - r * ROWP, px = i / NPB, blk = i - px * NPB;
                if (idx < NITEM) *(uint4*)(xs + (r * RW + px) * PS + blk * 16) = in[k] ? v[k] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    const int c0 = g * 4 * MT;                                      // this lane's 4*MT consecutive channels of its pixel (D layout, prep.rows_natural)
    const float slope = P.prelu;
    const int act = P.act != 1 ? 0 : ((slope >= 0.f && slope <= 1.f) ? 1 : 2);       // as conv3_fast_kernel
    __syncthreads();

    // ---- conv1 + PReLU on the (TH+2) x (TW+2) ring -> mid tile in LDS (bf16, [pixel][PS]); zero outside the image --------------------
#pragma unroll
    for (int ch = 0; ch < (PER1 + NC1 - 1) / NC1; ++ch) {
        int pixbase[NC1], moff[NC1];
        bool live[NC1], inimg[NC1];
#pragma unroll
        for (int n = 0; n < NC1; ++n) {
            const int j = wv + 4 * (ch * NC1 + n);                    // wave-uniform tile index
            const int q = j * 16 + p, qc = q < NM ? q : NM - 1;
            const int my = qc / MW, mx = qc - my * MW;
            pixbase[n] = (my * RW + mx) * PS;
            moff[n] = qc * PS;
            live[n] = (ch * NC1 + n < PER1) && q < NM;
            const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
            inimg[n] = gy >= 0 && gy < P.h && gx >= 0 && gx < P.w;
        }
        f32x4_t acc[MT][NC1];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 b4 = P.b1 ? *(const float4*)(P.b1 + c0 + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NC1; ++n) acc[m][n] = (f32x4_t){b4.x, b4.y, b4.z, b4.w};
        }
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
            for (int m = 0; m < MT; ++m) a[m] = as_frag(P.w1[(m * KS + s) * 64 + lane]);
            bf16x8_t b[NC1];
#pragma unroll
            for (int n = 0; n < NC1; ++n) b[n] = as_frag(*(const uint4*)(xs + pixbase[n] + toff));
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NC1; ++n) acc[m][n] = mfma16(a[m], b[n], acc[m][n]);
        }
#pragma unroll
        for (int n = 0; n < NC1; ++n) {
            if (ch * NC1 + n >= PER1) continue;                       // compile time
            uint32_t wd[MT][2];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
                if (act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], slope * v[r]);
                } else if (act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(slope, fminf(v[r], 0.f), fmaxf(v[r], 0.f));
                }
                wd[m][0] = inimg[n] ? pack_bf2(v[0], v[1]) : 0u;
                wd[m][1] = inimg[n] ? pack_bf2(v[2], v[3]) : 0u;
            }
            if (live[n]) {
                char* dst = ms + moff[n] + c0 * 2;
                if constexpr (MT == 2 || MT == 4) {
#pragma unroll
                    for (int m = 0; m < MT; m += 2)
                        if (c0 + m * 4 < CS) *(uint4*)(dst + m * 8) = make_uint4(wd[m][0], wd[m][1], wd[m + 1][0], wd[m + 1][1]);
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        if (c0 + m * 4 < CS) *(uint2*)(dst + m * 8) = make_uint2(wd[m][0], wd[m][1]);
                }
            }
        }
    }
    __syncthreads();

    // ---- conv2 from the mid tile; epilogue: CALayer scale, + x (staged region), + res2, store ----------------------------------------
    const bool full = (oy0 + TH <= P.h) && (ox0 + TW <= P.w);             // workgroup-uniform: no bounds masks at all
    const size_t tbase = (((size_t)t * P.h + oy0) * P.w + ox0) * CS;
    bf16_t* const outb = P.out + tbase;
    const bf16_t* const res2b = P.res2 ? P.res2 + tbase : nullptr;
    float4 osc[MT];
    const bool has_osc = P.oscale != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m)
        osc[m] = has_osc ? *(const float4*)(P.oscale + (size_t)t * P.oscale_stride + c0 + m * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int ch = 0; ch < (NT2 + NC2 - 1) / NC2; ++ch) {
        int pixbase[NC2], loff[NC2], xoff[NC2];
        bool valid[NC2];
#pragma unroll
        for (int n = 0; n < NC2; ++n) {
            const int nl = ch * NC2 + n < NT2 ? ch * NC2 + n : NT2 - 1;
            const int nn = wv * NT2 + nl, row = nn / XB, xb = nn - row * XB;
            pixbase[n] = (row * MW + xb * 16 + p) * PS;
            xoff[n] = ((row + 2) * RW + xb * 16 + p + 2) * PS + c0 * 2;
            loff[n] = (row * P.w + xb * 16 + p) * CS + c0;
            valid[n] = full || ((oy0 + row < P.h) && (ox0 + xb * 16 + p < P.w));
        }
        f32x4_t acc[MT][NC2];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 b4 = P.b2 ? *(const float4*)(P.b2 + c0 + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NC2; ++n) acc[m][n] = (f32x4_t){b4.x, b4.y, b4.z, b4.w};
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            int toff = 0;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int kk0 = (s * 4 + gg) * 8;
                const int tap = kk0 / CS, cc0 = kk0 - tap * CS, dy = tap / 3, dx = tap - dy * 3;
                const int o = kk0 < KTOT ? (dy * MW + dx) * PS + cc0 * 2 : 0;
                toff = g == gg ? o : toff;
            }
            bf16x8_t a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = as_frag(P.w2[(m * KS + s) * 64 + lane]);
            bf16x8_t b[NC2];
#pragma unroll
            for (int n = 0; n < NC2; ++n) b[n] = as_frag(*(const uint4*)(ms + pixbase[n] + toff));
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NC2; ++n) acc[m][n] = mfma16(a[m], b[n], acc[m][n]);
        }
#pragma unroll
        for (int n = 0; n < NC2; ++n) {
            if (ch * NC2 + n >= NT2) continue;                        // compile time
            float v[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                v[m][0] = acc[m][n][0]; v[m][1] = acc[m][n][1]; v[m][2] = acc[m][n][2]; v[m][3] = acc[m][n][3];
                if (has_osc) { v[m][0] *= osc[m].x; v[m][1] *= osc[m].y; v[m][2] *= osc[m].z; v[m][3] *= osc[m].w; }
            }
            // + x: the lane's channels of its own pixel, from the staged region (the two-launch form re-reads x from memory here)
            if constexpr (MT == 2 || MT == 4) {
#pragma unroll
                for (int m = 0; m < MT; m += 2)
                    if (c0 + m * 4 < CS) {
                        const uint4 rr = *(const uint4*)(xs + xoff[n] + m * 8);
                        v[m][0] += bf_lo(rr.x); v[m][1] += bf_hi(rr.x); v[m][2] += bf_lo(rr.y); v[m][3] += bf_hi(rr.y);
                        v[m + 1][0] += bf_lo(rr.z); v[m + 1][1] += bf_hi(rr.z); v[m + 1][2] += bf_lo(rr.w); v[m + 1][3] += bf_hi(rr.w);
                    }
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (c0 + m * 4 < CS) {
                        const uint2 rr = *(const uint2*)(xs + xoff[n] + m * 8);
                        v[m][0] += bf_lo(rr.x); v[m][1] += bf_hi(rr.x); v[m][2] += bf_lo(rr.y); v[m][3] += bf_hi(rr.y);
                    }
            }
            if (!valid[n]) continue;
            if (res2b) {                                             // wave-uniform
                const bf16_t* src = res2b + loff[n];
                if constexpr (MT == 2 || MT == 4) {
#pragma unroll
                    for (int m = 0; m < MT; m += 2)
                        if (c0 + m * 4 < CS) {
                            const uint4 rr = *(const uint4*)(src + m * 4);
                            v[m][0] += bf_lo(rr.x); v[m][1] += bf_hi(rr.x); v[m][2] += bf_lo(rr.y); v[m][3] += bf_hi(rr.y);
                            v[m + 1][0] += bf_lo(rr.z); v[m + 1][1] += bf_hi(rr.z); v[m + 1][2] += bf_lo(rr.w); v[m + 1][3] += bf_hi(rr.w);
                        }
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        if (c0 + m * 4 < CS) {
                            const uint2 rr = *(const uint2*)(src + m * 4);
                            v[m][0] += bf_lo(rr.x); v[m][1] += bf_hi(rr.x); v[m][2] += bf_lo(rr.y); v[m][3] += bf_hi(rr.y);
                        }
                }
            }
            bf16_t* dst = outb + loff[n];
            if constexpr (MT == 2 || MT == 4) {
#pragma unroll
                for (int m = 0; m < MT; m += 2)
                    if (c0 + m * 4 < CS)
                        *(uint4*)(dst + m * 4) = make_uint4(pack_bf2(v[m][0], v[m][1]), pack_bf2(v[m][2], v[m][3]),
                                                            pack_bf2(v[m + 1][0], v[m + 1][1]), pack_bf2(v[m + 1][2], v[m + 1][3]));
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (c0 + m * 4 < CS) {
                        uint2 o; o.x = pack_bf2(v[m][0], v[m][1]); o.y = pack_bf2(v[m][2], v[m][3]);
                        *(uint2*)(dst + m * 4) = o;
                    }
            }
        }
    }
}

template <int MT, int CS, int TH, int NC1, int NC2>
int launch_cab_fused(const CabK& K, int T, hipStream_t st) {
    constexpr int TW = 32, PS = 16 * cabf_lds_slots(CS / 8);
    CabK P = K; P.xg = sn_xcd_tiles((K.w + TW - 1) / TW, (K.h + TH - 1) / TH, T);
    const dim3 grid = sn_xcd_grid(P.xg);
    const size_t lds = (size_t)((TH + 4) * (TW + 4) + (TH + 2) * (TW + 2)) * PS;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)cab_fused_kernel<MT, CS, TH, NC1, NC2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return SN_ELAUNCH;
    }
    hipLaunchKernelGGL((cab_fused_kernel<MT, CS, TH, NC1, NC2>), grid, dim3(256), lds, st, P);
    return sn_check_launch();
}

// (M-tiles * 1000 + storage channels) of the instances built below, 0: none
int cabf_key(const sn_conv_desc* a, const sn_conv_desc* b) {
    if (!a || !b) return 0;
    const sn_conv_desc* ds[2] = {a, b};
    for (const sn_conv_desc* d : ds)
        if (!(d->k == 3 && d->stride == 1 && d->pad == 1 && d->in_mode == 0 && d->out_mode == 0 && d->n_in == 1 && d->cs_in == d->cs_out &&
              d->ks == (9 * d->cs_in + 31) / 32 && d->h_in == d->h_out && d->w_in == d->w_out && d->wfrag)) return 0;
    if (a->mt != b->mt || a->cs_in != b->cs_in || a->T != b->T || a->h_in != b->h_in || a->w_in != b->w_in) return 0;
    const int key = a->mt * 1000 + a->cs_in;
    return (key == 1016 || key == 2024) ? key : 0;
}

}  // namespace

extern "C" {

int sn_cab_fused_supported(const sn_conv_desc* conv1, const sn_conv_desc* conv2) { return cabf_key(conv1, conv2) != 0 ? 1 : 0; }

int sn_cab_fused(const sn_conv_desc* a, const sn_conv_desc* b, int tile_rows, void* stream) {
    sn_clear_error();
    const int key = cabf_key(a, b);
    if (!key) return SN_EINVAL;
    if (tile_rows == 0) {                                   // the streaming form (csrc/sn_conv3p.hip: cabp_kernel); 8 / 16: the one-workgroup-per-tile form below
        if (!a->in[0] || a->res || a->res2 || a->oscale || b->pool || b->act != 0 || !b->out || b->res != a->in[0] || a->h_in < 2 || a->w_in < 2) return SN_EINVAL;
        return sn_cabp_launch(a, b, stream);
    }
    // conv1: x -> PReLU(conv + bias), nothing else; conv2: mid -> conv (+ bias) * oscale + res (= x) + res2
    if (!a->in[0] || a->res || a->res2 || a->oscale || b->pool || b->act != 0 || !b->out || b->res != a->in[0]) return SN_EINVAL;   // (a->pool / a->out: sn_cab_stats')
    if (b->oscale && b->oscale_stride < 16 * b->mt) return SN_EINVAL;
    if (a->h_in < 2 || a->w_in < 2) return SN_EINVAL;
    CabK K;
    K.x = (const bf16_t*)a->in[0]; K.h = a->h_in; K.w = a->w_in;
    K.w1 = (const uint4*)a->wfrag; K.b1 = a->bias; K.prelu = a->prelu; K.act = a->act;
    K.w2 = (const uint4*)b->wfrag; K.b2 = b->bias; K.oscale = b->oscale; K.oscale_stride = b->oscale_stride;
    K.res2 = (const bf16_t*)b->res2; K.out = (bf16_t*)b->out;
    hipStream_t st = (hipStream_t)stream;
    const bool tall = tile_rows == 16;
    switch (key) {
        case 1016: return tall ? launch_cab_fused<1, 16, 16, 5, 4>(K, a->T, st) : launch_cab_fused<1, 16, 8, 6, 4>(K, a->T, st);
        case 2024: return tall ? launch_cab_fused<2, 24, 16, 5, 4>(K, a->T, st) : launch_cab_fused<2, 24, 8, 3, 4>(K, a->T, st);
        default: return SN_EINVAL;
    }
}

}  // extern "C"
