// fp32-storage kernels of the Shift-Net path (gfx950): the arithmetic type the reference's denoise CLI runs the
// "+" denoiser in (inference/test_denoise.py:83-85 keeps the module in float32) and the validation build of the
// engine's control flow: every activation is fp32 NHWC [T][H][W][C] (no channel padding), every weight is the
// checkpoint's fp32 value (only re-ordered so that consecutive lanes read consecutive output channels), every
// accumulation is a sequential fp32 FMA chain.  Results match the CPU oracle to fp32 round-off, so a wrong tap, slab,
// gate half or weight row shows up as an error >> 1e-4 instead of hiding inside bf16 noise.
//
// Dense and grouped convolutions run as implicit GEMMs on the matrix cores, in one of two arithmetics chosen per call (sn32_conv_desc.wsplit):
//   exact : v_mfma_f32_16x16x4_f32 (conv32m_kernel): fp32 products and sums, 157 TFLOP/s peak;
//   split : every fp32 operand as bf16 hi + lo, three v_mfma_f32_16x16x32_bf16 per k-step with fp32 accumulation (conv32s_kernel): ~2^-16 per
//           product, 5.3x fewer matrix-core cycles -- the engine's default for the convs it covers; both are within 1e-4 of the CPU oracle.
// Depthwise convolutions and the elementwise operators are direct kernels.  The bf16 kernels in sn_conv.hip / sn_gsts*.hip / sn_phase1.hip are the
// throughput path.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

namespace {

struct Conv32K {
    const float* in0; const float* in1; const float* in2;
    int cin0, cin1, cin2, cs0, cs1, cs2;
    int n_in, T, hin, win, in_mode;
    int k, stride, pad, groups, hout, wout, cout, cin_total;
    const float* w; const float* bias; int act; float prelu;
    const float* oscale; int oscale_stride;
    const float* res; int cs_res;
    void* out; int cs_out, out_mode, nchw_dtype; const void* sc;
    const uint4* wsplit;
    const float* iscale; int iscale_stride;      // NULL or [T][iscale_stride]: the input is multiplied by iscale[t][ci] while it is staged
    const float* rscale; int rscale_stride;      // NULL or [T][rscale_stride]: res is multiplied by rscale[t][co] before it is added
    const float* lnw; const float* lnb;          // NULL or [cin]: LayerNorm2d over the input channels of a pixel while it is staged (split 1x1 kernel)
    float* csum; int csum_cpad;                  // NULL or [T][tiles][csum_cpad]: per-workgroup channel sums of the stored output (split dense 3x3 kernel)
};

__device__ __forceinline__ float ld_bilinear32(const float* src, int hs, int ws, int cs, int c, int gy, int gx) {
    // nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (gshift_deblur1.py:344): src = dst*0.5 - 0.25, clamped at 0
    float sy = gy * 0.5f - 0.25f; if (sy < 0.f) sy = 0.f;
    float sx = gx * 0.5f - 0.25f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = src[((size_t)y0 * ws + x0) * cs + c], v01 = src[((size_t)y0 * ws + x1) * cs + c];
    const float v10 = src[((size_t)y1 * ws + x0) * cs + c], v11 = src[((size_t)y1 * ws + x1) * cs + c];
    return hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// one thread = NCO consecutive output channels of one output pixel (t, oy, ox): every input value is loaded once (a wave
// broadcast) and meets NCO weights from one 16-byte load, so the FMA : load ratio is 2 : 1 instead of 1 : 2.  NCO = 4 whenever the
// channels of a thread stay inside one group (dense convs, the "+" RepConv with 8 outputs per group); NCO = 1 for depthwise
// convs and for widths that are no multiple of 4 (conv_last: 3 outputs).
template <int NCO>
__global__ __launch_bounds__(256) void conv32_kernel(const Conv32K P) {
    const int ncq = P.cout / NCO;
    const size_t n = (size_t)P.T * P.hout * P.wout * ncq;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int co = (int)(e % ncq) * NCO;
        const size_t pix = e / ncq;
        const int ox = (int)(pix % P.wout), oy = (int)((pix / P.wout) % P.hout), t = (int)(pix / ((size_t)P.wout * P.hout));
        const int cin_g = P.cin_total / P.groups, cout_g = P.cout / P.groups, grp = co / cout_g;
        const int hs = P.in_mode == 1 ? P.hin >> 1 : P.hin, ws = P.in_mode == 1 ? P.win >> 1 : P.win;
        float acc[NCO];
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[j] = 0.f;
        auto mac = [&](float xv, const float* wrow) {          // wrow: NCO consecutive output channels of one (tap, ci)
            if constexpr (NCO == 4) {
                const float4 w4 = *(const float4*)wrow;
                acc[0] = fmaf(xv, w4.x, acc[0]); acc[1] = fmaf(xv, w4.y, acc[1]); acc[2] = fmaf(xv, w4.z, acc[2]); acc[3] = fmaf(xv, w4.w, acc[3]);
            } else {
                acc[0] = fmaf(xv, wrow[0], acc[0]);
            }
        };
        for (int ky = 0; ky < P.k; ++ky) {
            const int gy = oy * P.stride - P.pad + ky;
            if (gy < 0 || gy >= P.hin) continue;
            for (int kx = 0; kx < P.k; ++kx) {
                const int gx = ox * P.stride - P.pad + kx;
                if (gx < 0 || gx >= P.win) continue;
                const float* wt = P.w + ((size_t)(ky * P.k + kx) * cin_g) * P.cout + co;
                if (P.groups == 1) {
                    int cbase = 0;
                    for (int ii = 0; ii < P.n_in; ++ii) {
                        const float* src = ii == 0 ? P.in0 : (ii == 1 ? P.in1 : P.in2);
                        const int ci_n = ii == 0 ? P.cin0 : (ii == 1 ? P.cin1 : P.cin2), cs = ii == 0 ? P.cs0 : (ii == 1 ? P.cs1 : P.cs2);
                        const float* fr = src + (size_t)t * hs * ws * cs;
                        if (P.in_mode == 0) {
                            const float* px = fr + ((size_t)gy * ws + gx) * cs;
                            for (int ci = 0; ci < ci_n; ++ci) mac(px[ci], wt + (size_t)(cbase + ci) * P.cout);
                        } else {
                            for (int ci = 0; ci < ci_n; ++ci) mac(ld_bilinear32(fr, hs, ws, cs, ci, gy, gx), wt + (size_t)(cbase + ci) * P.cout);
                        }
                        cbase += ci_n;
                    }
                } else {        // grouped / depthwise: a single input tensor
                    const float* px = P.in0 + (((size_t)t * hs + gy) * ws + gx) * P.cs0 + grp * cin_g;
                    for (int ci = 0; ci < cin_g; ++ci) mac(px[ci], wt + (size_t)ci * P.cout);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NCO; ++j) {
            const int c = co + j;
            float a = acc[j];
            if (P.bias) a += P.bias[c];
            if (P.act == 1) a = a >= 0.f ? a : a * P.prelu;
            if (P.oscale) a *= P.oscale[(size_t)t * P.oscale_stride + c];
            if (P.res) a += P.res[pix * P.cs_res + c] * (P.rscale ? P.rscale[(size_t)t * P.rscale_stride + c] : 1.0f);
            if (P.out_mode == 0) {
                ((float*)P.out)[pix * P.cs_out + c] = a;
            } else if (P.out_mode == 1) {          // F.pixel_shuffle(., 2): out[cc][2y+i][2x+jj] = in[4cc+2i+jj][y][x]
                const int cc = c >> 2, i = (c >> 1) & 1, jj = c & 1;
                ((float*)P.out)[(((size_t)t * 2 * P.hout + 2 * oy + i) * (2 * P.wout) + 2 * ox + jj) * P.cs_out + cc] = a;
            } else {                               // NCHW of the module dtype + the NCHW shortcut
                const size_t oi = (((size_t)t * P.cout + c) * P.hout + oy) * P.wout + ox;
                if (P.nchw_dtype == SN_F32) ((float*)P.out)[oi] = a + ((const float*)P.sc)[oi];
                else if (P.nchw_dtype == SN_F16) ((__half*)P.out)[oi] = __float2half(a + __half2float(((const __half*)P.sc)[oi]));
                else ((bf16_t*)P.out)[oi] = f_to_bf(a + bf_to_f(((const bf16_t*)P.sc)[oi]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 (dense convs incl. the 1x1 / 2x2 / strided / concatenating / upsampling ones,
// and the "+" RepConv with groups = C/8).  Operand roles as in the bf16 kernels: weights = A (M = 16 output channels), activations
// = B (N = 16 pixels).  For this instruction lane l holds A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15] and
// D[m = 4 (l >> 4) + r][n = l & 15], r = 0..3: one float per operand, k = 4 per instruction.  Which input channel a (lane group g4,
// step s) slot means is our choice: within a block of 16 input channels lane group g4 takes channels 4 g4 + s in step s, so ONE
// ds_read_b128 of the staged pixel delivers a lane's B operands of four consecutive MFMAs.
//   workgroup = TH x TW output pixels x up to MTC M-tiles (blockIdx.z picks the M-chunk), 4 waves, NTW = TH TW / 64 N-tiles per wave;
//   input patch staged 16 channels at a time in LDS as [pixel][24 floats] (16 + 8 pad: conflict-free ds_read_b128 lane groups),
//   zero-filled outside the image and beyond the channel count (so ragged widths 3, 14, 18, 22 ... need no special case);
//   weights straight from global memory (L1 / L2 resident), one float per lane and MFMA, fetched one tap ahead;
//   grouped convs: an M-tile = two groups of 8 outputs whose inputs are the SAME 16 channels: block-diagonal A (half of it zero).
constexpr int C32_CB = 32, C32_PSL = C32_CB + 8;      // input channels staged per barrier; LDS floats per pixel (40: conflict-free ds_read_b128 lane groups)
// Epilogue shared by the two matrix-core conv kernels: lane (g4, p) holds output channels co0 + 16 m + 4 g4 + r of pixel p of each N-tile
// (same arithmetic as conv32_kernel: bias, PReLU, per-frame scale, residual; NHWC / pixel-shuffle / NCHW + shortcut outputs).
template <int MTC, int NTW, int XB>
__device__ __forceinline__ void conv32_epilogue(const Conv32K& P, const f32x4_t (&acc)[MTC][NTW], int t, int oy0, int ox0, int co0, int mt_n, int wv, int g4, int p, float (&sum)[MTC][4]) {
    const bool vout = P.out_mode == 0 && (P.cout & 3) == 0 && (P.cs_out & 3) == 0 && ((size_t)P.out & 15) == 0 &&
                      (!P.res || ((P.cs_res & 3) == 0 && ((size_t)P.res & 15) == 0));
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        const int oy = oy0 + row, ox = ox0 + xb * 16 + p;
        if (oy >= P.hout || ox >= P.wout) continue;
        const size_t pix = ((size_t)t * P.hout + oy) * P.wout + ox;
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (m >= mt_n) continue;
            const int c0 = co0 + 16 * m + 4 * g4;
            if (c0 >= P.cout) continue;
            if (vout) {                                                                    // four consecutive channels: 16-byte residual load and store
                float a[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
                if (P.bias) { const float4 bb = *(const float4*)(P.bias + c0); a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w; }
                if (P.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = a[r] >= 0.f ? a[r] : a[r] * P.prelu;
                }
                if (P.oscale) {
                    const float* os = P.oscale + (size_t)t * P.oscale_stride + c0;
                    a[0] *= os[0]; a[1] *= os[1]; a[2] *= os[2]; a[3] *= os[3];
                }
                if (P.res) {
                    float4 rr = *(const float4*)(P.res + pix * P.cs_res + c0);
                    if (P.rscale) { const float4 q = *(const float4*)(P.rscale + (size_t)t * P.rscale_stride + c0); rr.x *= q.x; rr.y *= q.y; rr.z *= q.z; rr.w *= q.w; }
                    a[0] += rr.x; a[1] += rr.y; a[2] += rr.z; a[3] += rr.w;
                }
                *(float4*)((float*)P.out + pix * P.cs_out + c0) = make_float4(a[0], a[1], a[2], a[3]);
                sum[m][0] += a[0]; sum[m][1] += a[1]; sum[m][2] += a[2]; sum[m][3] += a[3];
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + r;
                if (c >= P.cout) continue;
                float a = acc[m][n][r];
                if (P.bias) a += P.bias[c];
                if (P.act == 1) a = a >= 0.f ? a : a * P.prelu;
                if (P.oscale) a *= P.oscale[(size_t)t * P.oscale_stride + c];
                if (P.res) a += P.res[pix * P.cs_res + c] * (P.rscale ? P.rscale[(size_t)t * P.rscale_stride + c] : 1.0f);
                sum[m][r] += a;
                if (P.out_mode == 0) {
                    ((float*)P.out)[pix * P.cs_out + c] = a;
                } else if (P.out_mode == 1) {
                    const int cc = c >> 2, i = (c >> 1) & 1, jj = c & 1;
                    ((float*)P.out)[(((size_t)t * 2 * P.hout + 2 * oy + i) * (2 * P.wout) + 2 * ox + jj) * P.cs_out + cc] = a;
                } else {
                    const size_t oi = (((size_t)t * P.cout + c) * P.hout + oy) * P.wout + ox;
                    if (P.nchw_dtype == SN_F32) ((float*)P.out)[oi] = a + ((const float*)P.sc)[oi];
                    else if (P.nchw_dtype == SN_F16) ((__half*)P.out)[oi] = __float2half(a + __half2float(((const __half*)P.sc)[oi]));
                    else ((bf16_t*)P.out)[oi] = f_to_bf(a + bf_to_f(((const bf16_t*)P.sc)[oi]));
                }
            }
        }
    }
}

template <int MTC, int TH, int TW>
__global__ __launch_bounds__(256) void conv32m_kernel(const Conv32K P) {
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    constexpr int NTW = (TH * TW) / 64, XB = TW / 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g4 = lane >> 4, p = lane & 15;
    const int tiles_y = (P.hout + TH - 1) / TH;
    const int t = blockIdx.y / tiles_y, ty = blockIdx.y - t * tiles_y, tx = blockIdx.x;
    const int oy0 = ty * TH, ox0 = tx * TW, co0 = blockIdx.z * MTC * 16;
    const int rh = (TH - 1) * P.stride + P.k, rw = (TW - 1) * P.stride + P.k, npatch = rh * rw;
    const int iy0 = oy0 * P.stride - P.pad, ix0 = ox0 * P.stride - P.pad;
    const int hs = P.in_mode == 1 ? P.hin >> 1 : P.hin, ws = P.in_mode == 1 ? P.win >> 1 : P.win;
    const bool grouped = P.groups > 1;
    const int cin_g = P.cin_total / P.groups;                       // dense: all input channels; grouped: 8
    const int mt_n = min(MTC, (P.cout - co0 + 15) / 16);            // M-tiles of this workgroup that exist
    // channel blocks this workgroup walks: dense -> ceil(cin / 32) blocks of 32 feeding every M-tile; grouped -> block cb = the 16 input
    // channels of M-tile cb's two groups (they are the M-tile's own channel range), feeding that M-tile only
    const int ncb = grouped ? mt_n : (P.cin_total + C32_CB - 1) / C32_CB;
    const int nsub = grouped ? 1 : 2;                               // 16-channel sub-blocks (4 MFMA steps each) per staged block
    // single-input tensors whose pixel stride and channel counts are multiples of 4 are staged with 16-byte loads
    const bool vec = P.n_in == 1 && P.in_mode == 0 && (P.cs0 & 3) == 0 && (P.cin_total & 3) == 0 && ((size_t)P.in0 & 15) == 0;

    f32x4_t acc[MTC][NTW];
#pragma unroll
    for (int m = 0; m < MTC; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int pixbase[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        pixbase[n] = ((row * P.stride) * rw + (xb * 16 + p) * P.stride) * C32_PSL + 4 * g4;
    }
    const int ntap = P.k * P.k;
    const size_t tapstride = (size_t)cin_g * P.cout;               // weights: [tap][cin per group][cout]
    for (int cb = 0; cb < ncb; ++cb) {
        const int cbase = grouped ? co0 + 16 * cb : cb * C32_CB;    // first input channel of the block
        const int nch = grouped ? 16 : C32_CB;
        __syncthreads();                                            // everybody is done reading the previous block
        if (vec) {                                                  // element = (pixel, 4 consecutive channels)
            const int q4 = nch >> 2;
            for (int e = tid; e < npatch * q4; e += 256) {
                const int pix = e / q4, c = (e - pix * q4) * 4, ci = cbase + c;
                const int ry = pix / rw, rx = pix - ry * rw, gy = iy0 + ry, gx = ix0 + rx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < P.cin_total && gy >= 0 && gy < P.hin && gx >= 0 && gx < P.win)
                    v = *(const float4*)(P.in0 + (((size_t)t * hs + gy) * ws + gx) * P.cs0 + ci);
                if (P.iscale && ci < P.cin_total) {
                    const float4 sc4 = *(const float4*)(P.iscale + (size_t)t * P.iscale_stride + ci);
                    v.x *= sc4.x; v.y *= sc4.y; v.z *= sc4.z; v.w *= sc4.w;
                }
                *(float4*)(smem32 + pix * C32_PSL + c) = v;
            }
        } else {
            for (int e = tid; e < npatch * nch; e += 256) {
                const int pix = e / nch, c = e - pix * nch, ci = cbase + c;
                const int ry = pix / rw, rx = pix - ry * rw, gy = iy0 + ry, gx = ix0 + rx;
                float v = 0.f;
                if (ci < P.cin_total && gy >= 0 && gy < P.hin && gx >= 0 && gx < P.win) {
                    const float* src = P.in0; int cc = ci, cs = P.cs0;                    // torch.cat of up to three inputs along channels
                    if (P.n_in > 1 && cc >= P.cin0) { cc -= P.cin0; src = P.in1; cs = P.cs1; if (P.n_in > 2 && cc >= P.cin1) { cc -= P.cin1; src = P.in2; cs = P.cs2; } }
                    const float* fr = src + (size_t)t * hs * ws * cs;
                    v = P.in_mode == 0 ? fr[((size_t)gy * ws + gx) * cs + cc] : ld_bilinear32(fr, hs, ws, cs, cc, gy, gx);
                }
                smem32[pix * C32_PSL + c] = v;
            }
        }
        __syncthreads();
        const int m_lo = grouped ? cb : 0, m_hi = grouped ? cb + 1 : mt_n;                // M-tiles this block contributes to
        for (int sb = 0; sb < nsub; ++sb) {
            // a 16-channel sub-block past the last input channel contributes only zeros: skip its 4 MFMA steps per tap (36 -> 36 channels:
            // 3 of 4 sub-blocks; 80 input channels: 5 of 6).  Workgroup-uniform.
            if (!grouped && cbase + 16 * sb >= P.cin_total) break;
            // this lane's A operands: output channel co0 + 16 m + p (A row = lane & 15), input channels 4 g4 + s of the 16-channel sub-block
            const float* wp[MTC];                                                         // address of (tap 0, s = 0), or null: zero operand
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                wp[m] = nullptr;
                const int co = co0 + 16 * m + p;
                if (m < m_lo || m >= m_hi || co >= P.cout) continue;
                if (grouped) { if ((g4 >> 1) == (p >> 3)) wp[m] = P.w + (size_t)((4 * g4) & 7) * P.cout + co; }     // the M-tile's other group: zero
                else { const int ci = cbase + 16 * sb + 4 * g4; if (ci < P.cin_total) wp[m] = P.w + (size_t)ci * P.cout + co; }
            }
            const int cleft = grouped ? 4 : P.cin_total - (cbase + 16 * sb + 4 * g4);     // channels of this lane's quad that exist
            auto wload = [&](int tap, float (&d)[MTC][4]) {
#pragma unroll
                for (int m = 0; m < MTC; ++m)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) d[m][s2] = (wp[m] && s2 < cleft) ? wp[m][tap * tapstride + (size_t)s2 * P.cout] : 0.f;
            };
            float wa[MTC][4], wn[MTC][4];
            wload(0, wa);
            for (int tap = 0; tap < ntap; ++tap) {
                wload(tap + 1 < ntap ? tap + 1 : tap, wn);                                 // next tap's weights: in flight during this tap's MFMAs
                const int dy = tap / P.k, dx = tap - dy * P.k, toff = (dy * rw + dx) * C32_PSL + 16 * sb;
                // (k-step outermost, so that consecutive MFMAs go to different accumulators -- a dependent v_mfma_f32_16x16x4_f32 issues after
                // 40 cycles, an independent one after 32 -- measured SLOWER, bit-identical: 744 -> 791 ms per quadrant forward, the grouped 5x5
                // 1465 -> 1777 us.  The B fragments of all N-tiles are then live at once and the kernel is not MFMA-issue-bound.)
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const float4 b = *(const float4*)(smem32 + pixbase[n] + toff);
#pragma unroll
                    for (int m = 0; m < MTC; ++m) {
                        if (m < m_lo || m >= m_hi) continue;                               // workgroup-uniform
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][0], b.x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][1], b.y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][2], b.z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][3], b.w, acc[m][n], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < MTC; ++m)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) wa[m][s2] = wn[m][s2];
            }
        }
    }
    // ---- epilogue: lane (g4, p) holds output channels co0 + 16 m + 4 g4 + r of pixel p of each N-tile (same arithmetic as conv32_kernel) ----
    const bool vout = P.out_mode == 0 && (P.cout & 3) == 0 && (P.cs_out & 3) == 0 && ((size_t)P.out & 15) == 0 &&
                      (!P.res || ((P.cs_res & 3) == 0 && ((size_t)P.res & 15) == 0));
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        const int oy = oy0 + row, ox = ox0 + xb * 16 + p;
        if (oy >= P.hout || ox >= P.wout) continue;
        const size_t pix = ((size_t)t * P.hout + oy) * P.wout + ox;
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (m >= mt_n) continue;
            const int c0 = co0 + 16 * m + 4 * g4;
            if (c0 >= P.cout) continue;
            if (vout) {                                                                    // four consecutive channels: 16-byte residual load and store
                float a[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
                if (P.bias) { const float4 bb = *(const float4*)(P.bias + c0); a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w; }
                if (P.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = a[r] >= 0.f ? a[r] : a[r] * P.prelu;
                }
                if (P.oscale) {
                    const float* os = P.oscale + (size_t)t * P.oscale_stride + c0;
                    a[0] *= os[0]; a[1] *= os[1]; a[2] *= os[2]; a[3] *= os[3];
                }
                if (P.res) { const float4 rr = *(const float4*)(P.res + pix * P.cs_res + c0); a[0] += rr.x; a[1] += rr.y; a[2] += rr.z; a[3] += rr.w; }
                *(float4*)((float*)P.out + pix * P.cs_out + c0) = make_float4(a[0], a[1], a[2], a[3]);
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + r;
                if (c >= P.cout) continue;
                float a = acc[m][n][r];
                if (P.bias) a += P.bias[c];
                if (P.act == 1) a = a >= 0.f ? a : a * P.prelu;
                if (P.oscale) a *= P.oscale[(size_t)t * P.oscale_stride + c];
                if (P.res) a += P.res[pix * P.cs_res + c];
                if (P.out_mode == 0) {
                    ((float*)P.out)[pix * P.cs_out + c] = a;
                } else if (P.out_mode == 1) {
                    const int cc = c >> 2, i = (c >> 1) & 1, jj = c & 1;
                    ((float*)P.out)[(((size_t)t * 2 * P.hout + 2 * oy + i) * (2 * P.wout) + 2 * ox + jj) * P.cs_out + cc] = a;
                } else {
                    const size_t oi = (((size_t)t * P.cout + c) * P.hout + oy) * P.wout + ox;
                    if (P.nchw_dtype == SN_F32) ((float*)P.out)[oi] = a + ((const float*)P.sc)[oi];
                    else if (P.nchw_dtype == SN_F16) ((__half*)P.out)[oi] = __float2half(a + __half2float(((const __half*)P.sc)[oi]));
                    else ((bf16_t*)P.out)[oi] = f_to_bf(a + bf_to_f(((const bf16_t*)P.sc)[oi]));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Split-precision convolution: fp32 in, fp32 out, the products on the BF16 matrix-core instruction.  Every fp32 operand is written as
// hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits together) and the product as  w x ~ wh xh + wh xl + wl xh : three
// v_mfma_f32_16x16x32_bf16 (K = 32, fp32 accumulation) replace EIGHT v_mfma_f32_16x16x4_f32 of twice their issue time each, i.e. 5.3x
// fewer matrix-core cycles for an error of ~2^-16 per product (the dropped wl xl term) -- an order of magnitude finer than the TF32
// convolutions PyTorch runs by default on the reference's GPUs.  Activations are split ONCE per staged element (when the tile goes to
// LDS: hi image | lo image per pixel, the same 4 bytes per value as fp32), the weights on the host (prep.pack_conv32_split: A fragments).
//   dense (k = 1 / 3, stride 1): a channel block of 32 input channels is ONE k-step; k-step index = tap * blocks + block.
//   grouped (the "+" RepConv, 8 -> 8 per group, k = 5): block = M-tile = 16 channels (two groups, block-diagonal A); a k-step covers the
//   taps 2s, 2s + 1 (13 k-steps for k = 5, 5 for k = 3; lane group g reads tap 2s + (g >> 1), channel half g & 1), as sn_grp5_gemm_gate
//   does in bf16.
#ifndef SN_C32S_TH3         // tile height of the dense 3x3 instances: 4 rows = 33 KB of LDS (4 workgroups per CU) or 8 rows = 54 KB (2)
#define SN_C32S_TH3 4
#endif
template <int MTC, int KSZ, bool GROUPED>
__global__ __launch_bounds__(256, 2) void conv32s_kernel(const Conv32K P) {
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    char* const lds = (char*)smem32;
    constexpr int TH = (KSZ == 3 && !GROUPED) ? SN_C32S_TH3 : 8, TW = 32, NTW = (TH * TW) / 64, XB = TW / 16;
    constexpr int RH = TH + KSZ - 1, RW = TW + KSZ - 1, NPATCH = RH * RW, NTAP = KSZ * KSZ;
    constexpr int NCH = GROUPED ? 16 : 32;                          // channels of a staged block
    constexpr int PS = GROUPED ? 80 : 160;                          // LDS bytes per pixel: hi | lo | pad (5 / 10 slots of 16 B)
    constexpr int LO = NCH * 2;                                     // byte offset of the lo image inside a pixel
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g4 = lane >> 4, p = lane & 15;
    const int tiles_y = (P.hout + TH - 1) / TH;
    const int t = blockIdx.y / tiles_y, ty = blockIdx.y - t * tiles_y, tx = blockIdx.x;
    const int oy0 = ty * TH, ox0 = tx * TW, co0 = blockIdx.z * MTC * 16;
    const int iy0 = oy0 - P.pad, ix0 = ox0 - P.pad;
    const int mt_all = (P.cout + 15) / 16;
    const int mt_n = min(MTC, mt_all - blockIdx.z * MTC);
    const int ncb = GROUPED ? mt_n : (P.cin_total + 31) / 32;
    constexpr int KSG = (NTAP + 1) / 2;                             // grouped: two taps per k-step
    const int ks_all = GROUPED ? KSG : NTAP * ncb;                  // k-steps per M-tile in the weight array

    f32x4_t acc[MTC][NTW];
#pragma unroll
    for (int m = 0; m < MTC; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int pixbase[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nn = wv * NTW + n, row = nn / XB, xb = nn - row * XB;
        pixbase[n] = (row * RW + xb * 16 + p) * PS;
    }
    const uint4* const wh = P.wsplit;                               // [2 parts][mt_all][ks_all][64 lanes]
    const uint4* const wl = P.wsplit + (size_t)mt_all * ks_all * 64;
    for (int cb = 0; cb < ncb; ++cb) {
        const int cbase = GROUPED ? co0 + 16 * cb : cb * 32;
        __syncthreads();                                            // everybody is done reading the previous block
        constexpr int Q = NCH / 4;                                  // float4 quads per pixel
        constexpr int NIT = (NPATCH * Q + 255) / 256;
        // all loads of the block first (one memory round trip, not one per iteration), then split and write
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256, ec = e < NPATCH * Q ? e : 0;
            const int pix = ec / Q, q = ec - pix * Q, ci = cbase + 4 * q;
            const int ry = pix / RW, rx = pix - ry * RW, gy = iy0 + ry, gx = ix0 + rx;
            const bool in = e < NPATCH * Q && ci < P.cin_total && gy >= 0 && gy < P.hin && gx >= 0 && gx < P.win;
            // up to three inputs concatenated along channels (conv_hr0, rconcat): a quad never straddles two of them (every c_in % 4 == 0)
            const float* src = P.in0; int cs = P.cs0, cl = ci;
            if (ci >= P.cin0) { src = P.in1; cs = P.cs1; cl = ci - P.cin0; }
            if (ci >= P.cin0 + P.cin1) { src = P.in2; cs = P.cs2; cl = ci - P.cin0 - P.cin1; }
            v[it] = *(const float4*)(in ? src + (((size_t)t * P.hin + gy) * P.win + gx) * cs + cl : P.in0);
            if (!in) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            if (e >= NPATCH * Q) continue;
            const int pix = e / Q, q = e - pix * Q, ci = cbase + 4 * q;
            float4 x = v[it];
            if (P.iscale && ci < P.cin_total) {                       // CALayer scale of the producer, applied here instead of in a pass of its own
                const float4 sc4 = *(const float4*)(P.iscale + (size_t)t * P.iscale_stride + ci);
                x.x *= sc4.x; x.y *= sc4.y; x.z *= sc4.z; x.w *= sc4.w;
            }
            const uint32_t h01 = pack_bf2(x.x, x.y), h23 = pack_bf2(x.z, x.w);
            const uint32_t l01 = pack_bf2(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf2(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xffff0000u));
            *(uint2*)(lds + pix * PS + 8 * q) = make_uint2(h01, h23);
            *(uint2*)(lds + pix * PS + LO + 8 * q) = make_uint2(l01, l23);
        }
        __syncthreads();
        if constexpr (GROUPED) {
            const int m = cb;                                        // this block feeds M-tile cb only
            const size_t wbase = ((size_t)(blockIdx.z * MTC + m) * KSG) * 64 + lane;
            bf16x8_t ahx = as_frag(wh[wbase]), alx = as_frag(wl[wbase]);
#pragma unroll 1
            for (int s2 = 0; s2 < KSG; ++s2) {
                const int tap = min(2 * s2 + (g4 >> 1), NTAP - 1);   // (tap NTAP does not exist: its weights are zero)
                const int dy = tap / KSZ, dx = tap - dy * KSZ;
                const int toff = (dy * RW + dx) * PS + (g4 & 1) * 16;
                const bf16x8_t ah = ahx, al = alx;
                const int sn = s2 + 1 < KSG ? s2 + 1 : s2;           // next k-step's fragments in flight during this one's MFMAs
                ahx = as_frag(wh[wbase + (size_t)sn * 64]); alx = as_frag(wl[wbase + (size_t)sn * 64]);
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const bf16x8_t bh = as_frag(*(const uint4*)(lds + pixbase[n] + toff)), bl = as_frag(*(const uint4*)(lds + pixbase[n] + toff + LO));
#pragma unroll
                    for (int mm = 0; mm < MTC; ++mm) {
                        if (mm != m) continue;                       // workgroup-uniform
                        acc[mm][n] = mfma16(al, bh, acc[mm][n]);
                        acc[mm][n] = mfma16(ah, bl, acc[mm][n]);
                        acc[mm][n] = mfma16(ah, bh, acc[mm][n]);
                    }
                }
            }
        } else {
            // The A fragments of tap + 1 are fetched while tap's MFMAs run: loaded at the point of use, every tap waited for an L2 round trip
            // (nine per block and workgroup for a 3x3, against ~1.6 us of MFMAs).
            bf16x8_t ahc[MTC], alc[MTC], ahn[MTC], aln[MTC];
            auto wload = [&](int tap, bf16x8_t (&h)[MTC], bf16x8_t (&l)[MTC]) {
#pragma unroll
                for (int m = 0; m < MTC; ++m) {
                    const int mg = min((int)blockIdx.z * MTC + m, mt_all - 1);
                    const size_t wi = ((size_t)mg * ks_all + (size_t)tap * ncb + cb) * 64 + lane;
                    h[m] = as_frag(wh[wi]); l[m] = as_frag(wl[wi]);
                }
            };
            wload(0, ahc, alc);
#pragma unroll 1
            for (int tap = 0; tap < NTAP; ++tap) {
                const int dy = tap / KSZ, dx = tap - dy * KSZ;
                const int toff = (dy * RW + dx) * PS + g4 * 16;
                wload(tap + 1 < NTAP ? tap + 1 : tap, ahn, aln);
#pragma unroll
                for (int m = 0; m < MTC; ++m) {
                    if (m >= mt_n) continue;                         // workgroup-uniform
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        const bf16x8_t bh = as_frag(*(const uint4*)(lds + pixbase[n] + toff)), bl = as_frag(*(const uint4*)(lds + pixbase[n] + toff + LO));
                        acc[m][n] = mfma16(alc[m], bh, acc[m][n]);
                        acc[m][n] = mfma16(ahc[m], bl, acc[m][n]);
                        acc[m][n] = mfma16(ahc[m], bh, acc[m][n]);
                    }
                }
#pragma unroll
                for (int m = 0; m < MTC; ++m) { ahc[m] = ahn[m]; alc[m] = aln[m]; }
            }
        }
    }
    float sum[MTC][4];
#pragma unroll
    for (int m = 0; m < MTC; ++m) sum[m][0] = sum[m][1] = sum[m][2] = sum[m][3] = 0.f;
    conv32_epilogue<MTC, NTW, XB>(P, acc, t, oy0, ox0, co0, mt_n, wv, g4, p, sum);
    if (P.csum) {                                                   // workgroup-uniform: channel sums of this tile (AdaptiveAvgPool2d of the CAB behind it)
        __syncthreads();                                            // the staged tile is dead: its LDS becomes the [4 waves][MTC * 16] scratch
        float* const red = (float*)lds;
#pragma unroll
        for (int m = 0; m < MTC; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sum[m][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
                if (p == 0) red[wv * MTC * 16 + 16 * m + 4 * g4 + r] = v;
            }
        __syncthreads();
        if (tid < MTC * 16) {
            const int c = co0 + tid;
            if (c < P.csum_cpad) {
                float v = 0.f;
                if (c < P.cout)
                    for (int q = 0; q < 4; ++q) v += red[q * MTC * 16 + tid];
                P.csum[(((size_t)t * tiles_y + ty) * gridDim.x + tx) * P.csum_cpad + c] = v;
            }
        }
    }
}

template <int MTC, int KSZ, bool GROUPED>
int launch_conv32s(const Conv32K& K, hipStream_t st) {
    constexpr int TH = (KSZ == 3 && !GROUPED) ? SN_C32S_TH3 : 8, TW = 32;
    const size_t lds = (size_t)(TH + KSZ - 1) * (TW + KSZ - 1) * (GROUPED ? 80 : 160);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv32s_kernel<MTC, KSZ, GROUPED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SN_ELAUNCH;
    const int mt = (K.cout + 15) / 16;
    dim3 grid((K.wout + TW - 1) / TW, ((K.hout + TH - 1) / TH) * K.T, (mt + MTC - 1) / MTC);
    hipLaunchKernelGGL((conv32s_kernel<MTC, KSZ, GROUPED>), grid, dim3(256), lds, st, K);
    return sn_check_launch();
}

// Split-precision 1x1 convolution = a plain GEMM over the flat pixel list: a workgroup takes 128 consecutive pixels, stages ALL their input
// channels once (hi | lo images, 32-channel blocks side by side) and then walks the output M-tiles in groups of four from that one
// staged tile -- conv32s_kernel stages the input again for every group of M-tiles (three times for 160 output channels).
// NHWC fp32 output only, cout / pixel strides multiples of 4 (the caller checks).
#ifndef SN_1X1_NPX
#define SN_1X1_NPX 64         // pixels per workgroup: 64 (30 KB of LDS at 80 input channels: 5 workgroups per CU) or 128
#endif
// GATE: the output channels are consumed pairwise by SimpleGate2 (gshift_deblur1.py:179-182: x1 * sigmoid(x2), x1 = channels [0, C), x2 = [C, 2C)):
// a wave takes the M-tile PAIRS (m, m + C/16), gates in registers and stores the C-channel result -- the 2C-channel tensor is neither written
// nor read back -- and leaves the per-workgroup channel sums of the result for the CALayer2 behind the gate (partial [npix / NPX][cpad];
// a workgroup must not span two frames: hw % NPX == 0, checked by the entry point).
// LOAD 2: the input is [T][hin/2][win/2] and is upsampled x2 bilinearly while it is staged (SkipUpSample, gshift_deblur1.py:341-350: four 16-byte loads per
// staged quad, the arithmetic of ld_bilinear32) -- the exact-product kernel did this with four scalar loads per ELEMENT, 0.66 ms per launch for 1 GB.
// LOAD 1 (LNF): LayerNorm2d (gshift_deblur1.py:19-28) of the input pixel while it is staged -- the eight lanes that load a pixel's 32-channel blocks hold all of
// its channels in registers, so mean and variance (two-pass, as layernorm32_kernel) cost three lane exchanges each and the normalised tensor
// is neither written nor read.
template <int NCB, bool GATE, int LOAD>
__global__ __launch_bounds__(256, 2) void conv32s_1x1_kernel(const Conv32K P, const long long npix, float* partial, const int cpad) {
    constexpr int ncb = NCB;                                         // compile-time: the k-loop unrolls and a group's weight fragments load up front
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    char* const lds = (char*)smem32;
    constexpr int NPX = SN_1X1_NPX;
    const int PSK = ncb * 160 + ((ncb & 1) ? 0 : 32);               // LDS bytes per pixel: ncb blocks of (hi 64 | lo 64 | pad 32); 2 (mod 4) 16-byte slots
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g4 = lane >> 4, p = lane & 15;
    const long long pix0 = (long long)blockIdx.x * NPX;
    const int hw = P.hout * P.wout;
    const int t0 = (int)(pix0 / hw), prem = (int)(pix0 - (long long)t0 * hw);     // frame of the first pixel (a workgroup spans at most two: hw >= 128)
    constexpr bool LNF = LOAD == 1;
    if constexpr (LOAD == 2) {
        const int hs = P.hin >> 1, ws = P.win >> 1;
#pragma unroll
        for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
            for (int it = 0; it < NPX * 8 / 256; ++it) {
                const int e = tid + it * 256, pl = e >> 3, q = e & 7, ci = cb * 32 + 4 * q;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < P.cin_total && pix0 + pl < npix) {
                    int rem = prem + pl, t = t0;
                    if (rem >= hw) { rem -= hw; ++t; }
                    const int gy = rem / P.win, gx = rem - gy * P.win;
                    // nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False): src = dst * 0.5 - 0.25, clamped at 0 (as ld_bilinear32)
                    float sy = gy * 0.5f - 0.25f; if (sy < 0.f) sy = 0.f;
                    float sx = gx * 0.5f - 0.25f; if (sx < 0.f) sx = 0.f;
                    const int y0 = (int)sy, x0 = (int)sx, y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
                    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
                    const float* b = P.in0 + (size_t)t * hs * ws * P.cs0 + ci;
                    const float4 v00 = *(const float4*)(b + ((size_t)y0 * ws + x0) * P.cs0), v01 = *(const float4*)(b + ((size_t)y0 * ws + x1) * P.cs0);
                    const float4 v10 = *(const float4*)(b + ((size_t)y1 * ws + x0) * P.cs0), v11 = *(const float4*)(b + ((size_t)y1 * ws + x1) * P.cs0);
                    x.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
                    x.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
                    x.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
                    x.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
                }
                const uint32_t h01 = pack_bf2(x.x, x.y), h23 = pack_bf2(x.z, x.w);
                const uint32_t l01 = pack_bf2(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xffff0000u));
                const uint32_t l23 = pack_bf2(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xffff0000u));
                *(uint2*)(lds + pl * PSK + cb * 160 + 8 * q) = make_uint2(h01, h23);
                *(uint2*)(lds + pl * PSK + cb * 160 + 64 + 8 * q) = make_uint2(l01, l23);
            }
        }
    } else if constexpr (LNF) {
        constexpr int IT = NPX * 8 / 256;
        const int q = tid & 7;
        float4 v[NCB][IT];
#pragma unroll
        for (int cb = 0; cb < ncb; ++cb)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int pl = (tid + it * 256) >> 3, ci = cb * 32 + 4 * q;
                const long long pg = pix0 + pl;
                const bool in = ci < P.cin_total && pg < npix;
                v[cb][it] = *(const float4*)(P.in0 + (in ? (size_t)pg * P.cs0 + ci : 0));
                if (!in) v[cb][it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        const float rk = 1.0f / (float)P.cin_total;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            float mu = 0.f;
#pragma unroll
            for (int cb = 0; cb < ncb; ++cb) mu += (v[cb][it].x + v[cb][it].y) + (v[cb][it].z + v[cb][it].w);
            mu += __shfl_xor(mu, 1, 64); mu += __shfl_xor(mu, 2, 64); mu += __shfl_xor(mu, 4, 64);
            mu *= rk;
            float var = 0.f;
#pragma unroll
            for (int cb = 0; cb < ncb; ++cb) {
                if (cb * 32 + 4 * q < P.cin_total) {
                    const float dx = v[cb][it].x - mu, dy = v[cb][it].y - mu, dz = v[cb][it].z - mu, dw = v[cb][it].w - mu;
                    var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
            var += __shfl_xor(var, 1, 64); var += __shfl_xor(var, 2, 64); var += __shfl_xor(var, 4, 64);
            const float rstd = 1.0f / sqrtf(var * rk + 1e-6f);
            const int pl = (tid + it * 256) >> 3;
#pragma unroll
            for (int cb = 0; cb < ncb; ++cb) {
                const int ci = cb * 32 + 4 * q;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < P.cin_total) {
                    const float4 g = *(const float4*)(P.lnw + ci), b = *(const float4*)(P.lnb + ci);
                    x = make_float4((v[cb][it].x - mu) * rstd * g.x + b.x, (v[cb][it].y - mu) * rstd * g.y + b.y,
                                    (v[cb][it].z - mu) * rstd * g.z + b.z, (v[cb][it].w - mu) * rstd * g.w + b.w);
                }
                const uint32_t h01 = pack_bf2(x.x, x.y), h23 = pack_bf2(x.z, x.w);
                const uint32_t l01 = pack_bf2(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xffff0000u));
                const uint32_t l23 = pack_bf2(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xffff0000u));
                *(uint2*)(lds + pl * PSK + cb * 160 + 8 * q) = make_uint2(h01, h23);
                *(uint2*)(lds + pl * PSK + cb * 160 + 64 + 8 * q) = make_uint2(l01, l23);
            }
        }
    } else {
#pragma unroll
    for (int cb = 0; cb < ncb; ++cb) {
        float4 v[NPX * 8 / 256], sc[NPX * 8 / 256];
#pragma unroll
        for (int it = 0; it < NPX * 8 / 256; ++it) {                 // the block's loads (data and input scale) first, then split and write
            const int e = tid + it * 256, pl = e >> 3, q = e & 7, ci = cb * 32 + 4 * q;
            const long long pg = pix0 + pl;
            const bool in = ci < P.cin_total && pg < npix;
            v[it] = *(const float4*)(P.in0 + (in ? (size_t)pg * P.cs0 + ci : 0));
            if (!in) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            sc[it] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (P.iscale && in) sc[it] = *(const float4*)(P.iscale + (size_t)(t0 + (prem + pl >= hw ? 1 : 0)) * P.iscale_stride + ci);
        }
#pragma unroll
        for (int it = 0; it < NPX * 8 / 256; ++it) {
            const int e = tid + it * 256, pl = e >> 3, q = e & 7;
            const float4 x = make_float4(v[it].x * sc[it].x, v[it].y * sc[it].y, v[it].z * sc[it].z, v[it].w * sc[it].w);
            const uint32_t h01 = pack_bf2(x.x, x.y), h23 = pack_bf2(x.z, x.w);
            const uint32_t l01 = pack_bf2(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf2(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xffff0000u));
            *(uint2*)(lds + pl * PSK + cb * 160 + 8 * q) = make_uint2(h01, h23);
            *(uint2*)(lds + pl * PSK + cb * 160 + 64 + 8 * q) = make_uint2(l01, l23);
        }
    }
    }
    __syncthreads();
    // The four waves share the workgroup's pixels (NTW N-tiles each of them reads from LDS) and SPLIT THE M-TILES: wave w takes M-tiles
    // w, w + 4, ...  With the pixels split instead, every wave fetched the fragments of ALL M-tiles -- 4 x 60 KB per 64 pixels for 80 -> 160
    // channels, four times the bytes of the activations themselves, through one L1.  The next M-tile's fragments load during this one's MFMAs.
    constexpr int NT = NPX / 16;
    const int mt_all = (P.cout + 15) / 16;
    const uint4* const wh = P.wsplit;                               // [2 parts][mt_all][ncb][64 lanes]
    const uint4* const wl = P.wsplit + (size_t)mt_all * ncb * 64;
    bf16x8_t ahc[NCB], alc[NCB], ahn[NCB], aln[NCB];
    auto wload = [&](int m, bf16x8_t (&h)[NCB], bf16x8_t (&l)[NCB]) {
        const int mc = m < mt_all ? m : mt_all - 1;
#pragma unroll
        for (int cb = 0; cb < ncb; ++cb) {
            const size_t wi = ((size_t)mc * ncb + cb) * 64 + lane;
            h[cb] = as_frag(wh[wi]); l[cb] = as_frag(wl[wi]);
        }
    };
    if constexpr (GATE) {
        const int mh = mt_all >> 1, Cg = P.cout >> 1;                // C = 16 mh
        bf16x8_t bhc[NCB], blc[NCB];
        for (int m = wv; m < mh; m += 4) {
            wload(m, ahc, alc);
            wload(m + mh, bhc, blc);
            f32x4_t acc1[NT], acc2[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) { acc1[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc2[n] = acc1[n]; }
#pragma unroll
            for (int cb = 0; cb < ncb; ++cb)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const char* bp = lds + (n * 16 + p) * PSK + cb * 160 + g4 * 16;
                    const bf16x8_t bh = as_frag(*(const uint4*)bp), bl = as_frag(*(const uint4*)(bp + 64));
                    acc1[n] = mfma16(alc[cb], bh, acc1[n]);
                    acc1[n] = mfma16(ahc[cb], bl, acc1[n]);
                    acc1[n] = mfma16(ahc[cb], bh, acc1[n]);
                    acc2[n] = mfma16(blc[cb], bh, acc2[n]);
                    acc2[n] = mfma16(bhc[cb], bl, acc2[n]);
                    acc2[n] = mfma16(bhc[cb], bh, acc2[n]);
                }
            const int c0 = 16 * m + 4 * g4;
            float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const long long pg = pix0 + n * 16 + p;
                if (pg >= npix) continue;
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { g[r] = acc1[n][r] / (1.0f + expf(-acc2[n][r])); sum[r] += g[r]; }
                *(float4*)((float*)P.out + (size_t)pg * P.cs_out + c0) = make_float4(g[0], g[1], g[2], g[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) sum[r] += __shfl_xor(sum[r], off, 64);
            }
            if (p == 0) *(float4*)(partial + (size_t)blockIdx.x * cpad + c0) = make_float4(sum[0], sum[1], sum[2], sum[3]);
        }
        if (tid < cpad - Cg) partial[(size_t)blockIdx.x * cpad + Cg + tid] = 0.f;
        return;
    }
    wload(wv, ahc, alc);
    for (int m = wv; m < mt_all; m += 4) {                          // wave-uniform; no barrier inside
        wload(m + 4, ahn, aln);
        f32x4_t acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < ncb; ++cb)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const char* bp = lds + (n * 16 + p) * PSK + cb * 160 + g4 * 16;
                const bf16x8_t bh = as_frag(*(const uint4*)bp), bl = as_frag(*(const uint4*)(bp + 64));
                acc[n] = mfma16(alc[cb], bh, acc[n]);
                acc[n] = mfma16(ahc[cb], bl, acc[n]);
                acc[n] = mfma16(ahc[cb], bh, acc[n]);
            }
        // epilogue: lane (g4, p) holds output channels 16 m + 4 g4 + r of pixel n * 16 + p
        const int c0 = 16 * m + 4 * g4;
        if (c0 < P.cout) {
            float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (P.bias) bb = *(const float4*)(P.bias + c0);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const long long pg = pix0 + n * 16 + p;
                if (pg >= npix) continue;
                const int t = t0 + (prem + n * 16 + p >= hw ? 1 : 0);
                float a[4] = {acc[n][0] + bb.x, acc[n][1] + bb.y, acc[n][2] + bb.z, acc[n][3] + bb.w};
                if (P.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = a[r] >= 0.f ? a[r] : a[r] * P.prelu;
                }
                if (P.oscale) {
                    const float* os = P.oscale + (size_t)t * P.oscale_stride + c0;
                    a[0] *= os[0]; a[1] *= os[1]; a[2] *= os[2]; a[3] *= os[3];
                }
                if (P.res) { const float4 rr = *(const float4*)(P.res + (size_t)pg * P.cs_res + c0); a[0] += rr.x; a[1] += rr.y; a[2] += rr.z; a[3] += rr.w; }
                *(float4*)((float*)P.out + (size_t)pg * P.cs_out + c0) = make_float4(a[0], a[1], a[2], a[3]);
            }
        }
#pragma unroll
        for (int cb = 0; cb < ncb; ++cb) { ahc[cb] = ahn[cb]; alc[cb] = aln[cb]; }
    }
}

template <int MTC, int TH, int TW>
int launch_conv32m(const Conv32K& K, hipStream_t st) {
    const int rh = (TH - 1) * K.stride + K.k, rw = (TW - 1) * K.stride + K.k;
    const size_t lds = (size_t)rh * rw * C32_PSL * sizeof(float);
    if (lds > 160 * 1024) return SN_EINVAL;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv32m_kernel<MTC, TH, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SN_ELAUNCH;
    const int mt = (K.cout + 15) / 16;
    dim3 grid((K.wout + TW - 1) / TW, ((K.hout + TH - 1) / TH) * K.T, (mt + MTC - 1) / MTC);
    hipLaunchKernelGGL((conv32m_kernel<MTC, TH, TW>), grid, dim3(256), lds, st, K);
    return sn_check_launch();
}

// Depthwise k x k conv (conv1 of CAB2, RepConv2, the depthwise RepConv of the "-s" variants), NHWC output: one thread = 4 consecutive
// channels of one output pixel, 16-byte loads of activations, weights and the residual (the generic direct kernel above handled one
// channel per thread: 2.7 ms for the 160-channel 3x3 at 136 x 224 x 36 against 0.4 ms of memory time).
__global__ __launch_bounds__(256) void dw32_kernel(const Conv32K P) {
    const int c4n = P.cout >> 2;
    const size_t n = (size_t)P.T * P.hout * P.wout * c4n;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % c4n) * 4;
        const size_t pix = e / c4n;
        const int ox = (int)(pix % P.wout), oy = (int)((pix / P.wout) % P.hout), t = (int)(pix / ((size_t)P.wout * P.hout));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < P.k; ++ky) {
            const int gy = oy - P.pad + ky;
            if (gy < 0 || gy >= P.hin) continue;
            for (int kx = 0; kx < P.k; ++kx) {
                const int gx = ox - P.pad + kx;
                if (gx < 0 || gx >= P.win) continue;
                const float4 xv = *(const float4*)(P.in0 + (((size_t)t * P.hin + gy) * P.win + gx) * P.cs0 + c);
                const float4 wv = *(const float4*)(P.w + (size_t)(ky * P.k + kx) * P.cout + c);
                acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y); acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
            }
        }
        float a[4] = {acc.x, acc.y, acc.z, acc.w};
        if (P.bias) { const float4 bb = *(const float4*)(P.bias + c); a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w; }
        if (P.act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = a[r] >= 0.f ? a[r] : a[r] * P.prelu;
        }
        if (P.oscale) { const float* os = P.oscale + (size_t)t * P.oscale_stride + c; a[0] *= os[0]; a[1] *= os[1]; a[2] *= os[2]; a[3] *= os[3]; }
        if (P.res) { const float4 rr = *(const float4*)(P.res + pix * P.cs_res + c); a[0] += rr.x; a[1] += rr.y; a[2] += rr.z; a[3] += rr.w; }
        *(float4*)((float*)P.out + pix * P.cs_out + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// RepConv2 + SimpleGate in one pass (gshift_deblur1.py:143-157,175-178): g1[c] = a'[c] * a'[C + c] with a' = dw3x3(a) + a over the 2C channels of
// the first 1x1's output.  One thread = the channel quads c..c+3 and C+c..C+c+3 of a pixel (their 18 weight rows live in registers), a workgroup
// = 256 / (C/4) pixels abreast walking a contiguous chunk of one frame.  a' is never written (2C floats per pixel out and back in), and the
// products are summed in the order dw32_kernel + gate32_kernel use, so g1 is bit-identical to the unfused pair.  partial != NULL: the
// per-(frame, workgroup) channel sums of g1 for the CALayer2 behind the gate (denoisers), as sn32_gate_sum.
__global__ __launch_bounds__(256) void dwgate32_kernel(const float* a, int cs, const float* w, int C, int cpad, int h, int wd, int chunk, float* out, float* partial) {
    __shared__ float red[256 * 4];
    const int t = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int c4n = C >> 2, npp = 256 / c4n, q = tid % c4n, part = tid / c4n, c = 4 * q, hw = h * wd;
    float4 w1[9], w2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { w1[i] = *(const float4*)(w + (size_t)i * 2 * C + c); w2[i] = *(const float4*)(w + (size_t)i * 2 * C + C + c); }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* at = a + (size_t)t * hw * cs;
    float* ot = out + (size_t)t * hw * C;
    const int pend = min(hw, (blk + 1) * chunk);
    if (part < npp) {
        for (int pix = blk * chunk + part; pix < pend; pix += npp) {
            const int oy = pix / wd, ox = pix - oy * wd;
            float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1, r1 = a1, r2 = a1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int gy = oy - 1 + ky;
                if (gy < 0 || gy >= h) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int gx = ox - 1 + kx;
                    if (gx < 0 || gx >= wd) continue;
                    const float* px = at + ((size_t)gy * wd + gx) * cs + c;
                    const float4 x1 = *(const float4*)px, x2 = *(const float4*)(px + C);
                    const float4 u = w1[ky * 3 + kx], v = w2[ky * 3 + kx];
                    a1.x = fmaf(x1.x, u.x, a1.x); a1.y = fmaf(x1.y, u.y, a1.y); a1.z = fmaf(x1.z, u.z, a1.z); a1.w = fmaf(x1.w, u.w, a1.w);
                    a2.x = fmaf(x2.x, v.x, a2.x); a2.y = fmaf(x2.y, v.y, a2.y); a2.z = fmaf(x2.z, v.z, a2.z); a2.w = fmaf(x2.w, v.w, a2.w);
                    if (ky == 1 && kx == 1) { r1 = x1; r2 = x2; }
                }
            }
            const float4 g = make_float4((a1.x + r1.x) * (a2.x + r2.x), (a1.y + r1.y) * (a2.y + r2.y), (a1.z + r1.z) * (a2.z + r2.z), (a1.w + r1.w) * (a2.w + r2.w));
            *(float4*)(ot + (size_t)pix * C + c) = g;
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        }
    }
    if (!partial) return;
    *(float4*)(red + tid * 4) = s;                                  // [part][C] (tid * 4 = part * C + c for part < npp)
    __syncthreads();
    if (tid < cpad) {
        float m = 0.f;
        if (tid < C)
            for (int pq = 0; pq < npp; ++pq) m += red[pq * C + tid];
        partial[((size_t)t * gridDim.x + blk) * cpad + tid] = m;
    }
}

struct Unit32 { const float* x; const float* halo; int T, h, w, C, mode, wrap, t0; };

// u = cat(roll(x), spatial_shift2(borrowed half)) (gshift_deblur1.py:504-528); CU = 3C/2, or C for the roll alone (Shift_CAB)
// u2 (optional): a second [T][h][w][CU] tensor that receives the first C channels (the rolled tensor = CAB2's shortcut) as well: the LayerNorm
// input cat(shortcut, conv1(shifted)) is assembled in it without a second gather pass.
__global__ void gather32_kernel(const Unit32 U, const int8_t* offs, float* u, const int CU, float* u2) {
    const int t = U.t0 + blockIdx.y, Ch = U.C >> 1, hw = U.h * U.w;
    const SnSlabs<float> s = sn_unit_slabs<float>(U.x, U.halo, U.T, hw, U.C, U.mode, U.wrap, t);      // SURVEY.md 8a-1 table (sn_common.h)
    const size_t n = (size_t)hw * CU;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / CU), c = (int)(e - (size_t)i * CU);
        float v = 0.f;
        if (c < Ch) v = s.p0[(size_t)i * s.s0 + c];
        else if (c < U.C) v = s.p1[(size_t)i * s.s1 + c - Ch];
        else {
            const int k = c - U.C, y = i / U.w, x = i - y * U.w;
            const int sy = y + offs[2 * k], sx = x + offs[2 * k + 1];
            if (sy >= 0 && sy < U.h && sx >= 0 && sx < U.w) v = s.pb[((size_t)sy * U.w + sx) * s.sb + k];
        }
        u[(size_t)t * n + e] = v;
        if (u2 && c < U.C) u2[(size_t)t * n + e] = v;
    }
}

// channel_shift for CAB2 without the second copy of roll(x) that sn32_gsts_gather's cat(roll(x), shift(borrowed)) holds: roll(x) goes
// straight into the LayerNorm input vin[:, :C] (which also is the shortcut), and either (CONV = false, the default of the engine) the
// shifted half alone goes to u [T][hw][C/2] for the depthwise conv1, or (CONV = true) conv1 is computed here as well and vin is complete.
// The CONV form is bit-identical but SLOWER on MI355X (0.60 ms against 0.30 + 0.11 ms at 36 x 272 x 448 x 80): nine 4-byte loads per
// output whose addresses scatter over 24 shift offsets per wave, where the separate conv reads 16 bytes per lane, coalesced.  Phase A copies roll(x) in 16-byte pieces; phase B computes one (pixel, borrowed channel)
// per thread: nine taps of the SHIFTED image, each tap zero when it falls outside the frame (the conv's padding) or when its source does
// (the shift's zero fill), summed in the tap order of dw32_kernel (bit-identical to gather + conv).
template <bool CONV>
__global__ __launch_bounds__(256) void shiftconv32_kernel(const Unit32 U, const int8_t* offs, const float* w, float* vin, float* u) {
    const int t = U.t0 + blockIdx.y, Ch = U.C >> 1, hw = U.h * U.w, CU = U.C + Ch, c4n = U.C >> 2;
    const SnSlabs<float> s = sn_unit_slabs<float>(U.x, U.halo, U.T, hw, U.C, U.mode, U.wrap, t);      // SURVEY.md 8a-1 table (sn_common.h)
    float* const vt = vin + (size_t)t * hw * CU;
    const size_t na = (size_t)hw * c4n;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < na; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / c4n), c = (int)(e - (size_t)i * c4n) * 4;
        const float4 v = c < Ch ? *(const float4*)(s.p0 + (size_t)i * s.s0 + c) : *(const float4*)(s.p1 + (size_t)i * s.s1 + c - Ch);
        *(float4*)(vt + (size_t)i * CU + c) = v;
    }
    const size_t nb = (size_t)hw * Ch;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nb; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / Ch), k = (int)(e - (size_t)i * Ch);
        const int y = i / U.w, x = i - y * U.w, oy = offs[2 * k], ox = offs[2 * k + 1];
        if constexpr (!CONV) {                                     // the shifted half alone, for a separate conv1: u [T][hw][C/2]
            const int sy = y + oy, sx = x + ox;
            float xv = 0.f;
            if (sy >= 0 && sy < U.h && sx >= 0 && sx < U.w) xv = s.pb[((size_t)sy * U.w + sx) * s.sb + k];
            u[(size_t)t * nb + e] = xv;
            continue;
        }
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int gy = y - 1 + ky;
            if (gy < 0 || gy >= U.h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int gx = x - 1 + kx;
                if (gx < 0 || gx >= U.w) continue;
                const int sy = gy + oy, sx = gx + ox;
                float xv = 0.f;
                if (sy >= 0 && sy < U.h && sx >= 0 && sx < U.w) xv = s.pb[((size_t)sy * U.w + sx) * s.sb + k];
                acc = fmaf(xv, w[(ky * 3 + kx) * Ch + k], acc);
            }
        }
        vt[(size_t)i * CU + U.C + k] = acc;
    }
}

// LayerNorm2d (gshift_deblur1.py:19-28): per pixel over K channels, biased variance, eps inside the sqrt.  16 lanes (one DPP row) per
// pixel: lane i takes channels i, i + 16, ... (64-byte coalesced segments), the two reductions are DPP row sums.  K <= 128.
__global__ __launch_bounds__(256) void layernorm32_kernel(const float* x, int cs_x, int K, const float* w, const float* b, float* out, int cs_out, size_t npix) {
    const int i = threadIdx.x & 15;
    for (size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); p < npix + 15; p += (size_t)gridDim.x * 16) {      // (whole rows stay converged for the DPP sums)
        const bool live = p < npix;
        const float* xp = x + (live ? p : 0) * cs_x;
        float v[8];
        float mu = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = i + 16 * j; v[j] = c < K ? xp[c] : 0.f; mu += v[j]; }
        mu = row_sum16(mu) / (float)K;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (i + 16 * j) < K ? v[j] - mu : 0.f; var += d * d; }
        var = row_sum16(var) / (float)K;
        const float rstd = 1.0f / sqrtf(var + 1e-6f);
        if (!live) continue;
        float* op = out + p * cs_out;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = i + 16 * j; if (c < K) op[c] = (v[j] - mu) * rstd * w[c] + b[c]; }
    }
}

// SimpleGate / SimpleGate2 (gshift_deblur1.py:175-182): first half x second half (or sigmoid of it)
__global__ void gate32_kernel(const float* a, int C, int mode, float* out, size_t npix) {
    const size_t n = npix * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / C; const int c = (int)(e - p * C);
        const float x1 = a[p * 2 * C + c], x2 = a[p * 2 * C + C + c];
        out[e] = mode ? x1 / (1.0f + expf(-x2)) : x1 * x2;
    }
}

// per-(frame, block) channel sums for AdaptiveAvgPool2d(1): partial [T][nblk][cpad], deterministic (no atomics)
__global__ __launch_bounds__(256) void chan_sum32_kernel(const float* x, int cs, int C, int cpad, int hw, float* partial) {
    __shared__ float acc[256];
    const int t = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
    const int nsplit = 256 / cpad, ch = tid % cpad, part = tid / cpad;
    float s = 0.f;
    if (part < nsplit && ch < C) {
        const float* xt = x + (size_t)t * hw * cs + ch;
        for (int i = blk * nsplit + part; i < hw; i += nblk * nsplit) s += xt[(size_t)i * cs];
    }
    acc[tid] = s;
    __syncthreads();
    if (tid < cpad) {
        float m = 0.f;
        for (int q = 0; q < nsplit; ++q) m += acc[q * cpad + tid];
        partial[((size_t)t * nblk + blk) * cpad + tid] = m;
    }
}

// SimpleGate / SimpleGate2 AND the per-(frame, block) channel sums of its result in one pass (the CALayer2 that follows every gate of the
// denoisers, and SimpleGate2 of all variants): same pixel walk and summation order as chan_sum32_kernel, so the sums are the ones
// sn32_gate + sn32_chan_sum produce, without reading the gated tensor back.
__global__ __launch_bounds__(256) void gate_sum32_kernel(const float* a, int C, int cpad, int mode, int hw, float* out, float* partial) {
    __shared__ float acc[256];
    const int t = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
    const int nsplit = 256 / cpad, ch = tid % cpad, part = tid / cpad;
    float s = 0.f;
    if (part < nsplit && ch < C) {
        const float* at = a + (size_t)t * hw * 2 * C + ch;
        float* ot = out + (size_t)t * hw * C + ch;
        for (int i = blk * nsplit + part; i < hw; i += nblk * nsplit) {
            const float x1 = at[(size_t)i * 2 * C], x2 = at[(size_t)i * 2 * C + C];
            const float g = mode ? x1 / (1.0f + expf(-x2)) : x1 * x2;
            ot[(size_t)i * C] = g;
            s += g;
        }
    }
    acc[tid] = s;
    __syncthreads();
    if (tid < cpad) {
        float m = 0.f;
        for (int q = 0; q < nsplit; ++q) m += acc[q * cpad + tid];
        partial[((size_t)t * nblk + blk) * cpad + tid] = m;
    }
}

// out = r * ca[t][c] (+ x): CALayer scale with or without the CAB residual (gshift_deblur1.py:69-70,155-157)
__global__ void scale_res32_kernel(const float* r, const float* x, const float* ca, int ca_stride, float* out, int C, int hw, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int t = (int)(e / ((size_t)hw * C));
        float v = r[e] * ca[(size_t)t * ca_stride + c];
        if (x) v += x[e];
        out[e] = v;
    }
}

__global__ void ingest32_kernel(const void* src, int dt, const void* noise, float* dst, int C, int CD, int HW) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    auto ld = [&](const void* p, size_t idx) {
        return dt == SN_F32 ? ((const float*)p)[idx] : (dt == SN_F16 ? __half2float(((const __half*)p)[idx]) : bf_to_f(((const bf16_t*)p)[idx]));
    };
    for (int c = 0; c < C; ++c) dst[((size_t)t * HW + i) * CD + c] = ld(src, ((size_t)t * C + c) * HW + i);
    if (noise) dst[((size_t)t * HW + i) * CD + C] = ld(noise, (size_t)t * HW + i);
}

int grid_for(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 65535 * 16 ? 65535 * 16 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" {

int sn32_conv2d(const sn32_conv_desc* d, void* stream) {
    sn_clear_error();
    if (!d || d->n_in < 1 || d->n_in > 3 || !d->w || !d->out || d->k < 1 || d->k > 5 || (d->stride != 1 && d->stride != 2) ||
        d->groups < 1 || d->c_out < 1 || (d->c_out % d->groups) || d->T < 1) return SN_EINVAL;
    if (d->groups > 1 && d->n_in != 1) return SN_EINVAL;
    if (d->in_mode == 1 && (((d->h_in | d->w_in) & 1) || d->groups != 1)) return SN_EINVAL;
    if (d->out_mode == 2 && !d->sc) return SN_EINVAL;
    Conv32K K;
    K.in0 = d->in[0]; K.in1 = d->in[1]; K.in2 = d->in[2];
    K.cin0 = d->c_in[0]; K.cin1 = d->c_in[1]; K.cin2 = d->c_in[2];
    K.cs0 = d->cs_in[0]; K.cs1 = d->cs_in[1]; K.cs2 = d->cs_in[2];
    K.n_in = d->n_in; K.T = d->T; K.hin = d->h_in; K.win = d->w_in; K.in_mode = d->in_mode;
    K.k = d->k; K.stride = d->stride; K.pad = d->pad; K.groups = d->groups; K.hout = d->h_out; K.wout = d->w_out; K.cout = d->c_out;
    K.cin_total = 0;
    for (int i = 0; i < d->n_in; ++i) { if (!d->in[i] || d->c_in[i] < 1 || d->cs_in[i] < d->c_in[i]) return SN_EINVAL; K.cin_total += d->c_in[i]; }
    if (K.cin_total % d->groups) return SN_EINVAL;
    K.w = d->w; K.bias = d->bias; K.act = d->act; K.prelu = d->prelu; K.oscale = d->oscale; K.oscale_stride = d->oscale_stride;
    K.res = d->res; K.cs_res = d->cs_res; K.out = d->out; K.cs_out = d->cs_out; K.out_mode = d->out_mode; K.nchw_dtype = d->nchw_dtype; K.sc = d->sc;
    const size_t n = (size_t)d->T * d->h_out * d->w_out * d->c_out;
    const int cin_g = K.cin_total / d->groups, cout_g = d->c_out / d->groups;
    K.wsplit = (const uint4*)d->wsplit;
    K.iscale = d->iscale; K.iscale_stride = d->iscale_stride;
    K.rscale = d->rscale; K.rscale_stride = d->rscale_stride;
    K.lnw = d->ln_w; K.lnb = d->ln_b; K.csum = d->csum; K.csum_cpad = d->csum_cpad;
    {
        const bool split1 = d->wsplit && d->n_in == 1 && d->in_mode == 0 && d->stride == 1 && (d->cs_in[0] & 3) == 0 && (K.cin_total & 3) == 0 &&
                            ((size_t)d->in[0] & 15) == 0 && ((size_t)d->wsplit & 15) == 0 && d->groups == 1 && d->out_mode == 0;
        const int mt0 = (d->c_out + 15) / 16;
        // LayerNorm on load: the flat-pixel split 1x1 kernel only (same conditions as its dispatch below)
        if (d->ln_w && !(split1 && d->ln_b && !d->iscale && d->k == 1 && d->pad == 0 && (K.cin_total + 31) / 32 <= 4 && d->h_out * d->w_out >= SN_1X1_NPX &&
                         (d->c_out & 3) == 0 && (d->cs_out & 3) == 0 && ((size_t)d->out & 15) == 0 && (!d->res || ((d->cs_res & 3) == 0 && ((size_t)d->res & 15) == 0)) &&
                         mt0 > 1 && (((size_t)d->ln_w | (size_t)d->ln_b) & 15) == 0)) return SN_EINVAL;
        // channel sums of the output: the split dense 3x3 kernel only
        if (d->csum && !(split1 && d->k == 3 && d->pad == 1 && d->csum_cpad >= d->c_out && d->csum_cpad <= 16 * mt0)) return SN_EINVAL;
    }
    // the residual scale exists in the SPLIT-PRECISION grouped-by-8 kernels only (the "+" RepConv of the denoisers: res = g1 * ca1, never materialised;
    // in the exact kernels its operand cost conv32m_kernel<5, 8, 32> a wave of occupancy: 115 -> 188 VGPRs): same conditions as the dispatch below
    if (d->rscale && !(d->res && d->wsplit && ((size_t)d->wsplit & 15) == 0 && d->groups > 1 && cin_g == 8 && cout_g == 8 && d->c_out % 16 == 0 && d->n_in == 1 &&
                       d->in_mode == 0 && d->stride == 1 && (d->c_in[0] & 3) == 0 && (d->cs_in[0] & 3) == 0 && ((size_t)d->in[0] & 15) == 0 &&
                       ((d->k == 5 && d->pad == 2) || (d->k == 3 && d->pad == 1)) && (d->rscale_stride & 3) == 0 && ((size_t)d->rscale & 15) == 0)) return SN_EINVAL;
    // the input scale is implemented by the matrix-core kernels' 16-byte staging path only
    if (d->iscale && !((d->groups == 1 || (cin_g == 8 && cout_g == 8 && d->c_out % 16 == 0)) && d->n_in == 1 && d->in_mode == 0 && (d->cs_in[0] & 3) == 0 &&
                       (K.cin_total & 3) == 0 && ((size_t)d->in[0] & 15) == 0 && (d->iscale_stride & 3) == 0 && ((size_t)d->iscale & 15) == 0)) return SN_EINVAL;
    if (d->groups == 1 || (cin_g == 8 && cout_g == 8 && d->c_out % 16 == 0)) {          // matrix cores: dense convs and the "+" RepConv
        hipStream_t st = (hipStream_t)stream;
        const int mt = (d->c_out + 15) / 16;
        // split-precision path (bf16 hi + lo operands, three bf16 MFMAs per k-step): single float4-addressable input, stride 1, NHWC out
        bool quads = true;                    // every input addressable in aligned 16-byte quads that stay inside one input
        for (int i = 0; i < d->n_in; ++i) quads = quads && (d->c_in[i] & 3) == 0 && (d->cs_in[i] & 3) == 0 && ((size_t)d->in[i] & 15) == 0;
        // SkipUpSample: bilinear x2 on load + 1x1 (+ residual) on the flat-pixel split kernel; anything it does not cover keeps the exact kernel below
        if (d->wsplit && d->in_mode == 1 && d->n_in == 1 && quads && d->stride == 1 && ((size_t)d->wsplit & 15) == 0 && d->groups == 1 && d->k == 1 && d->pad == 0 &&
            d->out_mode == 0 && !d->iscale && !d->ln_w && !d->csum && (K.cin_total + 31) / 32 <= 4 && d->h_out * d->w_out >= SN_1X1_NPX && (d->c_out & 3) == 0 &&
            (d->cs_out & 3) == 0 && ((size_t)d->out & 15) == 0 && (!d->res || ((d->cs_res & 3) == 0 && ((size_t)d->res & 15) == 0)) && mt > 1) {
            const int ncb1 = (K.cin_total + 31) / 32;
            const long long npix = (long long)d->T * d->h_out * d->w_out;
            const size_t lds = (size_t)SN_1X1_NPX * (ncb1 * 160 + ((ncb1 & 1) ? 0 : 32));
            const dim3 grid((unsigned)((npix + SN_1X1_NPX - 1) / SN_1X1_NPX));
#define SN_1X1_CASE(N) case N: \
            if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv32s_1x1_kernel<N, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH; \
            hipLaunchKernelGGL((conv32s_1x1_kernel<N, false, 2>), grid, dim3(256), lds, st, K, npix, (float*)nullptr, 0); break;
            switch (ncb1) { SN_1X1_CASE(1) SN_1X1_CASE(2) SN_1X1_CASE(3) SN_1X1_CASE(4) }
#undef SN_1X1_CASE
            return sn_check_launch();
        }
        if (d->wsplit && (d->n_in == 1 || (d->groups == 1 && !d->iscale)) && quads && d->in_mode == 0 && d->stride == 1 && ((size_t)d->wsplit & 15) == 0) {
            const int ncb1 = (K.cin_total + 31) / 32;
            if (d->n_in == 1 && d->groups == 1 && d->k == 1 && d->pad == 0 && d->out_mode == 0 && ncb1 <= 4 && d->h_out * d->w_out >= SN_1X1_NPX && (d->c_out & 3) == 0 && (d->cs_out & 3) == 0 &&
                ((size_t)d->out & 15) == 0 && (!d->res || ((d->cs_res & 3) == 0 && ((size_t)d->res & 15) == 0)) && mt > 1) {
                const long long npix = (long long)d->T * d->h_out * d->w_out;
                const size_t lds = (size_t)SN_1X1_NPX * (ncb1 * 160 + ((ncb1 & 1) ? 0 : 32));
                const dim3 grid((unsigned)((npix + SN_1X1_NPX - 1) / SN_1X1_NPX));
#define SN_1X1_CASE(N) case N: \
                    if (d->ln_w) { \
                        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv32s_1x1_kernel<N, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH; \
                        hipLaunchKernelGGL((conv32s_1x1_kernel<N, false, 1>), grid, dim3(256), lds, st, K, npix, (float*)nullptr, 0); break; \
                    } \
                    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv32s_1x1_kernel<N, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH; \
                    hipLaunchKernelGGL((conv32s_1x1_kernel<N, false, 0>), grid, dim3(256), lds, st, K, npix, (float*)nullptr, 0); break;
                switch (ncb1) { SN_1X1_CASE(1) SN_1X1_CASE(2) SN_1X1_CASE(3) SN_1X1_CASE(4) }
#undef SN_1X1_CASE
                return sn_check_launch();
            }
            if (d->groups == 1 && d->k == 1 && d->pad == 0)
                return mt == 1 ? launch_conv32s<1, 1, false>(K, st) : (mt <= 3 ? launch_conv32s<3, 1, false>(K, st) : launch_conv32s<4, 1, false>(K, st));
            if (d->groups == 1 && d->k == 3 && d->pad == 1)
                return mt == 1 ? launch_conv32s<1, 3, false>(K, st) : (mt <= 3 ? launch_conv32s<3, 3, false>(K, st) : launch_conv32s<4, 3, false>(K, st));
            if (d->groups > 1 && d->k == 5 && d->pad == 2)
                return mt <= 3 ? launch_conv32s<3, 5, true>(K, st) : launch_conv32s<4, 5, true>(K, st);
            if (d->groups > 1 && d->k == 3 && d->pad == 1)
                return mt <= 3 ? launch_conv32s<3, 3, true>(K, st) : launch_conv32s<4, 3, true>(K, st);
        }
        if (d->stride == 2) return mt <= 2 ? launch_conv32m<2, 4, 16>(K, st) : launch_conv32m<5, 4, 16>(K, st);
        return mt == 1 ? launch_conv32m<1, 8, 32>(K, st) : (mt <= 3 ? launch_conv32m<3, 8, 32>(K, st) : launch_conv32m<5, 8, 32>(K, st));
    }
    if (d->groups == d->c_out && cin_g == 1 && d->stride == 1 && d->in_mode == 0 && d->out_mode == 0 && (d->c_out & 3) == 0 && (d->cs_in[0] & 3) == 0 &&
        (d->cs_out & 3) == 0 && (!d->res || (d->cs_res & 3) == 0) && (((size_t)d->in[0] | (size_t)d->out | (size_t)d->res | (size_t)d->w) & 15) == 0) {
        hipLaunchKernelGGL(dw32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, K);      // depthwise, 4 channels per thread
        return sn_check_launch();
    }
    if ((d->c_out / d->groups) % 4 == 0)      // a thread's four channels share a group (and the weight row is 16-byte aligned: c_out % 4 == 0)
        hipLaunchKernelGGL(conv32_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, K);
    else
        hipLaunchKernelGGL(conv32_kernel<1>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, K);
    return sn_check_launch();
}

int sn32_gsts_gather(const sn_unit_src* s, const int8_t* offs, float* u, float* u2, void* stream) {
    sn_clear_error();
    if (!s || !s->x || !u || (u2 && !offs) || (s->C & 1) || s->T < 1 || s->mode < 1 || s->mode > 2 || s->wrap < 0 || s->wrap > 2 || (s->wrap == 2 && !s->halo)) return SN_EINVAL;
    Unit32 U; U.x = (const float*)s->x; U.halo = (const float*)s->halo; U.T = s->T; U.h = s->h; U.w = s->w; U.C = s->C; U.mode = s->mode; U.wrap = s->wrap;
    SN_FRAME_RANGE(s, t0, nt);
    U.t0 = t0;
    hipLaunchKernelGGL(gather32_kernel, dim3(1024, nt), dim3(256), 0, (hipStream_t)stream, U, offs, u, offs ? s->C + s->C / 2 : s->C, u2);
    return sn_check_launch();
}

int sn32_gsts_shiftconv(const sn_unit_src* s, const int8_t* offs, const float* w, float* vin, float* u, void* stream) {
    sn_clear_error();
    if (!s || !s->x || !offs || (!w == !u) || !vin || (s->C & 7) || s->T < 1 || s->mode < 1 || s->mode > 2 || s->wrap < 0 || s->wrap > 2 || (s->wrap == 2 && !s->halo) ||
        (((size_t)s->x | (size_t)s->halo | (size_t)vin) & 15)) return SN_EINVAL;
    Unit32 U; U.x = (const float*)s->x; U.halo = (const float*)s->halo; U.T = s->T; U.h = s->h; U.w = s->w; U.C = s->C; U.mode = s->mode; U.wrap = s->wrap;
    SN_FRAME_RANGE(s, t0, nt);
    U.t0 = t0;
    if (w) hipLaunchKernelGGL(shiftconv32_kernel<true>, dim3(1024, nt), dim3(256), 0, (hipStream_t)stream, U, offs, w, vin, u);
    else hipLaunchKernelGGL(shiftconv32_kernel<false>, dim3(1024, nt), dim3(256), 0, (hipStream_t)stream, U, offs, w, vin, u);
    return sn_check_launch();
}

int sn32_layernorm(const float* x, int cs_x, int K, const float* w, const float* b, float* out, int cs_out, long long npix, void* stream) {
    sn_clear_error();
    if (!x || !w || !b || !out || K < 1 || K > 128 || cs_x < K || cs_out < K || npix < 1) return SN_EINVAL;
    hipLaunchKernelGGL(layernorm32_kernel, dim3(grid_for((size_t)npix * 16)), dim3(256), 0, (hipStream_t)stream, x, cs_x, K, w, b, out, cs_out, (size_t)npix);
    return sn_check_launch();
}

int sn32_gate(const float* a, int C, int mode, float* out, long long npix, void* stream) {
    sn_clear_error();
    if (!a || !out || C < 1 || npix < 1 || mode < 0 || mode > 1) return SN_EINVAL;
    hipLaunchKernelGGL(gate32_kernel, dim3(grid_for((size_t)npix * C)), dim3(256), 0, (hipStream_t)stream, a, C, mode, out, (size_t)npix);
    return sn_check_launch();
}

int sn32_gate_sum(const float* a, int C, int cpad, int mode, float* out, int T, int hw, int nblk, float* partial, void* stream) {
    sn_clear_error();
    if (!a || !out || !partial || C < 1 || cpad < C || cpad > 256 || nblk < 1 || T < 1 || hw < 1 || mode < 0 || mode > 1) return SN_EINVAL;
    hipLaunchKernelGGL(gate_sum32_kernel, dim3(nblk, T), dim3(256), 0, (hipStream_t)stream, a, C, cpad, mode, hw, out, partial);
    return sn_check_launch();
}

int sn32_conv_csum_tiles(int h_out, int w_out) {       // workgroups per frame of the kernel that fills sn32_conv_desc.csum
    return ((h_out + SN_C32S_TH3 - 1) / SN_C32S_TH3) * ((w_out + 31) / 32);
}

int sn32_conv1x1_gate2(const float* x, int cs_x, int cin, const void* wsplit, int C, int cpad, float* out, int T, int hw, float* partial, void* stream) {
    sn_clear_error();
    const int ncb1 = (cin + 31) / 32;
    if (!x || !wsplit || !out || !partial || cin < 4 || (cin & 3) || cs_x < cin || (cs_x & 3) || ncb1 > 4 || C < 16 || (C & 15) || cpad < C || (cpad & 3) || T < 1 ||
        hw < SN_1X1_NPX || (hw % SN_1X1_NPX) || (((size_t)x | (size_t)wsplit | (size_t)out | (size_t)partial) & 15)) return SN_EINVAL;
    Conv32K K{};
    K.in0 = x; K.cin0 = cin; K.cs0 = cs_x; K.n_in = 1; K.T = T; K.hin = 1; K.win = hw; K.k = 1; K.stride = 1; K.groups = 1; K.hout = 1; K.wout = hw;
    K.cout = 2 * C; K.cin_total = cin; K.out = out; K.cs_out = C; K.wsplit = (const uint4*)wsplit;
    const long long npix = (long long)T * hw;
    const size_t lds = (size_t)SN_1X1_NPX * (ncb1 * 160 + ((ncb1 & 1) ? 0 : 32));
    const dim3 grid((unsigned)(npix / SN_1X1_NPX));
    hipStream_t st = (hipStream_t)stream;
#define SN_1X1_CASE(N) case N: \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv32s_1x1_kernel<N, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH; \
    hipLaunchKernelGGL((conv32s_1x1_kernel<N, true, 0>), grid, dim3(256), lds, st, K, npix, partial, cpad); break;
    switch (ncb1) { SN_1X1_CASE(1) SN_1X1_CASE(2) SN_1X1_CASE(3) SN_1X1_CASE(4) }
#undef SN_1X1_CASE
    return sn_check_launch();
}

int sn32_dw_gate(const float* a, int cs_a, const float* w, int C, int cpad, float* out, int T, int h, int wd, int nblk, float* partial, void* stream) {
    sn_clear_error();
    if (!a || !w || !out || C < 4 || (C & 3) || C > 1024 || cs_a < 2 * C || (cs_a & 3) || T < 1 || h < 1 || wd < 1 || nblk < 1 ||
        (((size_t)a | (size_t)w | (size_t)out) & 15) || (partial && (cpad < C || cpad > 256))) return SN_EINVAL;
    const int hw = h * wd, chunk = (hw + nblk - 1) / nblk;
    hipLaunchKernelGGL(dwgate32_kernel, dim3(nblk, T), dim3(256), 0, (hipStream_t)stream, a, cs_a, w, C, cpad, h, wd, chunk, out, partial);
    return sn_check_launch();
}

int sn32_chan_sum(const float* x, int cs, int C, int cpad, int T, int hw, int nblk, float* partial, void* stream) {
    sn_clear_error();
    if (!x || !partial || C < 1 || cpad < C || cpad > 256 || nblk < 1 || T < 1 || hw < 1) return SN_EINVAL;
    hipLaunchKernelGGL(chan_sum32_kernel, dim3(nblk, T), dim3(256), 0, (hipStream_t)stream, x, cs, C, cpad, hw, partial);
    return sn_check_launch();
}

int sn32_scale_residual(const float* r, const float* x, const float* ca, int ca_stride, float* out, int T, int hw, int C, void* stream) {
    sn_clear_error();
    if (!r || !ca || !out || C < 1 || T < 1 || hw < 1) return SN_EINVAL;
    const size_t n = (size_t)T * hw * C;
    hipLaunchKernelGGL(scale_res32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, r, x, ca, ca_stride, out, C, hw, n);
    return sn_check_launch();
}

int sn32_ingest(const void* src, int dt, const void* noise, float* dst, int T, int C, int H, int W, void* stream) {
    sn_clear_error();
    if (!src || !dst || C < 1 || dt < 0 || dt > 2) return SN_EINVAL;
    const int hw = H * W;
    hipLaunchKernelGGL(ingest32_kernel, dim3((hw + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, src, dt, noise, dst, C, C + (noise ? 1 : 0), hw);
    return sn_check_launch();
}

}  // extern "C"
