// fp32-storage kernels of the Shift-Net path (gfx950): the arithmetic type the reference's denoise CLI runs the
// "+" denoiser in (inference/test_denoise.py:83-85 keeps the module in float32) and the validation build of the
// engine's control flow: every activation is fp32 NHWC [T][H][W][C] (no channel padding), every weight is the
// checkpoint's fp32 value (only re-ordered so that consecutive lanes read consecutive output channels), every
// accumulation is a sequential fp32 FMA chain.  Results match the CPU oracle to fp32 round-off, so a wrong tap, slab,
// gate half or weight row shows up as an error >> 1e-4 instead of hiding inside bf16 noise.
//
// These are direct (not MFMA) kernels: one thread per output element, operands served by L1/L2.  They are a
// correctness-first path (fp32 FMA rate, ~1/16 of the bf16 MFMA rate on this chip); the bf16 kernels in
// sn_conv.hip / sn_gsts*.hip are the throughput path.
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
            if (P.res) a += P.res[pix * P.cs_res + c];
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

struct Unit32 { const float* x; int T, h, w, C, mode, wrap; };

// u = cat(roll(x), spatial_shift2(borrowed half)) (gshift_deblur1.py:504-528); CU = 3C/2, or C for the roll alone (Shift_CAB)
__global__ void gather32_kernel(const Unit32 U, const int8_t* offs, float* u, const int CU) {
    const int t = blockIdx.y, Ch = U.C >> 1, hw = U.h * U.w;
    int f0 = t, o0 = 0, f1 = t, o1 = Ch, fb = t, ob = 0;      // SURVEY.md 8a-1 table (same as sn_gsts.hip::unit_slabs)
    if (U.mode == 1) {
        if (t > 0 || U.wrap) { f0 = sn_prev_frame(t, U.T, U.wrap); o0 = Ch; f1 = t; o1 = 0; fb = f0; ob = Ch; }
    } else if (U.mode == 2) {
        if (t < U.T - 1 || U.wrap) { f0 = t; o0 = Ch; f1 = sn_next_frame(t, U.T, U.wrap); o1 = 0; fb = f1; ob = 0; }
        else { fb = t; ob = Ch; }
    }
    const size_t n = (size_t)hw * CU;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / CU), c = (int)(e - (size_t)i * CU);
        float v = 0.f;
        if (c < Ch) v = U.x[((ptrdiff_t)f0 * hw + i) * U.C + o0 + c];
        else if (c < U.C) v = U.x[((ptrdiff_t)f1 * hw + i) * U.C + o1 + c - Ch];
        else {
            const int k = c - U.C, y = i / U.w, x = i - y * U.w;
            const int sy = y + offs[2 * k], sx = x + offs[2 * k + 1];
            if (sy >= 0 && sy < U.h && sx >= 0 && sx < U.w) v = U.x[(((ptrdiff_t)fb * U.h + sy) * U.w + sx) * U.C + ob + k];
        }
        u[(size_t)t * n + e] = v;
    }
}

// LayerNorm2d (gshift_deblur1.py:19-28): per pixel over K channels, biased variance, eps inside the sqrt
__global__ void layernorm32_kernel(const float* x, int cs_x, int K, const float* w, const float* b, float* out, int cs_out, size_t npix) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const float* xp = x + p * cs_x;
        float mu = 0.f;
        for (int c = 0; c < K; ++c) mu += xp[c];
        mu /= (float)K;
        float var = 0.f;
        for (int c = 0; c < K; ++c) { const float d = xp[c] - mu; var += d * d; }
        var /= (float)K;
        const float rstd = 1.0f / sqrtf(var + 1e-6f);
        float* op = out + p * cs_out;
        for (int c = 0; c < K; ++c) op[c] = (xp[c] - mu) * rstd * w[c] + b[c];
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
    if ((d->c_out / d->groups) % 4 == 0)      // a thread's four channels share a group (and the weight row is 16-byte aligned: c_out % 4 == 0)
        hipLaunchKernelGGL(conv32_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, K);
    else
        hipLaunchKernelGGL(conv32_kernel<1>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, K);
    return sn_check_launch();
}

int sn32_gsts_gather(const sn_unit_src* s, const int8_t* offs, float* u, void* stream) {
    sn_clear_error();
    if (!s || !s->x || !u || (s->C & 1) || s->T < 1 || s->mode < 1 || s->mode > 2) return SN_EINVAL;
    Unit32 U; U.x = (const float*)s->x; U.T = s->T; U.h = s->h; U.w = s->w; U.C = s->C; U.mode = s->mode; U.wrap = s->wrap;
    hipLaunchKernelGGL(gather32_kernel, dim3(1024, s->T), dim3(256), 0, (hipStream_t)stream, U, offs, u, offs ? s->C + s->C / 2 : s->C);
    return sn_check_launch();
}

int sn32_layernorm(const float* x, int cs_x, int K, const float* w, const float* b, float* out, int cs_out, long long npix, void* stream) {
    sn_clear_error();
    if (!x || !w || !b || !out || K < 1 || cs_x < K || cs_out < K || npix < 1) return SN_EINVAL;
    hipLaunchKernelGGL(layernorm32_kernel, dim3(grid_for((size_t)npix)), dim3(256), 0, (hipStream_t)stream, x, cs_x, K, w, b, out, cs_out, (size_t)npix);
    return sn_check_launch();
}

int sn32_gate(const float* a, int C, int mode, float* out, long long npix, void* stream) {
    sn_clear_error();
    if (!a || !out || C < 1 || npix < 1 || mode < 0 || mode > 1) return SN_EINVAL;
    hipLaunchKernelGGL(gate32_kernel, dim3(grid_for((size_t)npix * C)), dim3(256), 0, (hipStream_t)stream, a, C, mode, out, (size_t)npix);
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
