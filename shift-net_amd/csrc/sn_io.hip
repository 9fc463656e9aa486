// I/O edges of the inference CLIs on the device (gfx950): uint8 frames in, uint8 frames + PSNR sums out.
//
// Upstream converts on the host (inference/test_deblur.py:191-200 numpy2tensor: uint8 HWC -> float64 -> CHW -> float32
// -> x 1/255 -> stack), ships float32 over PCIe (12 B per pixel) and casts on the GPU (:134 in_tensor.half()); results
// come back as float frames for skimage PSNR against the uint8 ground truth (:139-143) and cv2.imwrite (:152).
// Here the uint8 frames cross PCIe (3 B per pixel) and both conversions are HBM-bound kernels:
//   sn_ingest_u8 : [T][H][W][3] u8 -> [T][3][H][W] of the module dtype, value = round_dtype(float(v) * (1/255))  -- bit
//                  identical to numpy2tensor(...).to(dtype);
//   sn_egress_u8 : [T][3][H][W] of the module dtype -> clamp(0,1) * 255 -> (a) [T][H][W][3] u8, rounded to nearest even
//                  like cv2.imwrite's saturate_cast, (b) per-frame sums of squared error of the UNROUNDED value against
//                  the uint8 ground truth (what skimage's PSNR with data_range=255 is computed from), as deterministic
//                  per-workgroup partial sums.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"
#include <math.h>

namespace {

__device__ __forceinline__ float ld_any(const void* p, int dt, size_t i) {
    return dt == SN_F32 ? ((const float*)p)[i] : (dt == SN_F16 ? __half2float(((const __half*)p)[i]) : bf_to_f(((const bf16_t*)p)[i]));
}
__device__ __forceinline__ void st_any(void* p, int dt, size_t i, float v) {
    if (dt == SN_F32) ((float*)p)[i] = v;
    else if (dt == SN_F16) ((__half*)p)[i] = __float2half(v);
    else ((bf16_t*)p)[i] = f_to_bf(v);
}

__global__ __launch_bounds__(256) void ingest_u8_kernel(const uint8_t* __restrict__ src, void* dst, int dt, int HW) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const uint8_t* s = src + ((size_t)t * HW + i) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) st_any(dst, dt, ((size_t)t * 3 + c) * HW + i, (float)s[c] * (1.0f / 255.0f));
}

#define SN_EGRESS_BLOCKS 64
__global__ __launch_bounds__(256) void egress_u8_kernel(const void* __restrict__ out, int dt, const uint8_t* __restrict__ gt,
                                                      uint8_t* __restrict__ img, float* sse, int HW) {
    __shared__ float red[4];
    const int t = blockIdx.y, tid = threadIdx.x;
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + tid; i < HW; i += SN_EGRESS_BLOCKS * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = ld_any(out, dt, ((size_t)t * 3 + c) * HW + i);
            v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
            const size_t o = ((size_t)t * HW + i) * 3 + c;
            if (img) img[o] = (uint8_t)__float2int_rn(v);
            if (gt) { const float d = v - (float)gt[o]; acc = fmaf(d, d, acc); }
        }
    }
    acc = sum_rows4(row_sum16(acc));
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && sse) sse[(size_t)t * SN_EGRESS_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- the CLIs' SSIM (inference/test_deblur.py:25-49) on the device ----------------------------------------------------------
// ssim_calculate(img1, img2): a = img1 / 255, b = img2 / 255 as (C,H,W) float32 volumes; mu = G(a), G(b); sigma = G(a*a) - mu^2,
// G(b*b) - mu^2, G(a*b) - mu1*mu2 with G = scipy.ndimage.gaussian_filter(sd = 1.5): a separable 13-tap Gaussian (radius
// int(4 * 1.5 + 0.5) = 6) along ALL THREE axes -- the 3-channel axis included -- with 'reflect' boundaries (d c b a | a b c d | d c b a);
// the result is the mean of the SSIM map over the volume.  Two kernels: W pass of the 15 planes (5 statistics x 3 channels) into
// scratch, then H pass + channel mixing (the 13 taps folded onto 3 channels by the reflection: a 3x3 matrix) + the SSIM formula +
// a deterministic block reduction.
struct SsimW { float g[13]; float m[3][3]; };

__device__ __forceinline__ int reflect_idx(int i, int n) {     // scipy 'reflect' (half-sample symmetric), any distance
    const int period = 2 * n;
    int m = i % period; if (m < 0) m += period;
    return m < n ? m : period - 1 - m;
}

__global__ __launch_bounds__(256) void ssim_wpass_kernel(const void* __restrict__ out, int dt, const uint8_t* __restrict__ gt, float* __restrict__ tmp,
                                                       const SsimW K, int H, int W) {
    const int t = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const size_t HW = (size_t)H * W;
    for (int c = 0; c < 3; ++c) {
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = -6; k <= 6; ++k) {
            const int xx = reflect_idx(x + k, W);
            float a = ld_any(out, dt, ((size_t)t * 3 + c) * HW + (size_t)y * W + xx);
            a = (fminf(fmaxf(a, 0.f), 1.f) * 255.0f) / 255.0f;                  // the CLI: clamp(0,1) * 255, then / 255 inside ssim_calculate
            const float b = (float)gt[((size_t)t * HW + (size_t)y * W + xx) * 3 + c] / 255.0f;
            const float gk = K.g[k + 6];
            s[0] = fmaf(gk, a, s[0]); s[1] = fmaf(gk, b, s[1]); s[2] = fmaf(gk, a * a, s[2]); s[3] = fmaf(gk, b * b, s[3]); s[4] = fmaf(gk, a * b, s[4]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) tmp[(((size_t)t * 15 + c * 5 + q) * H + y) * W + x] = s[q];
    }
}

#define SN_SSIM_BLOCKS 128
__global__ __launch_bounds__(256) void ssim_hpass_kernel(const float* __restrict__ tmp, float* partial, const SsimW K, int H, int W) {
    __shared__ float red[4];
    const int t = blockIdx.y, tid = threadIdx.x;
    const int HW = H * W;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + tid; i < HW; i += SN_SSIM_BLOCKS * 256) {
        const int y = i / W, x = i - y * W;
        float v[15];
#pragma unroll
        for (int pl = 0; pl < 15; ++pl) {
            float sm = 0.f;
            const float* base = tmp + ((size_t)t * 15 + pl) * HW + x;
#pragma unroll
            for (int k = -6; k <= 6; ++k) sm = fmaf(K.g[k + 6], base[(size_t)reflect_idx(y + k, H) * W], sm);
            v[pl] = sm;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float f[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) f[q] = K.m[c][0] * v[q] + K.m[c][1] * v[5 + q] + K.m[c][2] * v[10 + q];
            const float mu1 = f[0], mu2 = f[1];
            const float s1 = f[2] - mu1 * mu1, s2 = f[3] - mu2 * mu2, s12 = f[4] - mu1 * mu2;
            acc += ((2.f * mu1 * mu2 + c1) * (2.f * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2));
        }
    }
    acc = sum_rows4(row_sum16(acc));
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[(size_t)t * SN_SSIM_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" {

int sn_ssim_blocks(void) { return SN_SSIM_BLOCKS; }

int sn_ssim_u8(const void* out, int out_dtype, const uint8_t* gt, float* scratch, float* partial, int T, int H, int W, void* stream) {
    sn_clear_error();
    if (!out || !gt || !scratch || !partial || out_dtype < 0 || out_dtype > 2 || T < 1 || H < 1 || W < 1) return SN_EINVAL;
    SsimW K;
    double g[13], sum = 0.0;
    for (int k = -6; k <= 6; ++k) { g[k + 6] = exp(-0.5 * k * k / (1.5 * 1.5)); sum += g[k + 6]; }      // scipy _gaussian_kernel1d(1.5, 0, 6)
    for (int k = 0; k < 13; ++k) K.g[k] = (float)(g[k] / sum);
    for (int c = 0; c < 3; ++c) {
        double m[3] = {0.0, 0.0, 0.0};
        for (int k = -6; k <= 6; ++k) {
            int i = (c + k) % 6; if (i < 0) i += 6;
            m[i < 3 ? i : 5 - i] += g[k + 6] / sum;
        }
        for (int j = 0; j < 3; ++j) K.m[c][j] = (float)m[j];
    }
    hipLaunchKernelGGL(ssim_wpass_kernel, dim3((W + 255) / 256, H, T), dim3(256), 0, (hipStream_t)stream, out, out_dtype, gt, scratch, K, H, W);
    hipLaunchKernelGGL(ssim_hpass_kernel, dim3(SN_SSIM_BLOCKS, T), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, partial, K, H, W);
    return sn_check_launch();
}

int sn_ingest_u8(const uint8_t* src, void* dst, int dst_dtype, int T, int H, int W, void* stream) {
    sn_clear_error();
    if (!src || !dst || dst_dtype < 0 || dst_dtype > 2 || T < 1 || H < 1 || W < 1) return SN_EINVAL;
    const int hw = H * W;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3((hw + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, src, dst, dst_dtype, hw);
    return sn_check_launch();
}

int sn_egress_blocks(void) { return SN_EGRESS_BLOCKS; }

int sn_egress_u8(const void* out, int out_dtype, const uint8_t* gt, uint8_t* img, float* sse, int T, int H, int W, void* stream) {
    sn_clear_error();
    if (!out || out_dtype < 0 || out_dtype > 2 || T < 1 || H < 1 || W < 1 || (gt && !sse) || (!img && !gt)) return SN_EINVAL;
    hipLaunchKernelGGL(egress_u8_kernel, dim3(SN_EGRESS_BLOCKS, T), dim3(256), 0, (hipStream_t)stream, out, out_dtype, gt, img, sse, H * W);
    return sn_check_launch();
}

}  // extern "C"
