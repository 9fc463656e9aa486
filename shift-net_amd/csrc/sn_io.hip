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

}  // namespace

extern "C" {

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
