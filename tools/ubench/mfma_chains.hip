// v_mfma_f32_16x16x32_bf16 issue rate of ONE wave per SIMD against the number of independent accumulator chains (gfx950), and of two waves per SIMD.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mch mfma_chains.hip && /tmp/mch
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ void k(float* out, int iters) {
    f32x4_t acc[NCH];
    for (int i = 0; i < NCH; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % NCH], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH> void run(float* out) {
    for (int wps : {1, 2, 3}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int it = 4000;
        k<NCH><<<256, 256 * wps>>>(out, 10);
        (void)hipEventRecord(e0);
        k<NCH><<<256, 256 * wps>>>(out, it);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("chains %2d, %d wave(s)/SIMD: %.1f cyc per MFMA per SIMD (at 2.4 GHz)\n", NCH, wps, ms * 1e-3 * 2.4e9 / (it * 16.0 * wps));
    }
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    run<1>(out); run<2>(out); run<4>(out); run<8>(out); run<16>(out);
    return 0;
}
