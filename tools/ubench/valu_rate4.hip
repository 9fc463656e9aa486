// VALU rate micro-benchmark, part 4 (gfx950): v_fma_mixlo_f16 / v_fma_mixhi_f16 (packed fp16 storage, fp32 math, one half per instruction) against
// v_pk_fma_f16, at 8 / 3 / 1 waves per SIMD.  build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/vr4 valu_rate4.hip && /tmp/vr4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h2_t mixfma(h2_t v, h2_t w, h2_t c) {        // two v_fma_mix{lo,hi}_f16: per-half fp32 fma, fp16 in / out
    h2_t r;
    r[0] = (_Float16)__builtin_fmaf((float)v[0], (float)w[0], (float)c[0]);
    r[1] = (_Float16)__builtin_fmaf((float)v[1], (float)w[1], (float)c[1]);
    return r;
}
template <int OP>
__global__ void k(float* out, int iters, uint32_t wa, uint32_t wb) {
    h2_t d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = __builtin_bit_cast(h2_t, threadIdx.x * 77u + i * 13u + 0x3c003c00u);
    const h2_t w0 = __builtin_bit_cast(h2_t, wa), w1 = __builtin_bit_cast(h2_t, wb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) d[i] = mixfma(d[i], w0, w1);
            else if (OP == 1) d[i] = __builtin_elementwise_fma(d[i], w0, w1);
            else if (OP == 2) d[i] = mixfma(d[(i + 1) & 15], w0, d[i]);            // the stencil's form: acc += v * w, three different registers
            else if (OP == 3) d[i] = __builtin_elementwise_fma(d[(i + 1) & 15], w0, d[i]);
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= __builtin_bit_cast(uint32_t, d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(x & 0x3fffffff);
}
template <int OP>
void run(const char* name, int per) {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    printf("%-44s", name);
    for (int wps : {8, 3, 1}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, 10, 0x3c003c00u, 0x38003800u);
        (void)hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, iters, 0x3c003c00u, 0x38003800u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * 4 / 1024.0 * iters * 16 * per;
        printf("  %d w/SIMD: %6.2f cyc", wps, ms * 1e-3 * 2.4e9 / winst);
    }
    printf("   (per instruction per SIMD, at 2.4 GHz)\n");
}
int main() {
    run<0>("v_fma_mixlo_f16 + v_fma_mixhi_f16 (const)", 2);
    run<1>("v_pk_fma_f16 (const)", 1);
    run<2>("v_fma_mixlo/hi_f16 (acc += v * w)", 2);
    run<3>("v_pk_fma_f16 (acc += v * w)", 1);
    return 0;
}
