// Do MFMA and VALU work of DIFFERENT waves on the same SIMD overlap (gfx950)?  One 512-thread workgroup per CU (8 waves: waves w and w + 4 share
// a SIMD): waves 0-3 run an MFMA-only loop, waves 4-7 a VALU-only loop (v_pk_fma_f16 or plain v_fma_f32).  Times: MFMA alone, VALU alone, both.
// If "both" ~ max(alone) the pipes overlap; if ~ sum they serialise.  Also: ONE wave per SIMD interleaving both streams (ILP inside a wave).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl mfma_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int MODE, int VKIND>      // MODE bit 0: MFMA waves work, bit 1: VALU waves work, 4: every wave does both, interleaved
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wv = threadIdx.x >> 6;
    f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
    h2_t d[16]; float f[16];
    for (int i = 0; i < 16; ++i) { d[i] = __builtin_bit_cast(h2_t, threadIdx.x * 77u + i * 13u + 0x3c003c00u); f[i] = threadIdx.x * 0.001f + i; }
    const h2_t w0 = {(_Float16)1.0009765625f, (_Float16)0.99951171875f};
    const bool do_m = MODE == 4 || ((MODE & 1) && wv < 4), do_v = MODE == 4 || ((MODE & 2) && wv >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
                if (VKIND == 0) { d[i] = __builtin_elementwise_fma(d[(i + 1) & 15], w0, d[i]); d[(i + 8) & 15] = __builtin_elementwise_fma(d[(i + 9) & 15], w0, d[(i + 8) & 15]); }
                else { f[i] = __builtin_fmaf(f[(i + 1) & 15], 1.0001f, f[i]); f[(i + 8) & 15] = __builtin_fmaf(f[(i + 9) & 15], 1.0001f, f[(i + 8) & 15]); }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    if (VKIND == 0) d[i & 15] = __builtin_elementwise_fma(d[(i + 1) & 15], w0, d[i & 15]);
                    else f[i & 15] = __builtin_fmaf(f[(i + 1) & 15], 1.0001f, f[i & 15]);
                }
            }
        }
    }
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    uint32_t x = 0;
    for (int i = 0; i < 16; ++i) { s += f[i]; x ^= __builtin_bit_cast(uint32_t, d[i]); }
    out[blockIdx.x * 512 + threadIdx.x] = s + __uint_as_float(x & 0x3fffffff);
}
template <int MODE, int VKIND> float run(float* out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, VKIND><<<256, 512>>>(out, 10);
    (void)hipEventRecord(e0);
    k<MODE, VKIND><<<256, 512>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int it = 4000;
    // per iteration and SIMD: 16 MFMAs (one MFMA wave) and 32 VALU instructions (one VALU wave)
    for (int vk = 0; vk < 2; ++vk) {
        float m, v, both, ilp;
        if (vk == 0) { m = run<1, 0>(out, it); v = run<2, 0>(out, it); both = run<3, 0>(out, it); ilp = run<4, 0>(out, it); }
        else { m = run<1, 1>(out, it); v = run<2, 1>(out, it); both = run<3, 1>(out, it); ilp = run<4, 1>(out, it); }
        const double cyc = 2.4e9 * 1e-3 / it;
        printf("%s: MFMA-only waves %.0f cyc/iter (%.1f per MFMA)   VALU-only waves %.0f cyc/iter (%.1f per instr)   both kinds of waves %.0f cyc/iter   sum %.0f  max %.0f\n",
               vk == 0 ? "v_pk_fma_f16" : "v_fma_f32   ", m * cyc, m * cyc / 16, v * cyc, v * cyc / 32, both * cyc, (m + v) * cyc, (m > v ? m : v) * cyc);
        printf("              every wave interleaves 16 MFMA + 32 VALU per iteration (8 waves): %.0f cyc/iter for 2 x (16 MFMA + 32 VALU) per SIMD\n", ilp * cyc);
    }
    return 0;
}
