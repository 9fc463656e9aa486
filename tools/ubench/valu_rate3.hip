// VALU instruction-rate micro-benchmark, part 3 (gfx950): the instruction types of the role-split phase-1 kernel's stager and epilogue code
// (packed f32 math, v_dot2c, transcendentals, bf16 / f16 packing), cycles per wave64 instruction per SIMD at 8 / 3 / 1 waves per SIMD,
// independent streams (16 registers in flight) -- i.e. issue cost, not latency.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate3 valu_rate3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, int iters, float fa, uint32_t wa) {
    f32x2_t a[16];
    uint32_t d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = (f32x2_t){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i}; d[i] = threadIdx.x * 77u + i * 13u + 0x3c003c00u; }
    const f32x2_t c2 = {fa, fa * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_elementwise_fma(a[i], c2, c2);                                   // v_pk_fma_f32
            else if (OP == 1) a[i] = a[i] + c2;                                                           // v_pk_add_f32
            else if (OP == 2) a[i] = a[i] * c2;                                                           // v_pk_mul_f32
            else if (OP == 3) a[i][0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, d[i]), __builtin_bit_cast(bf16x2_t, wa), a[i][0], false);   // v_dot2c_f32_bf16
            else if (OP == 4) a[i][0] = __builtin_amdgcn_exp2f(a[i][0]);                                  // v_exp_f32
            else if (OP == 5) a[i][0] = __builtin_amdgcn_rcpf(a[i][0]);                                   // v_rcp_f32
            else if (OP == 6) { bf16x2_t hh = __builtin_convertvector(a[i], bf16x2_t); d[i] = __builtin_bit_cast(uint32_t, hh); a[i][0] += 1.0f; }   // v_cvt_pk_bf16_f32 + v_add_f32
            else if (OP == 7) a[i][0] = __builtin_fmaf(a[i][0], fa, a[i][1]);                             // v_fma_f32
            else if (OP == 8) { a[i][0] = __builtin_fmaf(a[i][0], fa, fa); a[i][1] = __builtin_fmaf(a[i][1], fa, fa); }   // 2 x v_fma_f32 (= one pk_fma of work)
            else if (OP == 9) d[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2_t, d[i]), __builtin_bit_cast(h2_t, wa), __builtin_bit_cast(h2_t, wa)));   // v_pk_fma_f16
            else if (OP == 10) a[i][0] = __builtin_amdgcn_rsqf(a[i][0]);                                  // v_rsq_f32
            else if (OP == 11) d[i] = (d[i] << 16) ^ (d[i] & 0xffff0000u);                                // v_lshlrev + v_and + v_xor: 3 plain int ops
        }
    }
    float s = 0; uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s += a[i][0] + a[i][1]; x ^= d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(x & 0x3fffffff);
}

template <int OP>
void run(const char* name, int per) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    printf("%-38s", name);
    for (int wps : {8, 3, 1}) {
        const int blocks = 256 * wps;                        // wps blocks of 256 threads per CU -> wps waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, 10, 1.0001f, 0x3c003c00u);
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, iters, 1.0001f, 0x3c003c00u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * 4 / 1024.0 * iters * 16 * per;
        printf("  %d w/SIMD: %6.2f cyc", wps, ms * 1e-3 * 2.4e9 / winst);
    }
    printf("   (per instruction per SIMD, at 2.4 GHz)\n");
}

int main() {
    run<0>("v_pk_fma_f32", 1);
    run<1>("v_pk_add_f32", 1);
    run<2>("v_pk_mul_f32", 1);
    run<3>("v_dot2c_f32_bf16", 1);
    run<4>("v_exp_f32", 1);
    run<5>("v_rcp_f32", 1);
    run<10>("v_rsq_f32", 1);
    run<6>("v_cvt_pk_bf16_f32 + v_add_f32", 2);
    run<7>("v_fma_f32", 1);
    run<8>("2 x v_fma_f32", 2);
    run<9>("v_pk_fma_f16", 1);
    run<11>("v_lshlrev + v_and + v_xor", 3);
    return 0;
}
