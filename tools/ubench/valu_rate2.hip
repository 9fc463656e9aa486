// VALU instruction-rate micro-benchmark, part 2 (gfx950): packed fp16 math, DPP moves, conversions -- the instruction mix of the fused
// phase-1 kernel (csrc/sn_phase1.hip).  Cycles per wave64 instruction per SIMD with 8 / 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, int iters, uint32_t wa, uint32_t wb) {
    uint32_t d[16];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { d[i] = threadIdx.x * 77u + i * 13u + 0x3c003c00u; a[i] = threadIdx.x * 0.001f + i; }
    const h2_t w0 = __builtin_bit_cast(h2_t, wa), w1 = __builtin_bit_cast(h2_t, wb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) d[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2_t, d[i]), w0, w1));          // v_pk_fma_f16, 2 sgpr-ish operands
            else if (OP == 1) d[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2_t, d[i]), __builtin_bit_cast(h2_t, d[(i + 1) & 15]), __builtin_bit_cast(h2_t, d[(i + 2) & 15])));   // 3 vgpr operands
            else if (OP == 2) d[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d[i], 0x111, 0xf, 0xf, true);                       // v_mov_b32_dpp row_shr:1 bound_ctrl
            else if (OP == 3) d[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)d[i], 0x121, 0xf, 0xf, false);                              // row_ror:1
            else if (OP == 4) d[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)d[(i + 1) & 15], (int)d[i], 0x111, 0xf, 0xf, false);    // row_shr:1 with old
            else if (OP == 5) { f32x2_t v = {a[i], a[(i + 1) & 15]}; d[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_t)); a[i] += 1.0f; }   // cvt_pk + add
            else if (OP == 6) d[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, d[i]) * w0);                                 // v_pk_mul_f16
            else if (OP == 7) a[i] = __builtin_fmaf(a[i], __builtin_bit_cast(float, wa), a[(i + 1) & 15]);                               // v_fma_f32 vgpr
            else if (OP == 8) d[i] = d[i] & wa;                                                                                          // v_and_b32
        }
    }
    float s = 0; uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s += a[i]; x ^= d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(x & 0x3fffffff);
}

template <int OP>
void run(const char* name) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    for (int wps : {8, 2}) {
        const int blocks = 256 * wps;                        // wps blocks of 256 threads per CU -> wps waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, 10, 0x3c003c00u, 0x38003800u);
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, iters, 0x3c003c00u, 0x38003800u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * 4 / 1024.0 * iters * 16;
        printf("%-34s %d waves/SIMD %8.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / winst);
    }
}

int main() {
    run<0>("v_pk_fma_f16 (const operands)");
    run<1>("v_pk_fma_f16 (3 vgprs)");
    run<2>("v_mov_b32_dpp row_shr bound_ctrl");
    run<3>("v_mov_b32_dpp row_ror");
    run<4>("v_mov_b32_dpp row_shr + old");
    run<5>("v_cvt_pk_f16_f32 + v_add_f32");
    run<6>("v_pk_mul_f16");
    run<7>("v_fma_f32 (vgprs)");
    run<8>("v_and_b32");
    return 0;
}
