// VALU instruction-rate micro-benchmark for gfx950: how many cycles does a wave64 instruction of each kind cost?
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, int iters, uint32_t wa, uint32_t wb) {
    float a[16];
    uint32_t d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; d[i] = threadIdx.x * 77u + i * 13u; }
    float w0 = __uint_as_float(wa), w1 = __uint_as_float(wb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], w0, w1);                                   // v_fma_f32 (sgpr operands)
            else if (OP == 1) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, d[i]), __builtin_bit_cast(bf16x2_t, wa), a[i], false);
            else if (OP == 2) { d[i] = (d[i] << 16) ^ wa; }                                     // shift (unpack-like)
            else if (OP == 3) { d[i] = (d[i] & 0xffff0000u) + wb; }                             // and (unpack-like)
            else if (OP == 4) a[i] = __expf(a[i]) ;                                             // v_exp_f32 (+mul)
            else if (OP == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);                               // v_rcp_f32
            else if (OP == 6) a[i] = a[i] * w0;                                                 // v_mul_f32
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
    const int iters = 2000, blocks = 256 * 8;           // 8 blocks of 256 threads per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10, 0x3f800000u, 0x3f000000u);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 0x3f800000u, 0x3f000000u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * iters*16
    double winst = (double)blocks * 4 / 1024.0 * iters * 16;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc / winst);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_dot2c_f32_bf16");
    run<2>("v_lshl+xor (2 ops)");
    run<3>("v_and+add (2 ops)");
    run<4>("expf (v_mul+v_exp)");
    run<5>("v_rcp_f32");
    run<6>("v_mul_f32");
    return 0;
}
