// VALU instruction-rate micro-benchmark, part 5 (gfx950): are the fp16 transcendentals cheaper than the fp32 ones?  SimpleGate2's sigmoid is 32 of the
// 107 VALU instructions of a B wave's step in the phase-1 kernel and 13 % of the step's issue cycles.  Cycles per wave64 instruction per
// SIMD at 8 / 3 / 1 waves per SIMD, 16 independent streams.   build: hipcc --offload-arch=gfx950 -O3 -o valu_rate5 valu_rate5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void k(float* out, int iters, float fa) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i * 0.01f + fa;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
            else if (OP == 2) asm volatile("v_exp_f16 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 3) asm volatile("v_rcp_f16 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 4) a[i] = fa * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a[i]));             // the fp32 sigmoid gate: exp, add, rcp, mul
            else if (OP == 5) {                                                                                     // the same in fp16: cvt, exp, add, rcp, cvt, mul
                float h, e, s, r, f;
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(a[i]));
                asm volatile("v_exp_f16 %0, %1" : "=v"(e) : "v"(h));
                asm volatile("v_add_f16 %0, 1.0, %1" : "=v"(s) : "v"(e));
                asm volatile("v_rcp_f16 %0, %1" : "=v"(r) : "v"(s));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f) : "v"(r));
                a[i] = fa * f;
            }
            else if (OP == 6) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int per) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    printf("%-44s", name);
    for (int wps : {8, 3, 1}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, 10, 1.0001f);
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, iters, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * 4 / 1024.0 * iters * 16 * per;
        printf("  %d w/SIMD: %6.2f cyc", wps, ms * 1e-3 * 2.4e9 / winst);
    }
    printf("   (per %s per SIMD, at 2.4 GHz)\n", per == 1 ? "instruction" : "element");
    hipFree(out);
}

int main() {
    run<0>("v_exp_f32", 1);
    run<1>("v_rcp_f32", 1);
    run<2>("v_exp_f16", 1);
    run<3>("v_rcp_f16", 1);
    run<6>("v_sqrt_f32", 1);
    run<4>("gate b * rcp(1 + exp2(c)), fp32: 4 instr", 1);
    run<5>("gate in fp16 (cvt exp add rcp cvt mul): 6 instr", 1);
    return 0;
}
