// Semantics check of v_permlane16_swap / v_permlane32_swap and DPP row shifts on gfx950 (prints per-lane results).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    const u32x2 a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
    o[256 + l] = (unsigned)__builtin_amdgcn_update_dpp(777, (int)l, 0x102, 0xf, 0xf, false);   // row_shl:2
    o[320 + l] = (unsigned)__builtin_amdgcn_update_dpp(777, (int)l, 0x112, 0xf, 0xf, false);   // row_shr:2
}
int main() {
    unsigned* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[6] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1", "shl2", "shr2"};
    for (int r = 0; r < 6; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
