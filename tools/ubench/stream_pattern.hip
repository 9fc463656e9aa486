// Streaming rate of "read two 128-byte-per-pixel tensors, write one" (K4's traffic: g2 + shortcut -> y, C = 64 bf16) against the ADDRESS PATTERN of a
// wave's 16-byte accesses (gfx950).  A lane (g = lane >> 4, p = lane & 15) of K4 owns 16 consecutive channels of pixel p: its shortcut loads and y stores
// are two 16-byte pieces at byte offsets g * 32 and g * 32 + 16 -- per INSTRUCTION a 16-byte checkerboard over each pixel's 128-byte record
// (pattern 0).  Pattern 1: per instruction the four lanes of a pixel cover one contiguous 64-byte half (offsets g * 16, then 64 + g * 16).
// Pattern 2: one thread per 16-byte piece, consecutive lanes = consecutive pieces (the plain copy).   W waves per SIMD via launch bounds / grid.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/sp stream_pattern.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT, int NT>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y, long npix) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, p = lane & 15;
    const long base = ((long)blockIdx.x * 4 + wv) * (16 * NT);          // a wave: NT tiles of 16 pixels (K4: NT = 4)
    uint4 va[NT][2], vb[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const long px = base + n * 16 + p;
        if (px >= npix) return;
        // tensor a: K4's B-operand pattern (k-step s: 64 contiguous bytes per pixel) for every variant
        va[n][0] = a[px * 8 + g]; va[n][1] = a[px * 8 + 4 + g];
        if (PAT == 0) { vb[n][0] = b[px * 8 + g * 2]; vb[n][1] = b[px * 8 + g * 2 + 1]; }
        else { vb[n][0] = b[px * 8 + g]; vb[n][1] = b[px * 8 + 4 + g]; }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const long px = base + n * 16 + p;
        uint4 o0 = va[n][0], o1 = va[n][1];
        o0.x ^= vb[n][0].x; o0.y += vb[n][0].y; o0.z ^= vb[n][0].z; o0.w += vb[n][0].w;
        o1.x ^= vb[n][1].x; o1.y += vb[n][1].y; o1.z ^= vb[n][1].z; o1.w += vb[n][1].w;
        if (PAT == 0) { y[px * 8 + g * 2] = o0; y[px * 8 + g * 2 + 1] = o1; }
        else { y[px * 8 + g] = o0; y[px * 8 + 4 + g] = o1; }
    }
}
__global__ __launch_bounds__(256) void kflat(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y, long n16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    uint4 o = a[i]; const uint4 q = b[i];
    o.x ^= q.x; o.y += q.y; o.z ^= q.z; o.w += q.w;
    y[i] = o;
}
int main() {
    const long npix = 20L * 360 * 640;                 // a level-1 tensor of config 2: 590 MB at 128 B per pixel
    const size_t bytes = (size_t)npix * 128;
    uint4 *a, *b, *y;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&y, bytes);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](auto launch, const char* name) {
        for (int rep = 0; rep < 2; ++rep) {
            launch();
            (void)hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-62s %7.1f us  %5.2f TB/s\n", name, ms * 100.f, 3.0 * bytes / (ms * 1e-4) / 1e12);
        }
    };
    for (int round = 0; round < 2; ++round) {
        time([&] { k<0, 4><<<(unsigned)((npix + 255) / 256), 256>>>(a, b, y, npix); }, "pattern 0 (K4 today: 16-byte checkerboard), 4 tiles per wave");
        time([&] { k<1, 4><<<(unsigned)((npix + 255) / 256), 256>>>(a, b, y, npix); }, "pattern 1 (64 contiguous bytes per pixel and instruction), 4");
        time([&] { k<0, 2><<<(unsigned)((npix + 127) / 128), 256>>>(a, b, y, npix); }, "pattern 0, 2 tiles per wave");
        time([&] { k<1, 2><<<(unsigned)((npix + 127) / 128), 256>>>(a, b, y, npix); }, "pattern 1, 2 tiles per wave");
        time([&] { k<1, 1><<<(unsigned)((npix + 63) / 64), 256>>>(a, b, y, npix); }, "pattern 1, 1 tile per wave");
        time([&] { kflat<<<(unsigned)((npix * 8 + 255) / 256), 256>>>(a, b, y, npix * 8); }, "flat: one 16-byte piece per thread, consecutive lanes");
    }
    return 0;
}
