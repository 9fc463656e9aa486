// Fused phase 1 of a CAB1 / CAB2 block of a GSTS unit (gfx950), depthwise variants (C = 64, deblur):
//     g2 = SimpleGate2(body[4](RepConv(SimpleGate(RepConv2(body[0](LayerNorm2d(u)))))))      (gshift_deblur2.py:186-258)
// in ONE kernel: u is read once, g2 is written once, and neither the 2C-channel tensor `a`, nor g1, nor r ever reaches HBM or even LDS
// (only r crosses LDS once, for the second 1x1).  Phase 2 (CALayer2 scale -> 1x1 -> beta-residual) is sn_scale_gemm_res; the global
// average pool of CALayer2 sits between the two and is the one unavoidable grid-wide dependency of the block.
//
// Idea.  With the weights as the MFMA A operand, the accumulators of a 1x1 conv are laid out D[channel 4g + r][pixel p]: a lane group g
// owns 4 channels, the 16 lanes of a DPP row are 16 CONSECUTIVE PIXELS of one image row.  A depthwise stencil on such a tensor needs
// only (i) the same registers of the rows above / below -- a sliding window over rows that a wave keeps in registers while it walks down
// a column strip -- and (ii) the neighbour pixels, i.e. the same register of the neighbour LANE: one v_mov_b32_dpp row shift (plus one
// row rotate of the adjacent 16-pixel tile for the lanes at the tile edge).  So the whole chain runs register to register:
//
//   wave q of a 4-wave team owns gate pair q of the first 1x1 (M-tiles 2q, 2q+1: a-channels 16g+4q+r and their partners C + ...),
//   hence g1 / r channels 16g+4q+r, and gate pair q of the second 1x1 (g2 channels 16g+4q+r); all four waves cover the SAME pixels.
//   Per input row y of a (16 NX)-pixel wide region (NX N-tiles side by side; the outer 3 columns on each side are halo):
//     raw bf16 pixels: staged ONCE per team through LDS one row ahead (thread = pixel x 8-channel piece)  -> B fragments, NO unpack / normalise pass
//     LayerNorm statistics from the raw pieces while staging (v_dot2c with ones / with itself, quad reduction), once per team
//     1x1 on the RAW operands; LayerNorm applied AFTER it: a = rstd (W v - mu W 1) + b       (W 1 = row sums of the bf16 weights, host)
//     a -> packed fp16 (two channels per register), zero outside the image
//     3x3 (+identity) as v_pk_fma_f16 in SCATTER form: row y completes the pending output row y-1 and opens row y+1
//     SimpleGate -> g1 row y-1 (packed fp16; the first factor carries 2^-4 so that the product stays in fp16 range)
//     5x5 (+3x3 +identity folded) the same way: g1 row y-1 completes r row y-3
//     r row (fp16) -> LDS ring slot, ONE workgroup barrier, every wave reads all 64 channels back as B fragments of the second 1x1
//     (fp16 MFMA, weights x 2^4), SimpleGate2, 8-byte NHWC stores of g2 row y-3, channel sums for CALayer2.
//   Nothing is recomputed vertically (only the 6 warm-up rows of a row segment), horizontally the region overlaps by 6 of 16 NX columns.
//
// Numerics: `a` and the 3x3 run in fp16 exactly as sn_ln_gemm_gate does; g1, the 5x5 and r run in fp16 (g1 was bf16 with fp32
// accumulation in sn_dw5m_gemm_gate: 3 more mantissa bits per operand, fp16 instead of fp32 accumulation over 25 taps); the first 1x1
// sees the un-normalised bf16 input and fp32 statistics, i.e. the normalised operand is no longer rounded to bf16.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct P1Args {
    const bf16_t* x; const bf16_t* hwb;
    int T, h, w, mode, wrap;
    const uint4* wfrag1; const float* bias; const float* wsum;
    const uint32_t* w3; const uint32_t* w5;
    const uint4* wfrag2;
    bf16_t* g2; float* pool;
    int nsx, nsy, seg, vw;
};

__device__ __forceinline__ f32x4_t mfma16h(const uint4 a, const uint4 b, const f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t cvt_pk_h2(float lo, float hi) {           // v_cvt_pk_f16_f32, round to nearest even
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_t));
}
__device__ __forceinline__ h2_t as_h2(uint32_t u) { return __builtin_bit_cast(h2_t, u); }
__device__ __forceinline__ uint32_t as_u(h2_t h) { return __builtin_bit_cast(uint32_t, h); }

// Register of the pixel N columns to the LEFT (x - N) for every lane of a 16-pixel tile: lanes p >= N take cur[p - N] (row_shr:N), lanes
// p < N keep `old` = the left neighbour tile rotated so that its last N lanes land on lanes 0..N-1 (row_ror:N).  At the region's left
// edge there is no neighbour: zeros (those columns are halo, their results are never used).
template <int N> __device__ __forceinline__ uint32_t from_left(uint32_t prev, uint32_t cur, bool has_prev) {
    if (!has_prev) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x110 + N, 0xf, 0xf, true);
    const int t = __builtin_amdgcn_mov_dpp((int)prev, 0x120 + N, 0xf, 0xf, false);      // every lane has a source in a rotate: no `old` value needed
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)cur, 0x110 + N, 0xf, 0xf, false);
}
// ... N columns to the RIGHT (x + N): lanes p < 16 - N take cur[p + N] (row_shl:N), the others the right neighbour's first N lanes (row_ror:16-N)
template <int N> __device__ __forceinline__ uint32_t from_right(uint32_t next, uint32_t cur, bool has_next) {
    if (!has_next) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x100 + N, 0xf, 0xf, true);
    const int t = __builtin_amdgcn_mov_dpp((int)next, 0x120 + 16 - N, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)cur, 0x100 + N, 0xf, 0xf, false);
}

#ifndef P1_ABLATE        // development only (tools/p1_variants.py): skip parts of the kernel to see what the time is made of; results are then wrong
#define P1_ABLATE 0
#endif
#define P1_ON(bit) (!(P1_ABLATE & (bit)))
#define P1_FENCE() do { if (!P1_ON(256)) __builtin_amdgcn_sched_barrier(0); } while (0)      // fences between the stencil groups measured 19 % SLOWER: off
constexpr int P1_PSO = 144;          // LDS bytes per pixel of a finished g2 row
constexpr int P1_PSR = 160;          // LDS bytes per pixel of an r row: 128 + 32 (10 slots of 16 B, 2 mod 4: conflict-free ds_read_b128 lane groups)

template <int KS, int NX>
__global__ __launch_bounds__(256, 2) void cab_phase1_kernel(const P1Args A) {
    constexpr int C = 64, CH = 32, K = 32 * KS, MT = 8, RWD = 16 * NX;
    constexpr int PSX = KS == 2 ? 160 : 224;                                  // LDS bytes per pixel of a staged input row (10 / 14 slots: 2 mod 4)
    static_assert(KS == 2 || KS == 3, "K = C (CAB1) or C + C/2 (CAB2)");
    static_assert(RWD * 4 <= 256, "staging: one thread per (region pixel, 8-channel piece of a k-step)");
    // stencil weights, packed fp16 pairs, one 32-byte record per (pass, kernel row): [wave][g][pass][row][8 words]
    //   3x3 pass kp: words 2 tx + kk = tap (row, tx) of a-register k = kp + 2 kk (kk = 0: channels (4g+.. r = 2kp, 2kp+1), kk = 1: their gate partners)
    //   5x5 pass k : words tx = tap (row, tx) of g1-register k (channels r = 2k, 2k+1)
    __shared__ __attribute__((aligned(16))) uint32_t lds_w3[4][4][2][3][8];
    __shared__ __attribute__((aligned(16))) uint32_t lds_w5[4][4][2][5][8];
    __shared__ __attribute__((aligned(16))) uint4 lds_w2[4][2][2][64];       // [wave][M-tile of the pair][k-step][lane]: second 1x1, fp16 fragments
    __shared__ __attribute__((aligned(16))) float4 lds_wb[4][4][4];          // [wave][g]{row sums, bias} x {M-tile 2q, 2q+1}: LayerNorm epilogue constants
    __shared__ __attribute__((aligned(16))) char lds_x[2][RWD * PSX];        // staged raw input rows (ring of 2)
    __shared__ __attribute__((aligned(8))) float2 lds_st[2][RWD];            // (rstd, -rstd * mean) per pixel of the staged row
    __shared__ __attribute__((aligned(16))) char lds_r[2][RWD * P1_PSR];     // r rows (ring of 2)
    __shared__ __attribute__((aligned(16))) char lds_o[2][RWD * P1_PSO];     // finished g2 rows (ring of 2): every wave holds 4 of a pixel's 64 channels,
                                                                              // the rows leave as 16-byte pieces of whole 128-byte pixels
    const int tid = threadIdx.x, lane = tid & 63, q = wave_id(), g = lane >> 4, p = lane & 15;
    const int b = blockIdx.x, sx = b % A.nsx, sy = (b / A.nsx) % A.nsy, t = b / (A.nsx * A.nsy);
    const int x0 = sx * A.vw, Y0 = sy * A.seg, Y1 = Y0 + A.seg < A.h ? Y0 + A.seg : A.h;
    if (Y0 >= A.h) return;                                                    // workgroup-uniform
    const int h = A.h, w = A.w, hw = h * w;

    // ---- per-wave constants -> LDS (wave-private slices: no barrier needed, the LDS operations of one wave execute in order) ----
    for (int e = lane; e < 4 * 2 * 3 * 8; e += 64) (&lds_w3[q][0][0][0][0])[e] = A.w3[q * (4 * 2 * 3 * 8) + e];
    for (int e = lane; e < 4 * 2 * 5 * 8; e += 64) (&lds_w5[q][0][0][0][0])[e] = A.w5[q * (4 * 2 * 5 * 8) + e];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 2; ++s) lds_w2[q][m][s][lane] = A.wfrag2[((2 * q + m) * 2 + s) * 64 + lane];
    bf16x8_t W1[2][KS];                                                       // first 1x1: M-tiles 2q, 2q+1 (bf16), resident
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        W1[0][s] = as_frag(A.wfrag1[((2 * q) * KS + s) * 64 + lane]);
        W1[1][s] = as_frag(A.wfrag1[((2 * q + 1) * KS + s) * 64 + lane]);
    }
    if (lane < 16) {                                                          // lane = 4 g' + i: (row sums | bias) x (M-tile 2q | 2q+1) of lane group g'
        const int gg = lane >> 2, i = lane & 3;
        lds_wb[q][gg][i] = *(const float4*)((i < 2 ? A.wsum : A.bias) + gg * 4 * MT + (2 * q + (i & 1)) * 4);
    }

    // ---- staging role: thread = (region pixel spx, piece c4): it moves the 8-channel pieces [32 s + 8 c4, +8) of the virtual input u
    //       of its pixel, s = 0..KS-1 -- exactly the B fragments of lane group g = c4 -- HBM -> registers -> LDS, one row
    //      AHEAD, and derives the LayerNorm statistics of the pixel (each input row is read and reduced once per team, not once per wave)
    const int spx = tid >> 2, c4 = tid & 3;
    const bool stager = tid < RWD * 4;                                        // (NX = 3: the last wave has no staging work)
    int f0 = t, o0 = 0, f1 = t, o1 = CH;
    if (A.mode == 1) { if (t > 0 || A.wrap) { f0 = sn_prev_frame(t, A.T, A.wrap); o0 = CH; f1 = t; o1 = 0; } }
    else if (A.mode == 2) { if (t < A.T - 1 || A.wrap) { f0 = t; o0 = CH; f1 = sn_next_frame(t, A.T, A.wrap); o1 = 0; } }
    // wave-uniform 64-bit frame bases (scalar registers) + 32-bit per-lane element offsets (a frame has < 2^31 elements)
    const bf16_t* slab[KS];
    int sstride[KS], soff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s == 0) { slab[s] = A.x + (ptrdiff_t)f0 * hw * C; soff[s] = o0 + 8 * c4; sstride[s] = C; }
        else if (s == 1) { slab[s] = A.x + (ptrdiff_t)f1 * hw * C; soff[s] = o1 + 8 * c4; sstride[s] = C; }
        else { slab[s] = A.hwb + (size_t)t * hw * CH; soff[s] = 8 * c4; sstride[s] = CH; }
    }
    const int sgx = x0 - 3 + spx, sgxc = (sgx >= 0 && sgx < w) ? sgx : 0;    // clamped column: loads are unconditional, masks come later
    // Input rows in flight: DIST = 2 rows ahead with two register sets that swap roles every iteration (loop unrolled by two) when the
    // registers allow it (CAB1: 247 VGPRs); CAB2 (three k-steps per row) spills with two sets -- and a spill reload issued behind the
    // prefetch waits for it (vmcnt retires in order) -- so it fetches one row ahead.
    constexpr int DIST = KS == 2 ? 2 : 1;
    uint4 XA[KS], XB[DIST == 2 ? KS : 1];
    auto issue_row = [&](int y, uint4* X) {
        const int yc = (y >= 0 && y < h) ? y : 0;
        const int ii = stager ? yc * w + sgxc : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) X[s] = *(const uint4*)(slab[s] + ((P1_ON(512) ? ii : 0) * sstride[s] + soff[s]));
    };
    auto stage_row = [&](int slot, const uint4* Xr) {                         // registers -> LDS + statistics of the pixel
        if (!stager) return;                                                  // wave-uniform
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *(uint4*)(lds_x[slot] + spx * PSX + (32 * s + 8 * c4) * 2) = Xr[s];
            const uint32_t wd[4] = {Xr[s].x, Xr[s].y, Xr[s].z, Xr[s].w};
#pragma unroll
            for (int i = 0; i < (P1_ON(1) ? 4 : 0); ++i) { s1 = dot2bf(wd[i], 0x3f803f80u, s1); s2 = dot2bf(wd[i], wd[i], s2); }
        }
        s1 += dpp_mov<0xB1>(s1); s1 += dpp_mov<0x4E>(s1);                     // sum over the 4 lanes of the pixel (quad_perm)
        s2 += dpp_mov<0xB1>(s2); s2 += dpp_mov<0x4E>(s2);
        const float mean = s1 * (1.0f / K);
        const float var = fmaxf(s2 * (1.0f / K) - mean * mean, 0.f);
        const float rstd = __builtin_amdgcn_rsqf(var + 1e-6f);
        if (c4 == 0) lds_st[slot][spx] = make_float2(rstd, -rstd * mean);
    };

    // region column of lane p in N-tile n = 16 n + p  <->  image column gx = x0 - 3 + 16 n + p
    bool colin[NX];
#pragma unroll
    for (int n = 0; n < NX; ++n) { const int gx = x0 - 3 + 16 * n + p; colin[n] = gx >= 0 && gx < w; }

    // pending (partially accumulated) output rows of the two stencils, packed fp16
    h2_t P0[NX][4], P1[NX][4];                   // 3x3 on a: when row y arrives P1 = row y-1 (lacks row y), P0 = row y (lacks rows y, y+1)
    h2_t Q0[NX][2], Q1[NX][2], Q2[NX][2], Q3[NX][2];      // 5x5 on g1: Q3 completes next
    const h2_t hz = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int n = 0; n < NX; ++n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { P0[n][k] = hz; P1[n][k] = hz; }
#pragma unroll
        for (int k = 0; k < 2; ++k) { Q0[n][k] = hz; Q1[n][k] = hz; Q2[n][k] = hz; Q3[n][k] = hz; }
    }
    float psum[4] = {0.f, 0.f, 0.f, 0.f};
    const int nit = (Y1 - Y0) + 6;
    // development only (P1_ABLATE & 2048): s_memtime clocks per phase of the row loop, written over the pool entries at the end
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = P1_ON(2048) ? 0 : __builtin_amdgcn_s_memtime();
    auto tick = [&](int slot) {
        if (!P1_ON(4096)) __builtin_amdgcn_sched_barrier(0);                  // variant: scheduling fences at the phase boundaries only
        if (!P1_ON(2048)) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tacc[slot] += now - tlast; tlast = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue_row(Y0 - 3, XA);
    stage_row(0, XA);
    if (DIST == 2) issue_row(Y0 - 2, XA);
    // every load of the prologue (weight fragments, bias / row-sum vectors) has landed: without this the compiler keeps conservative
    // s_waitcnt vmcnt(N) in front of their first uses INSIDE the loop, which in steady state wait for the previous row's stores
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();

    // ---- second 1x1 on a finished r row, SimpleGate2, store, channel sums.  Row yo = Y0 - 6 + jj was written to ring slot jj & 1 at the
    //      end of iteration jj; it is consumed at the TOP of iteration jj + 1, where its LDS -> MFMA -> exp / rcp -> store chain overlaps
    //      with the first 1x1 and the stencils of the next row instead of standing alone between the barrier and the loop end ----
    auto second_gemm = [&](int jj) {
        const int yo = Y0 - 6 + jj;
        if (yo < Y0) return;                                                  // workgroup-uniform; rows above the segment are warm-up
        const char* rs = lds_r[jj & 1];
        char* os = lds_o[jj & 1];
        uint4 W2[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int s = 0; s < 2; ++s) W2[m][s] = lds_w2[q][m][s][lane];
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            const uint4 b0 = *(const uint4*)(rs + (16 * n + p) * P1_PSR + (g * 8) * 2);
            const uint4 b1 = *(const uint4*)(rs + (16 * n + p) * P1_PSR + (32 + g * 8) * 2);
            f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            c0 = mfma16h(W2[0][0], b0, c0); c1 = mfma16h(W2[1][0], b0, c1);
            c0 = mfma16h(W2[0][1], b1, c0); c1 = mfma16h(W2[1][1], b1, c1);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = c0[r] * sigmoidf_(c1[r]);
            const int rc = 16 * n + p, gx = x0 - 3 + rc;
            const bool ok = rc >= 3 && rc < 3 + A.vw && gx < w;               // own columns of this strip
#pragma unroll
            for (int r = 0; r < 4; ++r) psum[r] += ok ? v[r] : 0.f;
            *(uint2*)(os + rc * P1_PSO + (16 * g + 4 * q) * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    };
    // ---- a finished g2 row leaves the CU one barrier after second_gemm wrote it to lds_o: thread = (pixel, 16-byte piece), i.e. whole
    //      128-byte pixels per 8 consecutive lanes (8-byte stores straight from the accumulator layout cost 170 us per launch) ----
    auto store_row = [&](int jj) {
        const int yo = Y0 - 6 + jj;
        if (yo < Y0) return;                                                  // workgroup-uniform
        const char* os = lds_o[jj & 1];
        bf16_t* const g2row = A.g2 + ((size_t)t * h + yo) * w * C;            // wave-uniform row base of the output
#pragma unroll
        for (int k = 0; k < (RWD * 8 + 255) / 256; ++k) {
            const int idx = tid + 256 * k, rc = idx >> 3, c8 = idx & 7, gx = x0 - 3 + rc;
            const bool ok = idx < RWD * 8 && rc >= 3 && rc < 3 + A.vw && gx < w;
            const uint4 v = *(const uint4*)(os + (ok ? rc : 0) * P1_PSO + c8 * 16);
            if (ok && (P1_ON(1024) || A.vw == 12345)) *(uint4*)(g2row + (gx * C + c8 * 8)) = v;
        }
    };

    // one row of the walk; Xs: registers holding row yin + 1 (loaded one iteration ago, staged at the end of this one), Xl: free set,
    // receives row yin + 2
    auto iteration = [&](const int j, uint4* Xs, uint4* Xl) {
        const int yin = Y0 - 3 + j;                                           // input row of this iteration (staged in slot j & 1)
        const bool rowin = yin >= 0 && yin < h;
        tick(7);
        // Input rows are fetched TWO iterations ahead (HBM round trips under load outlast one iteration: the wait before stage_row was 28 %
        // of the wave cycles with a distance of one), and BEFORE this iteration's stores in program order: vmcnt retires in order, a
        // load behind a store would wait for the store's acknowledgement.
        issue_row(yin + DIST, Xl);
        __builtin_amdgcn_sched_barrier(0);                                    // ... and they stay HERE: the scheduler otherwise sinks the loads to their use
        if (j > 1 && P1_ON(64)) store_row(j - 2);
        if (j > 0 && P1_ON(64)) second_gemm(j - 1);
        tick(0);
        // Weight records are fetched ONE GROUP AHEAD (group = one kernel row of one pass: 4 NX .. 6 NX packed FMAs) into two alternating
        // register sets, with scheduling fences between the groups: left alone, the scheduler hoists all 34 LDS reads of an iteration to
        // its top (86 more live registers, spills at two waves per SIMD).
        uint32_t wb[2][6];
        const uint32_t* w3p = &lds_w3[q][g][0][0][0];
        const uint32_t* w5p = &lds_w5[q][g][0][0][0];
        auto ld3 = [&](int kp, int ty, uint32_t* d) {
            const uint4 a4 = *(const uint4*)(w3p + (kp * 3 + ty) * 8); const uint2 b2 = *(const uint2*)(w3p + (kp * 3 + ty) * 8 + 4);
            d[0] = a4.x; d[1] = a4.y; d[2] = a4.z; d[3] = a4.w; d[4] = b2.x; d[5] = b2.y;
        };
        auto ld5 = [&](int k, int ty, uint32_t* d) {
            const uint4 a4 = *(const uint4*)(w5p + (k * 5 + ty) * 8);
            d[0] = a4.x; d[1] = a4.y; d[2] = a4.z; d[3] = a4.w; d[4] = w5p[(k * 5 + ty) * 8 + 4];
        };
        ld3(0, 2, wb[0]);
        // ---- first 1x1 on the RAW operands + LayerNorm epilogue -> a (packed fp16, zero outside the image) ----
        uint32_t ah[NX][4];
        {
            const char* xs = lds_x[j & 1];
            const float4 ws0 = lds_wb[q][g][0], ws1 = lds_wb[q][g][1], bs0 = lds_wb[q][g][2], bs1 = lds_wb[q][g][3];   // phase-local: 16 registers not held across the stencils
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const uint4 bqr = *(const uint4*)(xs + (16 * n + p) * PSX + (32 * s + 8 * g) * 2);
                    const bf16x8_t bq = as_frag(bqr);
                    if (P1_ON(2)) { acc0 = mfma16(W1[0][s], bq, acc0); acc1 = mfma16(W1[1][s], bq, acc1); }
                    else { acc0[0] += __uint_as_float(bqr.x); acc1[1] += __uint_as_float(bqr.w); }
                }
                const float2 st = lds_st[j & 1][16 * n + p];
                const float rstd = st.x, tm = st.y;
                const bool in = rowin && colin[n];
                const float a00 = fmaf(rstd, acc0[0], fmaf(tm, ws0.x, bs0.x)), a01 = fmaf(rstd, acc0[1], fmaf(tm, ws0.y, bs0.y));
                const float a02 = fmaf(rstd, acc0[2], fmaf(tm, ws0.z, bs0.z)), a03 = fmaf(rstd, acc0[3], fmaf(tm, ws0.w, bs0.w));
                const float a10 = fmaf(rstd, acc1[0], fmaf(tm, ws1.x, bs1.x)), a11 = fmaf(rstd, acc1[1], fmaf(tm, ws1.y, bs1.y));
                const float a12 = fmaf(rstd, acc1[2], fmaf(tm, ws1.z, bs1.z)), a13 = fmaf(rstd, acc1[3], fmaf(tm, ws1.w, bs1.w));
                const uint32_t msk = in ? 0xffffffffu : 0u;                   // AND, not a select of the expressions: no branch around the epilogue
                ah[n][0] = cvt_pk_h2(a00, a01) & msk; ah[n][1] = cvt_pk_h2(a02, a03) & msk;
                ah[n][2] = cvt_pk_h2(a10, a11) & msk; ah[n][3] = cvt_pk_h2(a12, a13) & msk;
                if (n & 1) P1_FENCE();                 // two N-tiles' operand reads in flight at a time, not all NX
            }
        }
        tick(1);
        // ---- depthwise 3x3 (+identity), scatter form: row yin completes output row yin-1; then SimpleGate.  Two passes: registers
        //      (k, k+2) = a channel pair and its gate partners, so a pass ends with a finished g1 register ----
        h2_t g1h[NX][2];
        {
            const bool rin = (yin - 1) >= 0 && (yin - 1) < h;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                uint32_t L[NX][2], R[NX][2];
#pragma unroll
                for (int n = 0; n < NX; ++n)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int k = kp + 2 * kk;
                        L[n][kk] = P1_ON(8) ? from_left<1>(n > 0 ? ah[n - 1][k] : 0u, ah[n][k], n > 0) : ah[n][k];
                        R[n][kk] = P1_ON(8) ? from_right<1>(n + 1 < NX ? ah[n + 1][k] : 0u, ah[n][k], n + 1 < NX) : ah[n][k];
                    }
                h2_t F[NX][2];
#pragma unroll
                for (int ti = 0; ti < 3; ++ti) {                              // ty = 2 first: it reads P1 before ty = 1 overwrites it (from P0), then ty = 0 reopens P0
                    const int ty = 2 - ti, cur = (kp * 3 + ti) & 1;
                    if (ti < 2) ld3(kp, ty - 1, wb[cur ^ 1]); else if (kp == 0) ld3(1, 2, wb[cur ^ 1]); else ld5(0, 4, wb[cur ^ 1]);
#pragma unroll
                    for (int tx = 0; tx < (P1_ON(4) ? 3 : 1); ++tx)
#pragma unroll
                        for (int n = 0; n < NX; ++n)
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) {
                                const int k = kp + 2 * kk;
                                const h2_t v = as_h2(tx == 0 ? L[n][kk] : (tx == 1 ? ah[n][k] : R[n][kk]));
                                const h2_t wk = as_h2(wb[cur][2 * tx + kk]);
                                // input row yin is row (y + ty - 1) of output row y = yin + 1 - ty: ty = 2 completes yin-1, 1 feeds yin, 0 opens yin+1
                                if (ty == 2) F[n][kk] = __builtin_elementwise_fma(v, wk, tx == 0 ? P1[n][k] : F[n][kk]);
                                else if (ty == 1) P1[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? P0[n][k] : P1[n][k]);
                                else P0[n][k] = tx == 0 ? v * wk : __builtin_elementwise_fma(v, wk, P0[n][k]);
                            }
                    P1_FENCE();
                }
#pragma unroll
                for (int n = 0; n < NX; ++n) {                                // SimpleGate; zero padding of the 5x5: g1 is zero outside the image
                    const h2_t m = F[n][0] * F[n][1];
                    g1h[n][kp] = as_h2(as_u(m) & ((rin && colin[n]) ? 0xffffffffu : 0u));
                }
            }
        }
        tick(2);
        // ---- depthwise 5x5 (3x3 and identity folded), scatter form: g1 row yin-1 completes r row yin-3 ----
        h2_t Rr[NX][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint32_t S[5][NX];
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                const uint32_t cur = as_u(g1h[n][k]);
                const uint32_t pv = n > 0 ? as_u(g1h[n - 1][k]) : 0u, nx = n + 1 < NX ? as_u(g1h[n + 1][k]) : 0u;
                S[2][n] = cur;
                if (P1_ON(32)) {
                    S[0][n] = from_left<2>(pv, cur, n > 0); S[1][n] = from_left<1>(pv, cur, n > 0);
                    S[3][n] = from_right<1>(nx, cur, n + 1 < NX); S[4][n] = from_right<2>(nx, cur, n + 1 < NX);
                } else { S[0][n] = cur; S[1][n] = pv; S[3][n] = nx; S[4][n] = cur; }
            }
#pragma unroll
            for (int ti = 0; ti < 5; ++ti) {                                  // ty = 4 first: it reads Q3 before ty = 3 overwrites it, and so on down to Q0
                const int ty = 4 - ti, cur = (6 + k * 5 + ti) & 1;
                if (ti < 4) ld5(k, ty - 1, wb[cur ^ 1]); else if (k == 0) ld5(1, 4, wb[cur ^ 1]);
#pragma unroll
                for (int tx = 0; tx < (P1_ON(16) ? 5 : 1); ++tx) {
                    const h2_t wk = as_h2(wb[cur][tx]);
#pragma unroll
                    for (int n = 0; n < NX; ++n) {
                        const h2_t v = as_h2(S[tx][n]);
                        // g1 row yi = yin-1 is row (y + ty - 2) of output row y = yi + 2 - ty
                        if (ty == 4) Rr[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q3[n][k] : Rr[n][k]);
                        else if (ty == 3) Q3[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q2[n][k] : Q3[n][k]);
                        else if (ty == 2) Q2[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q1[n][k] : Q2[n][k]);
                        else if (ty == 1) Q1[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q0[n][k] : Q1[n][k]);
                        else Q0[n][k] = tx == 0 ? v * wk : __builtin_elementwise_fma(v, wk, Q0[n][k]);
                    }
                }
                P1_FENCE();
            }
        }
        tick(3);
        // ---- hand-over: r row yo = yin - 3 -> ring slot j & 1; next input row + its statistics -> slot (j + 1) & 1; ONE barrier ----
        char* rs = lds_r[j & 1];
#pragma unroll
        for (int n = 0; n < NX; ++n)
            *(uint2*)(rs + (16 * n + p) * P1_PSR + (16 * g + 4 * q) * 2) = make_uint2(as_u(Rr[n][0]), as_u(Rr[n][1]));
        tick(4);
        stage_row((j + 1) & 1, Xs);
        tick(5);
        // Slot j & 1 of r and slot (j + 1) & 1 of x are complete after this barrier.  Both were last READ before the previous barrier
        // (r: the second 1x1 of row j - 2 runs at the top of iteration j - 1, x: the first 1x1 of iteration j - 1; both precede barrier j - 1).
        if (P1_ON(128)) __syncthreads();
        tick(6);
    };
#pragma unroll 1
    for (int j = 0; j < nit; j += 2) {
        if (DIST == 2) {
            iteration(j, XA, XB);
            if (j + 1 < nit) iteration(j + 1, XB, XA);
        } else {                                        // one set: loaded at the top, staged at the end of the same iteration
            iteration(j, XA, XA);
            if (j + 1 < nit) iteration(j + 1, XA, XA);
        }
    }
    if (P1_ON(64)) {                                                          // drain: the last two rows of the segment
        store_row(nit - 2);
        second_gemm(nit - 1);
        __syncthreads();
        store_row(nit - 1);
    }
    if (!P1_ON(2048)) {
        if (lane == 0)
            for (int k = 0; k < 8; ++k) A.pool[((size_t)t * (A.nsx * A.nsy) + sy * A.nsx + sx) * C + 16 * q + k] = (float)tacc[k];
        return;
    }
    // channel sums of this (frame, strip, segment) for CALayer2: wave q, lane group g own channels 16 g + 4 q + r
    if (A.pool) {
        const int nblk = A.nsx * A.nsy, blk = sy * A.nsx + sx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sm = row_sum16(psum[r]);
            if (p == 0) A.pool[((size_t)t * nblk + blk) * C + 16 * g + 4 * q + r] = sm;
        }
    }
}

// row segments per column strip: as few rounds of resident workgroups (2 per CU) as possible, each (segment + 6 warm-up rows) long
void p1_partition(int T, int h, int w, int ncu, int vwmax, int& nsx, int& vw, int& nsy, int& seg) {
    nsx = (w + vwmax - 1) / vwmax;
    vw = (w + nsx - 1) / nsx;                                                 // equal strips instead of a nearly empty last one
    long best = -1;
    nsy = 1;
    for (int cand = 1; cand <= (h + 7) / 8; ++cand) {
        const int sg = (h + cand - 1) / cand;
        if ((sg * (cand - 1)) >= h) continue;                                 // the last segment would be empty
        const long items = (long)T * nsx * cand, rounds = (items + 2L * ncu - 1) / (2L * ncu);
        const long cost = rounds * (sg + 6);
        if (best < 0 || cost < best) { best = cost; nsy = cand; }
    }
    seg = (h + nsy - 1) / nsy;
}

int p1_ncu() {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) return -1;
    return ncu;
}

constexpr int P1_NX = 3, P1_VWMAX = 16 * P1_NX - 6;      // NX = 4 needs ~300 registers per lane (spills at two waves per SIMD); 3: 254

}  // namespace

extern "C" {

int sn_cab_phase1_blocks(int T, int h, int w) {
    const int ncu = p1_ncu();
    if (ncu < 1 || T < 1 || h < 1 || w < 1) return SN_EINVAL;
    int nsx, vw, nsy, seg;
    p1_partition(T, h, w, ncu, P1_VWMAX, nsx, vw, nsy, seg);
    return nsx * nsy;
}

int sn_cab_phase1(const sn_unit_src* s, const void* hw, const void* wfrag1, const float* bias, const float* wsum, const uint32_t* w3,
                  const uint32_t* w5, const void* wfrag2, void* g2, float* pool, void* stream) {
    sn_clear_error();
    if (!s || !s->x || s->C != 64 || s->mode < 0 || s->mode > 2 || s->T < 1 || s->h < 1 || s->w < 1 || !wfrag1 || !bias || !wsum || !w3 || !w5 ||
        !wfrag2 || !g2 || (s->mode != 0 && !hw)) return SN_EINVAL;
    const int ncu = p1_ncu();
    if (ncu < 1) return SN_ELAUNCH;
    P1Args A;
    A.x = (const bf16_t*)s->x; A.hwb = (const bf16_t*)hw; A.T = s->T; A.h = s->h; A.w = s->w; A.mode = s->mode; A.wrap = s->wrap;
    A.wfrag1 = (const uint4*)wfrag1; A.bias = bias; A.wsum = wsum; A.w3 = w3; A.w5 = w5; A.wfrag2 = (const uint4*)wfrag2;
    A.g2 = (bf16_t*)g2; A.pool = pool;
    p1_partition(s->T, s->h, s->w, ncu, P1_VWMAX, A.nsx, A.vw, A.nsy, A.seg);
    const dim3 grid((unsigned)(s->T * A.nsx * A.nsy));
    sn_clear_error();
    if (s->mode) hipLaunchKernelGGL((cab_phase1_kernel<3, P1_NX>), grid, dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL((cab_phase1_kernel<2, P1_NX>), grid, dim3(256), 0, (hipStream_t)stream, A);
    return sn_check_launch();
}

}  // extern "C"
