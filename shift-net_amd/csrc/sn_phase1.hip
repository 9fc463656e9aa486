// Fused phase 1 of a CAB1 / CAB2 block of a GSTS unit (gfx950), depthwise variants (C = 64, deblur):
//     g2 = SimpleGate2(body[4](RepConv(SimpleGate(RepConv2(body[0](LayerNorm2d(u)))))))      (gshift_deblur2.py:186-258)
// in ONE kernel: u is read once, g2 is written once, and neither the 2C-channel tensor `a`, nor g1, nor r ever reaches HBM or even LDS
// (only r crosses LDS once, for the second 1x1).  Phase 2 (CALayer2 scale -> 1x1 -> beta-residual) is sn_gsts_cab2_phase2 / sn_cab1_phase2 (csrc/sn_gsts.hip); the global
// average pool of CALayer2 sits between the two and is the one unavoidable grid-wide dependency of the block.
//
// Idea.  With the weights as the MFMA A operand, the accumulators of a 1x1 conv are laid out D[channel 4g + r][pixel p]: a lane group g
// owns 4 channels, the 16 lanes of a DPP row are 16 CONSECUTIVE PIXELS of one image row.  A depthwise stencil on such a tensor needs
// only (i) the same registers of the rows above / below -- a sliding window over rows that a wave keeps in registers while it walks down
// a column strip -- and (ii) the neighbour pixels: with NX N-tiles whose columns INTERLEAVE (region column = NX p + n) these are the same
// lane's registers of the other tiles, plus one v_mov_b32_dpp per operand that crosses the lane boundary.  So the whole chain runs
// register to register:
//
//   wave q of a 4-wave team owns gate pair q of the first 1x1 (M-tiles 2q, 2q+1: a-channels 16g+4q+r and their partners C + ...),
//   hence g1 / r channels 16g+4q+r, and gate pair q of the second 1x1 (g2 channels 16g+4q+r); all four waves cover the SAME pixels.
//   Per input row y of a (16 NX)-pixel wide region (NX interleaved N-tiles; the outer 3 columns on each side are halo):
//     raw bf16 pixels: staged ONCE per team through LDS one row ahead (thread = pixel x 8-channel piece)  -> B fragments, NO unpack / normalise pass
//     LayerNorm statistics from the raw pieces while staging (v_dot2c with ones / with itself, quad reduction), once per team
//     1x1 on the RAW operands; LayerNorm applied AFTER it: a = rstd (W v - mu W 1 + sigma b)  (W 1 = row sums of the bf16 weights, host).
//     The two correction terms ride on ONE more k-step of the same MFMA chain: B rows (-mu, sigma) per pixel, A columns (W 1, b) per output
//     channel, each split into bf16 hi + lo parts (4 + 4 of the step's 32 k-slots; 2^-16 relative), so the epilogue is a = rstd * acc
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
    const bf16_t* x; const bf16_t* halo; const bf16_t* hwb;
    int T, h, w, mode, wrap, t0;
    const uint4* wfrag1; const uint4* wfragx;
    const uint32_t* w3; const uint32_t* w5;
    const uint4* wfrag2;
    bf16_t* g2; float* pool;
    int nsx, nsy, seg, vw;
    SeFold se;                           // se.ca == nullptr: partial channel sums only (sn_ca_mlp follows as its own launch)
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

// Lane <-> pixel.  The 16 lanes of a DPP row are the 16 N-columns of an MFMA tile, and WHICH 16 pixels of the region they stand for is a
// free choice (it only decides the LDS address a lane reads its B fragment from).  Region column of (N-tile n, lane p) = NX p + n:
// the NX tiles INTERLEAVE, so the horizontal neighbours of a pixel are the same lane's registers of the other tiles -- no instruction
// at all -- except across the lane boundary, where ONE v_mov_b32_dpp of a neighbour tile's register serves (row_shr:1 / row_shl:1,
// zero shifted in at the region's edge: those columns are halo, their results are never used).  With tiles of 16 consecutive pixels
// every shifted operand cost two DPP moves (shift + rotate of the adjacent tile): 96 of the ~570 VALU instructions per row.
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }   // lane p <- p - 1
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }   // lane p <- p + 1

// Two scheduling constants, overridable for A/B builds (tools/p1_variants.py).  One MI355X box, 20 x 360 x 640, us per launch CAB1 / CAB2,
// mean of 3:   DIST 1, WPF 4: 833 / 747     DIST 1, WPF 6: 744 / 736     DIST 2, WPF 4: 706 / 722     (WPF 1, DIST 2: 829 / 776 on another box).
// The kernel is latency-bound per wave (ONE workgroup per CU instead of two: only 1.17x - 1.29x slower); of a launch, ~640 us remain with
// every load served from one cache line, no global store and no row barrier, i.e. 75 - 90 % is the instruction stream itself.
#ifndef P1_DIST          // input rows in flight ahead of the one being staged
#define P1_DIST 2
#endif
#ifndef P1_WPF           // stencil weight records are fetched from LDS this many groups (kernel rows of a pass) ahead of their use
#define P1_WPF 4
#endif
#ifndef P1_LDS_PAD       // measurement builds only: extra LDS per workgroup (16384 leaves ONE workgroup per CU)
#define P1_LDS_PAD 0
#endif
constexpr int P1_PSO = 144;          // LDS bytes per pixel of a finished g2 row
constexpr int P1_PSR = 160;          // LDS bytes per pixel of an r row: 128 + 32 (10 slots of 16 B, 2 mod 4: conflict-free ds_read_b128 lane groups,
                                     // also with the odd lane stride NX of the interleaved columns)

template <int KS, int NX>
__global__ __launch_bounds__(256, 2) void cab_phase1_kernel(const P1Args A) {
    constexpr int C = 64, CH = 32, K = 32 * KS, MT = 8, RWD = 16 * NX;
    constexpr int PSX = KS == 2 ? 160 : 224;                                  // LDS bytes per pixel of a staged input row (10 / 14 slots: 2 mod 4)
    static_assert(KS == 2 || KS == 3, "K = C (CAB1) or C + C/2 (CAB2)");
    static_assert(NX >= 2 && (NX & 1), "5-tap rows reach two columns over the lane boundary; an odd lane stride keeps the LDS reads conflict free");
    static_assert(RWD * 4 <= 256, "staging: one thread per (region pixel, 8-channel piece of a k-step)");
    // stencil weights, packed fp16 pairs, one 32-byte record per (pass, kernel row): [wave][g][pass][row][8 words]
    //   3x3 pass kp: words 2 tx + kk = tap (row, tx) of a-register k = kp + 2 kk (kk = 0: channels (4g+.. r = 2kp, 2kp+1), kk = 1: their gate partners)
    //   5x5 pass k : words tx = tap (row, tx) of g1-register k (channels r = 2k, 2k+1)
    __shared__ __attribute__((aligned(16))) uint32_t lds_w3[4][4][2][3][8];
    __shared__ __attribute__((aligned(16))) uint32_t lds_w5[4][4][2][5][8];
    __shared__ __attribute__((aligned(16))) uint4 lds_w2[4][2][2][64];       // [wave][M-tile of the pair][k-step][lane]: second 1x1, fp16 fragments
    __shared__ __attribute__((aligned(16))) uint4 lds_wx[4][2][17];          // [wave][M-tile of the pair][row | zeros]: A fragment of the LayerNorm k-step
                                                                              // (row sums and bias of the row, bf16 hi / lo; only k-slots 0..7 = lanes 0..15 are non-zero)
    __shared__ __attribute__((aligned(16))) char lds_x[2][RWD * PSX];        // staged raw input rows (ring of 2)
    __shared__ float lds_st[2][RWD];                                         // rstd per pixel of the staged row, 0 outside the image (zero padding of the 3x3)
    __shared__ __attribute__((aligned(16))) char lds_r[2][RWD * P1_PSR];     // r rows (ring of 2)
    __shared__ __attribute__((aligned(16))) char lds_o[2][RWD * P1_PSO];     // finished g2 rows (ring of 2): every wave holds 4 of a pixel's 64 channels,
                                                                              // the rows leave as 16-byte pieces of whole 128-byte pixels
#if P1_LDS_PAD
    __shared__ char lds_pad[P1_LDS_PAD];
    if (A.T < 0) { lds_pad[threadIdx.x] = 1; __syncthreads(); A.pool[0] = lds_pad[threadIdx.x ^ 1]; }      // (never taken) keeps the array alive
#endif
    const int tid = threadIdx.x, lane = tid & 63, q = wave_id(), g = lane >> 4, p = lane & 15;
    const int b = blockIdx.x, sx = b % A.nsx, sy = (b / A.nsx) % A.nsy, t = A.t0 + b / (A.nsx * A.nsy);
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
    if (lane < 34) {
        const int m = lane / 17, i = lane - 17 * m;
        lds_wx[q][m][i] = i < 16 ? A.wfragx[(2 * q + m) * 16 + i] : make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- staging role: thread = (region pixel spx, piece c4): it moves the 8-channel pieces [32 s + 8 c4, +8) of the virtual input u
    //       of its pixel, s = 0..KS-1 -- exactly the B fragments of lane group g = c4 -- HBM -> registers -> LDS, one row
    //      AHEAD, and derives the LayerNorm statistics of the pixel (each input row is read and reduced once per team, not once per wave)
    const int spx = tid >> 2, c4 = tid & 3;
    const bool stager = tid < RWD * 4;                                        // (NX = 3: the last wave has no staging work)
    const SnSlabs<bf16_t> sl = sn_unit_slabs<bf16_t>(A.x, A.halo, A.T, hw, C, A.mode, A.wrap, t);
    // wave-uniform 64-bit frame bases (scalar registers) + 32-bit per-lane element offsets (a frame has < 2^31 elements)
    const bf16_t* slab[KS];
    int sstride[KS], soff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s == 0) { slab[s] = sl.p0; sstride[s] = sl.s0; }
        else if (s == 1) { slab[s] = sl.p1; sstride[s] = sl.s1; }
        else { slab[s] = A.hwb + (size_t)t * hw * CH; sstride[s] = CH; }
        soff[s] = 8 * c4;
    }
    const int sgx = x0 - 3 + spx, sgxc = (sgx >= 0 && sgx < w) ? sgx : 0;    // clamped column: loads are unconditional, masks come later
    // Input rows in flight: DIST = 2 rows ahead with two register sets that swap roles every iteration (loop unrolled by two).  (Spills
    // must be avoided at any price here: a spill reload issued behind the prefetch waits for it, vmcnt retires in order.)
    constexpr int DIST = P1_DIST;
    uint4 XA[KS], XB[DIST == 2 ? KS : 1];
    auto issue_row = [&](int y, uint4* X) {
        const int yc = (y >= 0 && y < h) ? y : 0;
        const int ii = stager ? yc * w + sgxc : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) X[s] = *(const uint4*)(slab[s] + (ii * sstride[s] + soff[s]));
    };
    auto stage_row = [&](int slot, const uint4* Xr, const int y) {            // registers -> LDS + statistics of the pixel
        if (!stager) return;                                                  // wave-uniform
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *(uint4*)(lds_x[slot] + spx * PSX + (32 * s + 8 * c4) * 2) = Xr[s];
            const uint32_t wd[4] = {Xr[s].x, Xr[s].y, Xr[s].z, Xr[s].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1 = dot2bf(wd[i], 0x3f803f80u, s1); s2 = dot2bf(wd[i], wd[i], s2); }
        }
        s1 += dpp_mov<0xB1>(s1); s1 += dpp_mov<0x4E>(s1);                     // sum over the 4 lanes of the pixel (quad_perm)
        s2 += dpp_mov<0xB1>(s2); s2 += dpp_mov<0x4E>(s2);
        const float mean = s1 * (1.0f / K);
        const float var = fmaxf(s2 * (1.0f / K) - mean * mean, 0.f);
        const float ve = var + 1e-6f, rstd = __builtin_amdgcn_rsqf(ve), sigma = ve * rstd;
        // B fragment of the LayerNorm k-step (lane group 0 reads it, the others the zero slot behind it): k-slots (-mu hi, -mu hi, -mu lo,
        // -mu lo, sigma hi, sigma hi, sigma lo, sigma lo) against the A columns (W1 hi, W1 lo, W1 hi, W1 lo, b hi, b lo, b hi, b lo)
        const uint32_t m0 = pack_bf2(-mean, -mean), s0 = pack_bf2(sigma, sigma);
        const float mlo = -mean - __uint_as_float(m0 << 16), slo = sigma - __uint_as_float(s0 << 16);
        const uint4 rec = make_uint4(m0, pack_bf2(mlo, mlo), s0, pack_bf2(slo, slo));
        if (c4 < 2) *(uint4*)(lds_x[slot] + spx * PSX + 2 * K + 16 * c4) = c4 == 0 ? rec : make_uint4(0u, 0u, 0u, 0u);
        if (c4 == 0) lds_st[slot][spx] = (y >= 0 && y < h && sgx >= 0 && sgx < w) ? rstd : 0.f;
    };

    // region column of lane p in N-tile n = NX p + n  <->  image column gx = x0 - 3 + NX p + n
    bool colin[NX];
    float own[NX];                                                            // 1: a column this strip owns (its g2 is stored and counted), 0: halo / beyond the image
#pragma unroll
    for (int n = 0; n < NX; ++n) {
        const int rc = NX * p + n, gx = x0 - 3 + rc;
        colin[n] = gx >= 0 && gx < w;
        own[n] = (rc >= 3 && rc < 3 + A.vw && gx < w) ? 1.f : 0.f;
    }

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
    const int nit = (Y1 - Y0) + 6;                                            // rows walked: the segment + 6 warm-up rows
    issue_row(Y0 - 3, XA);
    stage_row(0, XA, Y0 - 3);
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
            const int rc = NX * p + n;
            const uint4 b0 = *(const uint4*)(rs + rc * P1_PSR + (g * 8) * 2);
            const uint4 b1 = *(const uint4*)(rs + rc * P1_PSR + (32 + g * 8) * 2);
            f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            c0 = mfma16h(W2[0][0], b0, c0); c1 = mfma16h(W2[1][0], b0, c1);
            c0 = mfma16h(W2[0][1], b1, c0); c1 = mfma16h(W2[1][1], b1, c1);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                     // b1 * sigmoid(b2); the gate rows carry -log2(e) (prep.pack_phase1): one v_exp_f32, no multiply
                v[r] = c0[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(c1[r]));
                psum[r] = fmaf(v[r], own[n], psum[r]);
            }
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
            if (ok) *(uint4*)(g2row + (gx * C + c8 * 8)) = v;
        }
    };

    // one row of the walk; Xs: registers holding row yin + 1 (loaded one iteration ago, staged at the end of this one), Xl: free set,
    // receives row yin + 2
    auto iteration = [&](const int j, uint4* Xs, uint4* Xl) {
        const int yin = Y0 - 3 + j;                                           // input row of this iteration (staged in slot j & 1)
        // Input rows are fetched TWO iterations ahead (HBM round trips under load outlast one iteration: the wait before stage_row was 28 %
        // of the wave cycles with a distance of one), and BEFORE this iteration's stores in program order: vmcnt retires in order, a
        // load behind a store would wait for the store's acknowledgement.
        issue_row(yin + DIST, Xl);
        __builtin_amdgcn_sched_barrier(0);                                    // ... and they stay HERE: the scheduler otherwise sinks the loads to their use
        if (j > 1) store_row(j - 2);
        if (j > 0) second_gemm(j - 1);
        // Weight records are fetched P1_WPF GROUPS AHEAD (group = one kernel row of one pass: 5 NX .. 6 NX packed FMAs) into a ring of
        // register sets.  Groups of a row: 0..5 = 3x3 (pass kp = gi / 3, kernel row 2 - gi % 3), 6..15 = 5x5 (pass (gi - 6) / 5, row 4 - (gi - 6) % 5)
        constexpr int WR = P1_WPF + 1;
        uint32_t wb[WR][6];
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
        auto ldg = [&](int gi) {                                              // gi is a compile-time constant after unrolling
            if (gi < 6) ld3(gi / 3, 2 - gi % 3, wb[gi % WR]);
            else if (gi < 16) ld5((gi - 6) / 5, 4 - (gi - 6) % 5, wb[gi % WR]);
        };
#pragma unroll
        for (int gi = 0; gi < P1_WPF; ++gi) ldg(gi);
        // ---- first 1x1 on the RAW operands + LayerNorm epilogue -> a (packed fp16, zero outside the image) ----
        uint32_t ah[NX][4];
        {
            const char* xs = lds_x[j & 1];
            const int xl = lane < 16 ? lane : 16;
            const bf16x8_t Wx0 = as_frag(lds_wx[q][0][xl]), Wx1 = as_frag(lds_wx[q][1][xl]);
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                const int rc = NX * p + n;
                f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8_t bq = as_frag(*(const uint4*)(xs + rc * PSX + (32 * s + 8 * g) * 2));
                    acc0 = mfma16(W1[0][s], bq, acc0); acc1 = mfma16(W1[1][s], bq, acc1);
                }
                const bf16x8_t bx = as_frag(*(const uint4*)(xs + rc * PSX + 2 * K + (g ? 16 : 0)));
                acc0 = mfma16(Wx0, bx, acc0); acc1 = mfma16(Wx1, bx, acc1);
                const float rstd = lds_st[j & 1][rc];                         // 0 outside the image: a = 0 there
                ah[n][0] = cvt_pk_h2(rstd * acc0[0], rstd * acc0[1]); ah[n][1] = cvt_pk_h2(rstd * acc0[2], rstd * acc0[3]);
                ah[n][2] = cvt_pk_h2(rstd * acc1[0], rstd * acc1[1]); ah[n][3] = cvt_pk_h2(rstd * acc1[2], rstd * acc1[3]);
            }
        }
        // ---- depthwise 3x3 (+identity), scatter form: row yin completes output row yin-1; then SimpleGate.  Two passes: registers
        //      (k, k+2) = a channel pair and its gate partners, so a pass ends with a finished g1 register ----
        h2_t g1h[NX][2];
        {
            const bool rin = (yin - 1) >= 0 && (yin - 1) < h;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                uint32_t Lw[2], Rw[2];                                        // the two operands that cross the lane boundary
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) { Lw[kk] = lane_prev(ah[NX - 1][kp + 2 * kk]); Rw[kk] = lane_next(ah[0][kp + 2 * kk]); }
                h2_t F[NX][2];
#pragma unroll
                for (int ti = 0; ti < 3; ++ti) {                              // ty = 2 first: it reads P1 before ty = 1 overwrites it (from P0), then ty = 0 reopens P0
                    const int ty = 2 - ti, gi = kp * 3 + ti, cur = gi % WR;
                    ldg(gi + P1_WPF);
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int n = 0; n < NX; ++n)
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) {
                                const int k = kp + 2 * kk;
                                const uint32_t src = tx == 0 ? (n > 0 ? ah[n - 1][k] : Lw[kk]) : (tx == 1 ? ah[n][k] : (n + 1 < NX ? ah[n + 1][k] : Rw[kk]));
                                const h2_t v = as_h2(src);
                                const h2_t wk = as_h2(wb[cur][2 * tx + kk]);
                                // input row yin is row (y + ty - 1) of output row y = yin + 1 - ty: ty = 2 completes yin-1, 1 feeds yin, 0 opens yin+1
                                if (ty == 2) F[n][kk] = __builtin_elementwise_fma(v, wk, tx == 0 ? P1[n][k] : F[n][kk]);
                                else if (ty == 1) P1[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? P0[n][k] : P1[n][k]);
                                else P0[n][k] = tx == 0 ? v * wk : __builtin_elementwise_fma(v, wk, P0[n][k]);
                            }
                }
#pragma unroll
                for (int n = 0; n < NX; ++n) {                                // SimpleGate; zero padding of the 5x5: g1 is zero outside the image
                    const h2_t m = F[n][0] * F[n][1];
                    g1h[n][kp] = as_h2(as_u(m) & ((rin && colin[n]) ? 0xffffffffu : 0u));
                }
            }
        }
        // ---- depthwise 5x5 (3x3 and identity folded), scatter form: g1 row yin-1 completes r row yin-3 ----
        h2_t Rr[NX][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // operand tx of tile n = region column NX p + n + tx - 2 = tile (n + tx - 2) mod NX of lane p - 1, p or p + 1
            uint32_t S[NX + 4];                                               // S[i + 2] = tile i for i = -2 .. NX + 1
#pragma unroll
            for (int n = 0; n < NX; ++n) S[n + 2] = as_u(g1h[n][k]);
            S[0] = lane_prev(S[NX]); S[1] = lane_prev(S[NX + 1]);            // tiles NX - 2, NX - 1 of the lane to the left
            const uint32_t n0 = lane_next(S[2]), n1 = lane_next(S[3]);        // tiles 0, 1 of the lane to the right
            S[NX + 2] = n0; S[NX + 3] = n1;
#pragma unroll
            for (int ti = 0; ti < 5; ++ti) {                                  // ty = 4 first: it reads Q3 before ty = 3 overwrites it, and so on down to Q0
                const int ty = 4 - ti, gi = 6 + k * 5 + ti, cur = gi % WR;
                ldg(gi + P1_WPF);
#pragma unroll
                for (int tx = 0; tx < 5; ++tx) {
                    const h2_t wk = as_h2(wb[cur][tx]);
#pragma unroll
                    for (int n = 0; n < NX; ++n) {
                        const h2_t v = as_h2(S[n + tx]);
                        // g1 row yi = yin-1 is row (y + ty - 2) of output row y = yi + 2 - ty
                        if (ty == 4) Rr[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q3[n][k] : Rr[n][k]);
                        else if (ty == 3) Q3[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q2[n][k] : Q3[n][k]);
                        else if (ty == 2) Q2[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q1[n][k] : Q2[n][k]);
                        else if (ty == 1) Q1[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? Q0[n][k] : Q1[n][k]);
                        else Q0[n][k] = tx == 0 ? v * wk : __builtin_elementwise_fma(v, wk, Q0[n][k]);
                    }
                }
            }
        }
        // ---- hand-over: r row yo = yin - 3 -> ring slot j & 1; next input row + its statistics -> slot (j + 1) & 1; ONE barrier ----
        char* rs = lds_r[j & 1];
#pragma unroll
        for (int n = 0; n < NX; ++n)
            *(uint2*)(rs + (NX * p + n) * P1_PSR + (16 * g + 4 * q) * 2) = make_uint2(as_u(Rr[n][0]), as_u(Rr[n][1]));
        stage_row((j + 1) & 1, Xs, yin + 1);
        // Slot j & 1 of r and slot (j + 1) & 1 of x are complete after this barrier.  Both were last READ before the previous barrier
        // (r: the second 1x1 of row j - 2 runs at the top of iteration j - 1, x: the first 1x1 of iteration j - 1; both precede barrier j - 1).
        __syncthreads();
    };
    // Two rows per trip: with DIST = 2 the register sets swap roles.  (The form matters to the register allocator: with the second half
    // unconditional, or with loop exits instead of the skip, the same code spills 8 - 30 registers.  Its price with DIST = 2: a static path
    // from the first half to the loop header on which that half's loads are still in flight, so the header carries an s_waitcnt vmcnt(0).)
#pragma unroll 1
    for (int j = 0; j < nit; j += 2) {
        iteration(j, XA, DIST == 2 ? XB : XA);
        if (j + 1 < nit) iteration(j + 1, DIST == 2 ? XB : XA, XA);
    }
    store_row(nit - 2);                                                       // drain: the last two rows of the segment
    second_gemm(nit - 1);
    __syncthreads();
    store_row(nit - 1);
    // channel sums of this (frame, strip, segment) for CALayer2: wave q, lane group g own channels 16 g + 4 q + r
    if (A.pool) {
        const int nblk = A.nsx * A.nsy, blk = sy * A.nsx + sx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sm = row_sum16(psum[r]);
            if (p == 0) sn_pool_store(&A.pool[((size_t)t * nblk + blk) * C + 16 * g + 4 * q + r], sm);
        }
        // the last workgroup of the frame finishes CALayer2 (lds_o is free: its last reader, store_row, is behind the tail's first barrier)
        if (A.se.ca) sn_se_tail(A.se, A.pool + (size_t)t * nblk * C, nblk, C, t, (float*)&lds_o[0][0], tid, 256);
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

// layout 1 of sn_phase1_weights: the role-split kernel (csrc/sn_phase1r.hip)
int sn_p1r_pool_blocks(int T, int h, int w);
int sn_p1r_launch(const sn_unit_src* s, const void* hw, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se,
                  const sn_phase1_opts* opt, void* stream);

extern "C" {

int sn_phase1_pool_blocks(int T, int h, int w, int layout) {
    if (layout == 1) return sn_p1r_pool_blocks(T, h, w);
    if (layout != 0) return SN_EINVAL;
    const int ncu = p1_ncu();
    if (ncu < 1 || T < 1 || h < 1 || w < 1) return SN_EINVAL;
    int nsx, vw, nsy, seg;
    p1_partition(T, h, w, ncu, P1_VWMAX, nsx, vw, nsy, seg);
    return nsx * nsy;
}

static int cab_phase1(const sn_unit_src* s, const void* hw, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se,
                      const sn_phase1_opts* opt, void* stream) {
    sn_clear_error();
    if (wt && wt->layout == 1) return sn_p1r_launch(s, hw, wt, g2, pool, se, opt, stream);
    if (opt && (opt->g1_scale || opt->g1_sums)) return SN_EINVAL;          // the VALU kernel has no inner CALayer2
    if (!s || !s->x || s->C != 64 || (wt && wt->layout != 0) || s->mode < 0 || s->mode > 2 || s->T < 1 || s->h < 1 || s->w < 1 || !wt || !wt->wfrag1 || !wt->wfragx || !wt->w3 ||
        !wt->w5 || !wt->wfrag2 || !g2 || (s->mode != 0 && !hw) || s->wrap < 0 || s->wrap > 2 || (s->wrap == 2 && s->mode != 0 && !s->halo)) return SN_EINVAL;
    const int ncu = p1_ncu();
    if (ncu < 1) return SN_ELAUNCH;
    P1Args A;
    A.x = (const bf16_t*)s->x; A.halo = (const bf16_t*)s->halo; A.hwb = (const bf16_t*)hw; A.T = s->T; A.h = s->h; A.w = s->w; A.mode = s->mode; A.wrap = s->wrap;
    A.wfrag1 = (const uint4*)wt->wfrag1; A.wfragx = (const uint4*)wt->wfragx; A.w3 = wt->w3; A.w5 = wt->w5; A.wfrag2 = (const uint4*)wt->wfrag2;
    A.g2 = (bf16_t*)g2; A.pool = pool;
    A.se.ca = nullptr;
    if (se) {
        if (!pool || !se->wa || !se->wb || !se->ticket || !se->ca || se->c != 64 || se->cr < 1 || se->cr > 128) return SN_EINVAL;
        A.se.wa = se->wa; A.se.wb = se->wb; A.se.ca = se->ca; A.se.ticket = se->ticket; A.se.inv_hw = 1.0f / ((float)s->h * (float)s->w);
        A.se.c = se->c; A.se.cr = se->cr;
    }
    p1_partition(s->T, s->h, s->w, ncu, P1_VWMAX, A.nsx, A.vw, A.nsy, A.seg);       // from the WHOLE unit: it fixes the pool layout
    SN_FRAME_RANGE(s, t0, nt);
    A.t0 = t0;
    const dim3 grid((unsigned)(nt * A.nsx * A.nsy));
    sn_clear_error();
    if (s->mode) hipLaunchKernelGGL((cab_phase1_kernel<3, P1_NX>), grid, dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL((cab_phase1_kernel<2, P1_NX>), grid, dim3(256), 0, (hipStream_t)stream, A);
    return sn_check_launch();
}

int sn_gsts_cab2_phase1(const sn_unit_src* s, const void* hw, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se,
                        const sn_phase1_opts* opt, void* stream) {
    if (!s || (s->mode != 1 && s->mode != 2)) return SN_EINVAL;
    return cab_phase1(s, hw, wt, g2, pool, se, opt, stream);
}

int sn_cab1_phase1(const sn_unit_src* s, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se, const sn_phase1_opts* opt, void* stream) {
    if (!s || s->mode != 0) return SN_EINVAL;
    return cab_phase1(s, nullptr, wt, g2, pool, se, opt, stream);
}

}  // extern "C"
