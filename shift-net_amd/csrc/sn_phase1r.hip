// Role-split fused phase 1 of a CAB1 / CAB2 block of a GSTS unit (gfx950), C = 64 and C = 80, depthwise or grouped RepConv:
//     g2 = SimpleGate2(body[4](RepConv(SimpleGate(RepConv2(body[0](LayerNorm2d(u)))))))      (gshift_deblur1.py:183-255, gshift_deblur2.py:186-258)
// in ONE kernel, with the RepConv (grouped 8 -> 8 per group in the "+" models, gshift_deblur1.py:157-165; depthwise in Shift-Net-s) on the
// MATRIX CORES.  u is read once, g2 written once; `a`, g1 and r never leave the CU.
//
// One workgroup per CU walks a 64-pixel wide column strip (58 own columns + 3 halo columns per side) top to bottom, one image row per
// step, ONE workgroup barrier per step.  Its waves have three roles that form a pipeline through LDS rings:
//
//   S (2 waves)   stagers.  Global loads of the virtual input u (two half-channel slabs of x, the shift-conv output hw),
//                 two rows ahead; TWO-PASS LayerNorm statistics in registers (mean, then sum (x - mean)^2: no E[x^2] - mean^2 cancellation);
//                 the NORMALISED operand (bf16) goes to the x ring together with a constant 1 in two spare k-slots, whose weight columns
//                 carry the folded LayerNorm bias (bf16 hi + lo) -- out-of-image pixels are all-zero operands, so `a` is exactly 0 there,
//                 the zero padding of the 3x3.  They also move finished g2 rows from the out ring to HBM as whole pixels (16-byte stores).
//                 Only these waves ever wait for HBM.
//   A (C/16 waves) wave q owns a-channels 16q .. 16q+15 and their gate partners: first 1x1 (bf16 MFMA, weights resident in registers) ->
//                 packed fp16 -> depthwise 3x3 (+identity) in scatter form on the accumulator layout (v_pk_fma_f16; rows above / below are the
//                 wave's own registers, horizontal neighbours are the same lane's registers of the other N-tiles because the four tiles
//                 INTERLEAVE: region column = 4 p + n) -> SimpleGate -> g1 row (fp16, 16 channels = two whole RepConv groups) -> g1 ring.
//   B (C/16 waves) wave q owns RepConv groups 2q, 2q+1 and gate pair q of the second 1x1.  RepConv as an x-PAIR TOEPLITZ GEMM: one MFMA row
//                 is (output channel oc of the group, pixel xp of a PAIR of neighbouring pixels), one column a pixel pair, k = (kernel row dy,
//                 input column dx6 in 0..5 relative to the pair, input channel): K = 5 x 6 x 8 = 240 in 8 k-steps, 5/6 of the matrix is real
//                 work -- the block-diagonal form of sn_grp5_gemm_gate (two groups per M-tile, half of every fragment zero) needs 13 k-steps
//                 per 16 pixels, this one 8 per 32.  A depthwise RepConv runs through the same code with diagonal 8 x 8 blocks (the matrix
//                 cores are otherwise idle; the packed-fp16 VALU form costs 4x the issue slots).  r (fp16) -> r ring; one step later
//                 every B wave reads all C channels of the row back as the B operand of the second 1x1 (fp16 MFMA), SimpleGate2, channel
//                 sums for CALayer2, g2 row -> out ring.
//
// Step j of a segment [Y0, Y1):   S stages input row Y0-2+j and stores g2 row Y0-10+j;  A turns the `a` row of step j-1 into g1 row Y0-5+j, then
// input row Y0-3+j into the next `a` row;  B computes r row Y0-8+j from g1 rows Y0-10+j .. Y0-6+j and g2 row Y0-9+j from the r row of the step
// before.  seg + 10 steps per segment.
//
// LDS (C = 80, CAB2: 149 KB, one workgroup per CU): x ring 2 rows, g1 ring 6 rows (5 being read + 1 being written), r ring 2, out ring 2.
// Every ring is laid out [N-tile n][lane p] so that the 16 lanes of a ds_read_b128 lane group hit 16 distinct 16-byte slots; the g1 ring is
// [ring row][wave][group][column mod 4][column / 4] with a row pitch that is a multiple of 256 bytes: the four lane groups of a RepConv
// B fragment read the same columns of four different ring rows (prep.p1r_tap), conflict free.
//
// Numerics: `a` and the 3x3 in fp16 as in sn_ln_gemm_gate; g1 and r in fp16 with the factor 2^-4 carried by g1 (exact power of two, undone
// in the second 1x1's weights), RepConv and the second 1x1 accumulate in fp32 on the matrix cores (the VALU kernel accumulated 25 taps in fp16).
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// Pool rows (partial channel sums of g2 / g1) are per (frame, column strip, block of P1R_RB image rows) -- a property of the image, not of the
// launch: whichever workgroups walk a frame, in whatever chunks, frame ranges and team sizes, every pool row is the same sum in the same order,
// and so is the squeeze-excite tail's reduction over them.  A temporally split window or a frame-range launch is bit-identical to the whole one.
#define P1R_RB 8

// Work decomposition (p1r_plan, host).  The (strip, frame) walks of one launch form ONE list of rows, cut into equal chunks -- one chunk per
// TEAM of F workgroups that walk F consecutive frames of the same strip rows in lock step (the half-channel roll of a CAB2 reads 64- / 80-byte
// halves of 128-byte lines whose other half belongs to the neighbouring frame: team members sit on one XCD and touch such a line at the same
// time).  Chunks are whole row blocks (P1R_RB rows).  A chunk that crosses the end of a strip simply continues with the next (strip, frame
// block) after a new warm-up.  Round 4 gave every workgroup whole (strip, segment) items: 20 frames x 12 strips = 240 walks of 370 steps on
// 256 CUs, 16 of them idle and no way to use them; the row list of 20 x 11 strips in 64 chunks of 312 rows takes ~330 steps.
struct P1RPlan {
    int nsx, sd, sr;      // column strips: count and how the slack of their capacity is spread (p1r_strip_begin)
    int F, nfb;           // frames walked in lock step by a team; frame blocks = ceil(nfr / F)
    int q, nteam;         // row BLOCKS of the list per team; teams with work
};
// first own column of strip s (s == nsx: w).  Strip 0 starts its 64-pixel region AT the image edge (no halo columns to the left of column 0)
// and so does the last one on the right: capacities 61, 58, ..., 58, 61 own columns (one strip: 64); 1280 / 2 = 640 columns are 11 strips, not 12.
__host__ __device__ inline int p1r_strip_begin(int nsx, int sd, int sr, int s, int w) {
    if (s <= 0) return 0;
    if (s >= nsx) return w;
    return 61 + 58 * (s - 1) - sd * s - (s < sr ? s : sr);
}

struct P1RArgs {
    const bf16_t* x; const bf16_t* halo; const bf16_t* hwb;
    int T, h, w, mode, wrap, t0, nfr;
    const uint4* wfrag1; const uint4* w3; const uint4* wgrp; const uint4* wfrag2;
    bf16_t* g2; float* pool;
    P1RPlan P;
    SeFold se;
    const float* g1_scale;               // denoisers: [T][C] scale of the inner CALayer2 (gshift_denoise1.py:224,257), applied to g1 by the A waves; or NULL
    int g1_sums;                         // 1: only the channel sums of g1 (the A waves' SimpleGate output) are produced: pool / se describe g1, no g2
    char* g1_store;                      // denoisers, or NULL: the sums pass also writes every g1 row it computes ([frame][strip][image row][P1R g1 row], fp16, unscaled),
                                         // and the second pass (ICA 3) reads them back instead of running the stagers' LayerNorm and the A waves again
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
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }   // lane p <- p - 1 (row_shr:1), 0 at p = 0
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }   // lane p <- p + 1 (row_shl:1), 0 at p = 15

#ifndef P1R_LDS_PAD      // measurement builds only
#define P1R_LDS_PAD 0
#endif
// how many LDS fragment reads the MFMA loops keep in flight ahead of their consumers (A: first 1x1, B: RepConv, B2: second 1x1)
// measurement builds only (tools/p1r_variants.py): switch off parts of the roles (results are wrong) to see which one paces the step
//   1: stagers skip the LayerNorm / LDS staging   2: stagers skip the global loads   4: stagers skip the g2 stores
//   8: A waves idle   16: B waves skip the second 1x1   32: B waves skip the RepConv   64: SimpleGate2 without exp / rcp
#ifndef P1R_SKIP
#define P1R_SKIP 0
#endif
#ifndef P1R_DA
#define P1R_DA 3
#endif
// Issue priority of the three roles (s_setprio, 0..3).  The SIMD's arbiter prefers the OLDEST wave, i.e. the A and B waves of a workgroup; the
// stagers, launched last and with the longest instruction stream, would run after them and alone (measured with s_memtime: A done after 1850,
// B after 2700 - 3000, S after 4600 of a step's 4800 cycles).  With the longest stream on top, the others fill its stalls: 2.34 / 2.30 -> 2.04 / 2.25 ms
// (C = 80, CAB1 / CAB2, 52 x 360 x 640), 0.69 / 0.75 -> 0.56 / 0.70 ms (C = 64, 20 x 360 x 640); S2 B1 A0 measured against 300, 311, 312, 231.
#ifndef P1R_PRIO_S
#define P1R_PRIO_S 2
#endif
#ifndef P1R_PRIO_B
#define P1R_PRIO_B 1
#endif
#ifndef P1R_PRIO_A
#define P1R_PRIO_A 0
#endif
#ifndef P1R_AV           // VALU instructions (the 3x3 of the first M-tile pass) scheduled behind every MFMA of the second pass of an A wave (0: the compiler's order)
#define P1R_AV 0
#endif
#ifndef P1R_BV           // VALU instructions (SimpleGate2 of the second 1x1) scheduled behind every RepConv MFMA of a B wave (0: the compiler's order)
#define P1R_BV 0
#endif
#ifndef P1R_DB
#define P1R_DB 6
#endif
#ifndef P1R_DB2
#define P1R_DB2 6
#endif

#ifndef P1R_NSW64        // stager waves at C = 64 (measurement builds: 2 = the round-4 split)
#define P1R_NSW64 4
#endif

// compile-time geometry shared by the kernel and the launcher
template <int C, bool HW> struct P1RShape {
    // Stager waves.  C = 80: 5 A + 5 B + 2 S = three waves on every SIMD (the 168-register limit).  C = 64 had 4 A + 4 B + 2 S; measured
    // alone (tools/p1_ab.py, 20 x 360 x 640, CAB1) the roles take A 251, B 307, S 368 us of the launch's 555: ONE stager wave's instruction
    // stream (two-pass LayerNorm of 4 - 6 sixteen-byte pieces per lane, then the g2 stores) is the longest dependency chain of a step.  Four
    // stagers with half the pieces each: 525 / 580 us (CAB1 / CAB2) against 555 / 621 with two, interleaved A/B on one device.
    static constexpr int NGP = C / 16, NSW = NGP == 4 ? P1R_NSW64 : 2, NW = 2 * NGP + NSW, NTHR = 64 * NW;
    static constexpr int CH = C / 2, K = HW ? C + CH : C, KS1 = (K + 2 + 31) / 32, KS2 = (C + 31) / 32;
    static constexpr int NX = 4, RWD = 16 * NX, HALO = 3, VWMAX = RWD - HALO;      // a border strip has no halo on its image side
    static constexpr int PSX = KS1 * 64 + 32;                 // bytes per pixel of a staged row: 4 KS1 + 2 slots of 16 B (2 mod 4)
    static constexpr int XPL = 16 * PSX + 32;                 // bytes per N-tile plane of a staged row: + 32 so that the stagers' ds_write_b128 lane groups (4 pixels
                                                              // = 4 planes x 2 interleaved pieces) cover all 32 banks; the readers' plane term is an immediate
    static constexpr int XSLOT = NX * XPL;
    static constexpr int GPL = 18 * 16;                       // bytes per g1 plane: 16 columns + one pad column on each side, 16 B each
    static constexpr int GROW = NGP * 2 * 4 * GPL;            // bytes per g1 ring row: [wave][group][column mod 4][18]; 11520 / 9216: multiples of 256
    static constexpr int GRING = 6;
    static constexpr int PSR = 160;                           // bytes per pixel of an r row (10 slots: 2 mod 4), also for C = 64
    static constexpr int RSLOT = RWD * PSR + 64;              // + pad: the last k-step of the second 1x1 reads past a pixel's C channels (zero weights)
    static constexpr int PSO = C * 2 + 16;
    static constexpr int OSLOT = RWD * PSO;
    static constexpr int OFF_X = 0, OFF_G = OFF_X + 2 * XSLOT, OFF_R = OFF_G + GRING * GROW, OFF_O = OFF_R + 2 * RSLOT;
    static constexpr int LDS = OFF_O + 2 * OSLOT + P1R_LDS_PAD;
    static constexpr int WARM = 10;                           // steps per walk beyond its rows
    static_assert(GROW % 256 == 0, "the four lane groups of a RepConv B fragment read four ring rows: the pitch must keep their bank phase");
    static_assert(LDS <= 160 * 1024, "LDS");
    static_assert((16 + (NTHR / (C / 4)) * C + 256) * 4 <= 2 * OSLOT, "sn_se_tail scratch lives in the out ring");
    static_assert(NSW == 2 || NSW == 4, "two or four lanes per region pixel");
};

typedef const P1RArgs __attribute__((address_space(4))) * P1RArgsP;
__device__ __forceinline__ P1RArgsP p1r_args() {
    P1RArgsP p = (P1RArgsP)__builtin_amdgcn_kernarg_segment_ptr();          // the kernel's only explicit argument sits at offset 0
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ P1RArgsP p1r_fresh(P1RArgsP p) { asm volatile("" : "+s"(p)); return p; }
// lane index from the execution mask: costs two VALU instructions where it is used instead of a register that lives across the step loops
__device__ __forceinline__ int p1r_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// one walk of a workgroup: rows [Y0, Y1) of strip s of frame t; all wave-uniform
struct P1RItem {
    int s, t, Y0, Y1;
    int xo, olo, ohi;     // image column of region column 0; own columns [olo, ohi) of the region
};

// ICA: the denoisers' inner CALayer2 on g1 (sn_phase1_opts).  0: none (deblur models); 1: sums pass (stagers + A waves only, channel sums of g1);
// 2: g1 times A.g1_scale before the RepConv.  A template parameter: the deblur kernels sit at the 168-register limit of three waves per SIMD.
// threads of a workgroup: the sums pass of the denoisers (ICA 1) has no B waves -- launching them idle cost the A waves a third of the register file
// 3: second pass from stored g1 rows (P1RArgs::g1_store): B waves + stagers that load a g1 row per step, scale it and put it into the ring -- no A waves.
template <int C, bool HW, int ICA> constexpr int p1r_threads() { return (ICA == 1 || ICA == 3) ? 64 * (P1RShape<C, HW>::NGP + P1RShape<C, HW>::NSW) : P1RShape<C, HW>::NTHR; }
// bytes of a stored g1 row of one strip: [wave][group][column mod 4][16 lanes] x 16 B -- the ring row without its pad columns
template <int C> constexpr int p1r_g1_row_bytes() { return (C / 16) * 8 * 256; }

template <int C, bool HW, int ICA>
__global__ __launch_bounds__((p1r_threads<C, HW, ICA>())) void cab_phase1r_kernel(const P1RArgs A_) {
    using SH = P1RShape<C, HW>;
    // The arguments are read from the kernarg segment where they are needed, through a pointer the compiler cannot see through (p1r_args): as SSA
    // values of the by-value parameter everything the end of a walk needs (plan, pool, squeeze-excite operands) stayed live across the step
    // loops -- 110 - 145 spilled SGPRs and, through their spill lanes, 7 - 30 spilled VGPRs in kernels that sit at the 168-register limit.
    P1RArgsP Ap = p1r_args();
#define A (*Ap)
    constexpr int NGP = SH::NGP, NTHR = p1r_threads<C, HW, ICA>(), CH = SH::CH, K = SH::K, KS1 = SH::KS1, KS2 = SH::KS2, NX = SH::NX;
    constexpr int PSX = SH::PSX, XPL = SH::XPL, XSLOT = SH::XSLOT, GPL = SH::GPL, GROW = SH::GROW, PSR = SH::PSR, RSLOT = SH::RSLOT, PSO = SH::PSO, OSLOT = SH::OSLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_x = smem + SH::OFF_X;
    char* const lds_g = smem + SH::OFF_G;
    char* const lds_r = smem + SH::OFF_R;
    char* const lds_o = smem + SH::OFF_O;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    // workgroup b runs on XCD b % 8: XCD k takes the k-th contiguous eighth of the logical workgroup list, so the F members of a team (and the
    // teams of neighbouring row chunks) share one L2
    const int per = (int)(gridDim.x >> 3), L = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const int team = L / A.P.F, fm = L - team * A.P.F;
    if (team >= A.P.nteam) return;                                            // workgroup-uniform (padding of the last eighth)
    const int h = A.h, w = A.w, hw = h * w;
    const int nbh = (h + P1R_RB - 1) / P1R_RB;                                // row blocks of a strip
    const int blocks_all = A.P.nsx * A.P.nfb * nbh;
    const int r0 = team * A.P.q, r1 = r0 + A.P.q < blocks_all ? r0 + A.P.q : blocks_all;
    const int u0 = r0 / nbh, u1 = (r1 + nbh - 1) / nbh;                       // the (strip, frame block) walks this team's chunk touches
    const int nrows = A.P.nsx * nbh;                                          // pool rows per frame: [strip][row block]
    auto chunks = [&](const int u) -> int { return ((u + 1) * nbh - 1) / A.P.q - (u * nbh) / A.P.q + 1; };
    auto item = [&](const int u, P1RItem& I) -> bool {                        // false: this member's frame of the block does not exist (ragged last block)
        const int s = u / A.P.nfb, fb = u - s * A.P.nfb, f = fb * A.P.F + fm;
        if (f >= A.nfr) return false;
        I.s = s; I.t = A.t0 + f;
        const int b0_ = r0 > u * nbh ? r0 - u * nbh : 0, b1_ = r1 - u * nbh < nbh ? r1 - u * nbh : nbh;
        I.Y0 = b0_ * P1R_RB;
        I.Y1 = b1_ * P1R_RB < h ? b1_ * P1R_RB : h;
        const int b0 = p1r_strip_begin(A.P.nsx, A.P.sd, A.P.sr, s, w), b1 = p1r_strip_begin(A.P.nsx, A.P.sd, A.P.sr, s + 1, w);
        I.xo = s == 0 ? 0 : b0 - SH::HALO;
        I.olo = b0 - I.xo; I.ohi = b1 - I.xo;
        return true;
    };
    // end of a walk, ALL threads: the last workgroup of the frame finishes CALayer2 (the out ring is free: its last reader is behind the final
    // barrier of the walk)
    auto finish = [&](const int u) {
        if (!A.pool) return;
        P1RItem I;
        item(u, I);
        const int tid = wv * 64 + p1r_lane();                                 // (not the kernel's `tid`: that one would stay live across the walk)
        if (A.se.ca) {
            const int fb = (I.t - A.t0) / A.P.F;
            int narr = 0;                                                     // walks that contribute to this frame
            for (int s = 0; s < A.P.nsx; ++s) narr += chunks(s * A.P.nfb + fb);
            SeFold se;
            se.wa = A.se.wa; se.wb = A.se.wb; se.ca = A.se.ca; se.ticket = A.se.ticket; se.bad = A.se.bad; se.inv_hw = A.se.inv_hw; se.c = A.se.c; se.cr = A.se.cr;
            sn_se_tail(se, A.pool + (size_t)I.t * nrows * C, nrows, narr, C, I.t, (float*)lds_o, tid, NTHR);
        }
    };

    // role of this wave.  A workgroup's waves go to the four SIMDs round-robin, so waves wv, wv + 4, wv + 8 share a SIMD: the roles are laid
    // out so that every SIMD gets one A wave (VALU-heavy), one B wave (MFMA-heavy) and one of {A4, B4, S0, S1} (C = 80) / one stager (C = 64)
    int role, q;                                                              // 0: A, 1: B, 2: S
    if (ICA == 1) { role = wv < NGP ? 0 : 2; q = wv < NGP ? wv : wv - NGP; }
    else if (ICA == 3) { role = wv < NGP ? 1 : 2; q = wv < NGP ? wv : wv - NGP; }
    else if (wv < 4) { role = 0; q = wv; }
    else if (wv < 8) { role = 1; q = wv - 4; }
    else if (wv - 8 < 2 * (NGP - 4)) { role = (wv - 8) & 1; q = 4 + ((wv - 8) >> 1); }
    else { role = 2; q = wv - 8 - 2 * (NGP - 4); }

    // ---- zero all LDS once: ring pads, unused k-slots and the rows B reads before A has produced them must be finite ----
    for (int e = tid; e < SH::LDS / 16; e += NTHR) ((uint4*)smem)[e] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    if (role == 2 && ICA == 3) {
        // ============================== S, second pass of the denoisers from stored g1 rows: load, scale, ring; g2 rows out ==============================
        __builtin_amdgcn_s_setprio(P1R_PRIO_S);
        constexpr int NSTH = 64 * SH::NSW, DROW = p1r_g1_row_bytes<C>(), NPG = DROW / 16 / NSTH;      // 16-byte pieces per lane and row: 2 (C = 64), 5 (C = 80)
        static_assert(NPG * NSTH * 16 == DROW, "whole pieces");
        const int stid = q * 64 + lane;
        constexpr int NPO = C / 8, NIT = (SH::VWMAX * NPO + NSTH - 1) / NSTH;
#pragma unroll 1
        for (int u = u0; u < u1; ++u) {
            Ap = p1r_fresh(Ap);
            P1RItem I;
            if (!item(u, I)) continue;
            const int t = I.t, Y0 = I.Y0, Y1 = I.Y1, seg = Y1 - Y0;
            const int NS = (seg + SH::WARM + 1) & ~1;
            // piece e of a row = plane e / 16 (wave, group, column mod 4), lane position e % 16: the 8 channels of group e / 64
            h2_t cm[NPG][4];
            int gl[NPG];
#pragma unroll
            for (int k = 0; k < NPG; ++k) {
                const int e = stid + NSTH * k, plane = e >> 4, gi = plane >> 2;
                gl[k] = plane * GPL + ((e & 15) + 1) * 16;
                const float4 c0 = *(const float4*)(A.g1_scale + (size_t)t * C + 8 * gi), c1 = *(const float4*)(A.g1_scale + (size_t)t * C + 8 * gi + 4);
                cm[k][0] = (h2_t){(_Float16)c0.x, (_Float16)c0.y}; cm[k][1] = (h2_t){(_Float16)c0.z, (_Float16)c0.w};
                cm[k][2] = (h2_t){(_Float16)c1.x, (_Float16)c1.y}; cm[k][3] = (h2_t){(_Float16)c1.z, (_Float16)c1.w};
            }
            const char* const drow = A.g1_store + ((size_t)t * A.P.nsx + I.s) * (size_t)h * DROW + stid * 16;
            uint4 XA[NPG], XB[NPG];
            auto issue_row = [&](int y, uint4* X) {
                const int yc = y < 0 ? 0 : (y < h ? y : h - 1);
#pragma unroll
                for (int k = 0; k < NPG; ++k) X[k] = *(const uint4*)(drow + (size_t)yc * DROW + k * NSTH * 16);
            };
            auto stage_row = [&](int slot, const uint4* X, int y) {           // g1 row y times the inner CALayer2's scale -> ring row `slot` (zero outside the image)
                const uint32_t m = (y >= 0 && y < h) ? 0xffffffffu : 0u;
                char* gs = lds_g + slot * GROW;
#pragma unroll
                for (int k = 0; k < NPG; ++k) {
                    uint4 o;
                    o.x = as_u(as_h2(X[k].x) * cm[k][0]) & m; o.y = as_u(as_h2(X[k].y) * cm[k][1]) & m;
                    o.z = as_u(as_h2(X[k].z) * cm[k][2]) & m; o.w = as_u(as_h2(X[k].w) * cm[k][3]) & m;
                    *(uint4*)(gs + gl[k]) = o;
                }
            };
            int so_l[NIT], so_g[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int e = stid + NSTH * k, px = e / NPO, pc = e - px * NPO, rc = I.olo + px;
                so_l[k] = ((rc & 3) * 16 + (rc >> 2)) * PSO + pc * 16;
                so_g[k] = rc < I.ohi ? (I.xo + rc) * C + pc * 8 : -1;
                if (so_g[k] < 0) so_l[k] = 0;
            }
            auto store_row = [&](int j) {
                const int yo = Y0 - 10 + j;
                const char* os = lds_o + ((j - 1) & 1) * OSLOT;
                bf16_t* const g2row = A.g2 + ((size_t)t * h + yo) * w * C;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const uint4 v = *(const uint4*)(os + so_l[k]);
                    if (so_g[k] >= 0) *(uint4*)(g2row + so_g[k]) = v;
                }
            };
            issue_row(Y0 - 5, XB);
            issue_row(Y0 - 4, XA);
            __syncthreads();
            int gslot = 0;
            // step j: g1 row Y0 - 5 + j (loaded two steps ago) -> ring row j mod 6 (what the A waves of the one-kernel pass write in step j), refill, store a g2 row
            auto step = [&](const int j, uint4* X) {
                if (j <= seg + 6) stage_row(gslot, X, Y0 - 5 + j);
                issue_row(Y0 - 3 + j, X);
                __builtin_amdgcn_sched_barrier(0);
                if (j >= 10 && j <= seg + 9) store_row(j);
                gslot = gslot == SH::GRING - 1 ? 0 : gslot + 1;
                __syncthreads();
            };
#pragma unroll 1
            for (int j = 0; j < NS; j += 2) {
                step(j, XB);
                step(j + 1, XA);
            }
            Ap = p1r_fresh(Ap);
            finish(u);
        }
    } else if (ICA != 3 && role == 2) {
        // =================================================== S: stagers ===================================================
        __builtin_amdgcn_s_setprio(P1R_PRIO_S);
        constexpr int LPP = SH::NSW, NSTH = 64 * SH::NSW;                     // lanes per region pixel; stager threads
        const int stid = q * 64 + lane, quad = stid / LPP, sub = stid % LPP;
        // Lane `sub` of a pixel moves the 16-byte pieces sub, sub + LPP, ...: the lanes' stores are neighbours, and the 8 lanes of a ds_write_b128
        // lane group must hit 8 distinct 16-byte bank groups (a [half][piece] split put all 8 on one: 48 % of all LDS cycles of the first version
        // were bank conflicts of these stores).  LPP = 2: 8 lanes = 4 consecutive pixels = the 4 N-tile planes, 32 bytes apart (mod 128).
        // LPP = 4: 8 lanes = 2 pixels x 4 pieces; the quads are permuted so that the two pixels are 2 apart = planes 64 bytes apart.
        const int spx = LPP == 4 ? ((quad & ~3) | ((quad & 1) << 1) | ((quad >> 1) & 1)) : quad;
        constexpr int NPC = K / 8, NP0 = (NPC + LPP - 1) / LPP;             // 16-byte pieces of a pixel's K channels; pieces per lane
        const bool lastv = LPP * (NP0 - 1) + sub < NPC;                       // (only the last piece of a lane can be missing)
        const int xpix = (spx & 3) * XPL + (spx >> 2) * PSX;
        constexpr int NPO = C / 8, NIT = (SH::VWMAX * NPO + NSTH - 1) / NSTH;
#pragma unroll 1
        for (int u = u0; u < u1; ++u) {
            Ap = p1r_fresh(Ap);
            P1RItem I;
            if (!item(u, I)) continue;
            const int t = I.t, Y0 = I.Y0, Y1 = I.Y1, seg = Y1 - Y0;
            const int NS = (seg + (ICA == 1 ? 7 : SH::WARM) + 1) & ~1;        // even: the stagers rotate two register sets (a padding step only has the barrier)
            const SnSlabs<bf16_t> sl = sn_unit_slabs<bf16_t>(A.x, A.halo, A.T, hw, C, A.mode, A.wrap, t);
            const bf16_t* sp[NP0];
            int sst[NP0];
#pragma unroll
            for (int i = 0; i < NP0; ++i) {
                const int pi = LPP * i + sub, pc = pi < NPC ? pi : 0;
                if (pc < CH / 8) { sp[i] = sl.p0 + 8 * pc; sst[i] = sl.s0; }
                else if (pc < C / 8) { sp[i] = sl.p1 + 8 * (pc - CH / 8); sst[i] = sl.s1; }
                else { sp[i] = A.hwb + (size_t)t * hw * CH + 8 * (pc - C / 8); sst[i] = CH; }
            }
            const int sgx = I.xo + spx, sgxc = (sgx >= 0 && sgx < w) ? sgx : 0;
            uint4 XA[NP0], XB[NP0];
            auto issue_row = [&](int y, uint4* X) {
                const int yc = (y >= 0 && y < h) ? y : 0;
                const int ii = yc * w + sgxc;
#pragma unroll
                for (int i = 0; i < NP0; ++i) X[i] = *(const uint4*)(sp[i] + ii * sst[i]);
            };
            auto stage_row = [&](int slot, const uint4* X, int y) {
                const bool inimg = y >= 0 && y < h && sgx >= 0 && sgx < w;
                float s1[2] = {0.f, 0.f};                                     // two partial sums: half as long dependency chains
                uint32_t wd[NP0][4];
#pragma unroll
                for (int i = 0; i < NP0; ++i) {
                    const bool v = i + 1 < NP0 || lastv;
                    wd[i][0] = v ? X[i].x : 0u; wd[i][1] = v ? X[i].y : 0u; wd[i][2] = v ? X[i].z : 0u; wd[i][3] = v ? X[i].w : 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s1[k & 1] = dot2bf(wd[i][k], 0x3f803f80u, s1[k & 1]);
                }
                float sm = s1[0] + s1[1];
                sm += dpp_mov<0xB1>(sm);                                      // the pixel's other lanes (quad_perm [1,0,3,2], then [2,3,0,1])
                if constexpr (LPP == 4) sm += dpp_mov<0x4E>(sm);
                const float mean = sm * (1.0f / K);
                const f32x2_t mean2 = {mean, mean};
                f32x2_t d[NP0][4];
                f32x2_t sq2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int i = 0; i < NP0; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x2_t v = {bf_lo(wd[i][k]), bf_hi(wd[i][k])};
                        d[i][k] = (i + 1 < NP0 || lastv) ? v - mean2 : (f32x2_t){0.f, 0.f};
                        sq2[k & 1] = __builtin_elementwise_fma(d[i][k], d[i][k], sq2[k & 1]);
                    }
                float sq = (sq2[0][0] + sq2[0][1]) + (sq2[1][0] + sq2[1][1]);
                sq += dpp_mov<0xB1>(sq);
                if constexpr (LPP == 4) sq += dpp_mov<0x4E>(sq);
                const float rstd = inimg ? __builtin_amdgcn_rsqf(sq * (1.0f / K) + 1e-6f) : 0.f;      // 0: an all-zero operand outside the image
                const f32x2_t rstd2 = {rstd, rstd};
                char* xs = lds_x + slot * XSLOT + xpix + sub * 16;
#pragma unroll
                for (int i = 0; i < NP0; ++i) {
                    uint4 o;
                    f32x2_t e0 = d[i][0] * rstd2, e1 = d[i][1] * rstd2, e2 = d[i][2] * rstd2, e3 = d[i][3] * rstd2;
                    o.x = pack_bf2(e0[0], e0[1]); o.y = pack_bf2(e1[0], e1[1]); o.z = pack_bf2(e2[0], e2[1]); o.w = pack_bf2(e3[0], e3[1]);
                    if (i + 1 < NP0 || lastv) *(uint4*)(xs + i * 16 * LPP) = o;
                }
                // the constant-one slots K, K + 1 (the bias columns of the weights); the rest of that piece stays zero
                if (sub == LPP - 1) *(uint32_t*)(lds_x + slot * XSLOT + xpix + K * 2) = inimg ? 0x3f803f80u : 0u;
            };
            // g2 rows leave as whole pixels: item e = (own pixel, 16-byte piece), fixed per lane for the whole walk
            int so_l[NIT], so_g[NIT];                                         // LDS byte offset inside an out slot / element offset inside a g2 row; -1: no item
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int e = stid + NSTH * k, px = e / NPO, pc = e - px * NPO, rc = I.olo + px;
                so_l[k] = ((rc & 3) * 16 + (rc >> 2)) * PSO + pc * 16;
                so_g[k] = rc < I.ohi ? (I.xo + rc) * C + pc * 8 : -1;
                if (so_g[k] < 0) so_l[k] = 0;
            }
            auto store_row = [&](int j) {                                     // g2 row Y0 - 10 + j, written to out slot (j - 1) & 1 by the B waves in step j - 1
                const int yo = Y0 - 10 + j;
                const char* os = lds_o + ((j - 1) & 1) * OSLOT;
                bf16_t* const g2row = A.g2 + ((size_t)t * h + yo) * w * C;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const uint4 v = *(const uint4*)(os + so_l[k]);
                    if (so_g[k] >= 0) *(uint4*)(g2row + so_g[k]) = v;
                }
            };
            issue_row(Y0 - 3, XA);
            issue_row(Y0 - 2, XB);
            stage_row(0, XA, Y0 - 3);
            issue_row(Y0 - 1, XA);
            __syncthreads();
            // step j: stage row Y0 - 2 + j (loaded two steps ago) into slot (j + 1) & 1, refill that register set with row Y0 + j, store a g2 row
            // (The loads are issued on EVERY path, past the walk's last row from a clamped row that hits in cache: the compiler's s_waitcnt
            //  counts assume the path with the fewest younger operations, and a path without the refill degrades every wait to "everything landed".)
            const int ylast = Y1 + 2;
            auto step = [&](const int j, uint4* X) {
                if (!(P1R_SKIP & 1) && j <= seg + 4) stage_row((j + 1) & 1, X, Y0 - 2 + j);
                if (!(P1R_SKIP & 2)) issue_row(Y0 + j < ylast ? Y0 + j : ylast, X);
                __builtin_amdgcn_sched_barrier(0);                            // the loads stay in front of the stores (vmcnt retires in order)
                if (!(P1R_SKIP & 4) && ICA != 1 && j >= 10 && j <= seg + 9) store_row(j);                        // (NS may contain one padding step)
                __syncthreads();
            };
#pragma unroll 1
            for (int j = 0; j < NS; j += 2) {
                step(j, XB);
                step(j + 1, XA);
            }
            Ap = p1r_fresh(Ap);
            finish(u);
        }
    } else if (ICA != 3 && role == 0) {
        // =================================================== A: first 1x1, 3x3, gate ===================================================
        __builtin_amdgcn_s_setprio(P1R_PRIO_A);
        bf16x8_t W1[2][KS1];
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            W1[0][s] = as_frag(A.wfrag1[((2 * q) * KS1 + s) * 64 + lane]);
            W1[1][s] = as_frag(A.wfrag1[((2 * q + 1) * KS1 + s) * 64 + lane]);
        }
        h2_t w3r[9][4];                                                       // packed-fp16 taps of the lane's four packed registers
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const uint4 v = A.w3[(q * 4 + g) * 9 + tp];
            w3r[tp][0] = as_h2(v.x); w3r[tp][1] = as_h2(v.y); w3r[tp][2] = as_h2(v.z); w3r[tp][3] = as_h2(v.w);
        }
        const h2_t hz = {(_Float16)0.f, (_Float16)0.f};
        const int xrd = p * PSX + g * 16;                                     // + slot, + n * XPL + 64 s (immediates)
        const int gwr = ((2 * q + (g >> 1)) * 4) * GPL + (p + 1) * 16 + (g & 1) * 8;      // + ring row, + n * GPL
#pragma unroll 1
        for (int u = u0; u < u1; ++u) {
            Ap = p1r_fresh(Ap);
            P1RItem I;
            if (!item(u, I)) continue;
            const int t = I.t, Y0 = I.Y0, Y1 = I.Y1, seg = Y1 - Y0;
            const int NS = (seg + (ICA == 1 ? 7 : SH::WARM) + 1) & ~1;
            bool colin[NX];
#pragma unroll
            for (int n = 0; n < NX; ++n) { const int gx = I.xo + NX * p + n; colin[n] = gx >= 0 && gx < w; }
            h2_t P0[NX][4], P1[NX][4];               // pending rows of the 3x3: when row y arrives P1 = row y-1 (lacks row y), P0 = row y (lacks y, y+1)
#pragma unroll
            for (int n = 0; n < NX; ++n)
#pragma unroll
                for (int k = 0; k < 4; ++k) { P0[n][k] = hz; P1[n][k] = hz; }
            // The wave is software-pipelined ACROSS steps: step j first runs the 3x3 + gate (VALU) on the packed `a` row of step j - 1, THEN the
            // 1x1 MFMAs of input row Y0 - 3 + j.  A step of a B wave starts with MFMAs (second 1x1) and ends its first half with VALU (exp / rcp),
            // so the roles sharing a SIMD are in complementary phases after every barrier -- with the MFMAs first, A and B waves queued for the
            // matrix core together and for the VALU together, and a SIMD's step took the SUM of its VALU and MFMA time (4500 cycles: 2000 - 2700
            // VALU + 1700 - 2700 MFMA) although the two pipes do overlap across waves (tools/ubench/mfma_valu_overlap.hip).  The stencil and the
            // MFMAs of one step are independent: the compiler may mix them.
            uint32_t ah[NX][4];                                               // packed fp16 `a` row of the previous step's input row
#pragma unroll
            for (int n = 0; n < NX; ++n)
#pragma unroll
                for (int k = 0; k < 4; ++k) ah[n][k] = 0u;
            // denoisers: the inner CALayer2 between SimpleGate and RepConv.  Pass 1 (g1_sums) reduces the channel sums of g1 over the walk's own
            // pixels (pool -> the squeeze-excite tail gives the scale), pass 2 multiplies g1 by that scale here; g1 itself never leaves the CU.
            h2_t cmul[2] = {{(_Float16)1.f, (_Float16)1.f}, {(_Float16)1.f, (_Float16)1.f}};
            if constexpr (ICA == 2) {
                const float4 cs = *(const float4*)(A.g1_scale + (size_t)t * C + 16 * q + 4 * g);
                cmul[0] = (h2_t){(_Float16)cs.x, (_Float16)cs.y}; cmul[1] = (h2_t){(_Float16)cs.z, (_Float16)cs.w};
            }
            // sums pass with a g1 store: this lane's 8 bytes of the strip's row y, N-tile n sit at + y * row bytes + n * 256
            char* const gdump = (ICA == 1 && A.g1_store) ? A.g1_store + ((size_t)t * A.P.nsx + I.s) * (size_t)h * p1r_g1_row_bytes<C>()
                                                             + ((2 * q + (g >> 1)) * 4) * 256 + p * 16 + (g & 1) * 8 : nullptr;
            float gsum[4] = {0.f, 0.f, 0.f, 0.f};
            float* const psums = (ICA == 1 && A.pool) ? A.pool + ((size_t)t * nrows + (size_t)I.s * nbh) * C + 16 * q : nullptr;      // + row block * C
            bool ownc[NX];                                                    // lane masks, not registers: this pass sits at the register limit
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                const int rc = NX * p + n;
                ownc[n] = ICA == 1 && rc >= I.olo && rc < I.ohi && colin[n];
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the weights have landed (no conservative waits inside the loop)
            __syncthreads();
            int gslot = 0;                                                    // j mod 6
#pragma unroll 1
            for (int j = 0; j < NS; ++j) {
                if (!(P1R_SKIP & 8) && j <= seg + 6) {
                    // ---- (1) depthwise 3x3 (+identity) on a row ya = Y0 - 4 + j, scatter form: it completes output row ya - 1, feeds row ya, opens row ya + 1 ----
                    h2_t F[NX][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t Lw = lane_prev(ah[NX - 1][k]), Rw = lane_next(ah[0][k]);      // the two operands that cross the lane boundary
#pragma unroll
                        for (int ti = 0; ti < 3; ++ti) {                      // ty = 2 first: it reads P1 before ty = 1 overwrites it (from P0), then ty = 0 reopens P0
                            const int ty = 2 - ti;
#pragma unroll
                            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                                for (int n = 0; n < NX; ++n) {
                                    const uint32_t src = tx == 0 ? (n > 0 ? ah[n - 1][k] : Lw) : (tx == 1 ? ah[n][k] : (n + 1 < NX ? ah[n + 1][k] : Rw));
                                    const h2_t v = as_h2(src), wk = w3r[ty * 3 + tx][k];
                                    if (ty == 2) F[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? P1[n][k] : F[n][k]);
                                    else if (ty == 1) P1[n][k] = __builtin_elementwise_fma(v, wk, tx == 0 ? P0[n][k] : P1[n][k]);
                                    else P0[n][k] = tx == 0 ? v * wk : __builtin_elementwise_fma(v, wk, P0[n][k]);
                                }
                        }
                    }
                    // SimpleGate -> g1 row Y0 - 5 + j (zero outside the image: the zero padding of the RepConv) -> ring row j mod 6
                    const int yg = Y0 - 5 + j;
                    const bool rin = yg >= 0 && yg < h;
                    char* gs = lds_g + gslot * GROW + gwr;
#pragma unroll
                    for (int n = 0; n < NX; ++n) {
                        const uint32_t m = (rin && colin[n]) ? 0xffffffffu : 0u;
                        const h2_t g1a = F[n][0] * F[n][2], g1b = F[n][1] * F[n][3];
                        if constexpr (ICA == 1) {
                            if (gdump && yg >= Y0 && yg < Y1)                              // every image row of the strip is stored by exactly one walk
                                *(uint2*)(gdump + (size_t)yg * p1r_g1_row_bytes<C>() + n * 256) = make_uint2(as_u(g1a) & m, as_u(g1b) & m);
                            const bool cnt = yg >= Y0 && yg < Y1 && ownc[n];               // every pixel of the frame is counted by exactly one walk
                            gsum[0] += cnt ? (float)g1a[0] : 0.f; gsum[1] += cnt ? (float)g1a[1] : 0.f;
                            gsum[2] += cnt ? (float)g1b[0] : 0.f; gsum[3] += cnt ? (float)g1b[1] : 0.f;
                        } else if constexpr (ICA == 2) {
                            *(uint2*)(gs + n * GPL) = make_uint2(as_u(g1a * cmul[0]) & m, as_u(g1b * cmul[1]) & m);
                        } else {
                            *(uint2*)(gs + n * GPL) = make_uint2(as_u(g1a) & m, as_u(g1b) & m);
                        }
                    }
                    if constexpr (ICA == 1) {
                        if (psums && yg >= Y0 && yg < Y1 && (((yg + 1) & (P1R_RB - 1)) == 0 || yg == Y1 - 1)) {
                            // the row block is complete: its channel sums of g1 (carried times 2^-4: undone here, exactly) -> its pool row
                            const int ln = p1r_lane();
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float sm = row_sum16(gsum[r]) * 16.0f;
                                if ((ln & 15) == 0) sn_pool_store(psums + (size_t)(yg / P1R_RB) * C + 4 * (ln >> 4) + r, sm);
                                gsum[r] = 0.f;
                            }
                        }
                    }
                    // ---- (2) first 1x1 on input row Y0 - 3 + j (x slot j & 1) -> packed fp16 `a` row for the next step.  Item i = (k-step i / NX,
                    //      tile i % NX): one fragment, two MFMAs (the wave's two M-tiles); fragments are read P1R_DA items ahead ----
                    const char* xs = lds_x + (j & 1) * XSLOT + xrd;
                    // (two N-tiles at a time: 16 accumulator registers live; the MFMA issue rate does not depend on the number of chains,
                    //  tools/ubench/mfma_chains.hip)
#pragma unroll
                    for (int n0 = 0; n0 < NX; n0 += 2) {
                        constexpr int NI = 2 * KS1, DA0 = ICA != 0 ? 2 : P1R_DA, DA = DA0 < NI ? DA0 : NI;      // (ICA 1 / 2 carry the sums / the scale registers: one fragment less in flight, no spill)
                        auto rdx = [&](const int i) -> uint4 { return *(const uint4*)(xs + (n0 + (i & 1)) * XPL + 64 * (i >> 1)); };
                        uint4 bq[NI];
#pragma unroll
                        for (int i = 0; i < DA; ++i) bq[i] = rdx(i);
                        __builtin_amdgcn_sched_group_barrier(0x100, DA, 0);
                        f32x4_t acc[2][2];
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int n = i & 1, s_ = i >> 1;
                            if (i + DA < NI) bq[i + DA] = rdx(i + DA);
                            if (s_ == 0) { acc[n][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[n][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
                            acc[n][0] = mfma16(W1[0][s_], as_frag(bq[i]), acc[n][0]); acc[n][1] = mfma16(W1[1][s_], as_frag(bq[i]), acc[n][1]);
                            if (i + DA < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        }
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            ah[n0 + n][0] = cvt_pk_h2(acc[n][0][0], acc[n][0][1]); ah[n0 + n][1] = cvt_pk_h2(acc[n][0][2], acc[n][0][3]);
                            ah[n0 + n][2] = cvt_pk_h2(acc[n][1][0], acc[n][1][1]); ah[n0 + n][3] = cvt_pk_h2(acc[n][1][2], acc[n][1][3]);
                        }
                    }
                }
                gslot = gslot == SH::GRING - 1 ? 0 : gslot + 1;
                __syncthreads();
            }
            Ap = p1r_fresh(Ap);
            finish(u);
        }
    } else if constexpr (ICA == 1) {
        // (the sums pass has no B waves)
    } else {
        // (ICA 3: no wave has role 0; the branch above is dead code there)
        // =================================================== B: RepConv, second 1x1, gate2 ===================================================
        __builtin_amdgcn_s_setprio(P1R_PRIO_B);
        uint4 Wg[2][8];
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int s = 0; s < 8; ++s) Wg[G][s] = A.wgrp[((q * 2 + G) * 8 + s) * 64 + lane];
        uint4 W2[2][KS2];
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            W2[0][s] = A.wfrag2[((2 * q) * KS2 + s) * 64 + lane];
            W2[1][s] = A.wfrag2[((2 * q + 1) * KS2 + s) * 64 + lane];
        }
        // RepConv B fragments.  Pair-tile u, lane p = pixels (4p + 2u, 4p + 2u + 1) of the region; tap (dy, dx6) reads column 4p + 2u + dx6 - 2 of
        // g1 row yr - 2 + dy.  Ring column = column + 4: plane (column mod 4), position p + 1 + floor((2u + dx6 - 2) / 4).
        // steps 0..5: dy = lane group, dx6 = step  -> per-lane ring row, compile-time plane / position
        // steps 6, 7: dy = 4, dx6 = lane group (+ 4)   -> uniform ring row, per-lane plane / position
        const int gq0 = q * 2 * 4 * GPL + p * 16;
        int off67[2][2];                                                      // [u][step - 6], without the ring row
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int s7 = 0; s7 < 2; ++s7) {
                const int dx6 = s7 ? 4 + (g & 1) : g, e = 2 * uu + dx6 - 2 + 4;                  // + 4: non-negative
                off67[uu][s7] = gq0 + (e & 3) * GPL + (e >> 2) * 16;
            }
        const int rwr = p * PSR + (16 * q + 4 * (g & 1)) * 2;                // + slot, + (2u + (g >> 1)) * 16 PSR, + 16 G
        const int rrd = p * PSR + g * 16;                                     // + slot, + n * 16 PSR + 64 s
        const int owr = p * PSO + (16 * q + 4 * g) * 2;                       // + slot, + n * 16 PSO
#pragma unroll 1
        for (int u = u0; u < u1; ++u) {
            Ap = p1r_fresh(Ap);
            P1RItem I;
            if (!item(u, I)) continue;
            const int seg = I.Y1 - I.Y0;
            const int NS = (seg + SH::WARM + 1) & ~1;
            float own[NX];
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                const int rc = NX * p + n;
                own[n] = (rc >= I.olo && rc < I.ohi) ? 1.f : 0.f;
            }
            float psum[4] = {0.f, 0.f, 0.f, 0.f};
            float* const psums = A.pool ? A.pool + ((size_t)I.t * nrows + (size_t)I.s * nbh) * C + 16 * q : nullptr;      // + row block * C
            const int Y0 = I.Y0, Y1 = I.Y1;
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            int jm = 1;                                                       // (j - 5) mod 6
#pragma unroll 1
            for (int j = 0; j < NS; ++j) {
                // Per step: (1) second 1x1 + SimpleGate2 on the r row of the PREVIOUS step (g2 row Y0 - 9 + j -> out slot j & 1), (2) RepConv: r row
                // Y0 - 8 + j from g1 ring rows (j - 5 + dy) mod 6 -> r slot j & 1.  In the steady state both run in ONE basic block, the 1x1 in tile
                // pairs, so that the exp / rcp / pack work of a pair is scheduled between the MFMAs that follow it (next pair, RepConv).
                const bool do1 = !(P1R_SKIP & 16) && j >= 9 && j <= seg + 8, do2 = !(P1R_SKIP & 32) && j >= 8 && j <= seg + 7;
                const char* rs1 = lds_r + ((j - 1) & 1) * RSLOT + rrd;
                char* os = lds_o + (j & 1) * OSLOT + owr;
                int r6 = jm + g;
                r6 = r6 >= SH::GRING ? r6 - SH::GRING : r6;
                const char* gb = lds_g + r6 * GROW + gq0;                     // steps 0..5: this lane group's ring row
                const int r4 = jm + 4 >= SH::GRING ? jm + 4 - SH::GRING : jm + 4;
                const char* g4 = lds_g + r4 * GROW;                           // steps 6, 7: ring row of dy = 4
                char* rs2 = lds_r + (j & 1) * RSLOT + rwr;
                auto gemm2 = [&](const int n0) {                              // tiles n0, n0 + 1: item i = (k-step i / 2, tile n0 + i % 2), two MFMAs each: four chains
                    constexpr int NI = 2 * KS2;
                    uint4 bq[NI];
#pragma unroll
                    for (int i = 0; i < NI; ++i) bq[i] = *(const uint4*)(rs1 + (n0 + (i & 1)) * 16 * PSR + 64 * (i >> 1));
                    __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
                    f32x4_t c[2][2];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int n = i & 1, s_ = i >> 1;
                        if (s_ == 0) { c[n][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; c[n][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
                        c[n][0] = mfma16h(W2[0][s_], bq[i], c[n][0]); c[n][1] = mfma16h(W2[1][s_], bq[i], c[n][1]);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                         // b1 * sigmoid(b2); the gate rows carry -log2(e) (prep.pack_phase1r)
                            if (P1R_SKIP & 64) v[r] = c[n][0][r] * c[n][1][r];               // (measurement: the step without its 32 transcendentals)
                            else v[r] = c[n][0][r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(c[n][1][r]));
                            psum[r] = fmaf(v[r], own[n0 + n], psum[r]);
                        }
                        *(uint2*)(os + (n0 + n) * 16 * PSO) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
                };
                auto repconv = [&](auto nv) {
                    constexpr int nvalu = decltype(nv)::value;
                    // item i = (k-step i / 4, group (i / 2) % 2, pair-tile i % 2): one fragment, one MFMA, four accumulator chains in turn;
                    // fragments are read P1R_DB items ahead (see the A waves)
                    auto rd = [&](const int i) -> uint4 {
                        const int s_ = i >> 2, G = (i >> 1) & 1, uu = i & 1;
                        if (s_ < 6) {
                            const int e = 2 * uu + s_ - 2 + 4;
                            return *(const uint4*)(gb + (G * 4 + (e & 3)) * GPL + (e >> 2) * 16);
                        }
                        return *(const uint4*)(g4 + off67[uu][s_ - 6] + G * 4 * GPL);
                    };
                    constexpr int NI = 32, DB = P1R_DB;
                    uint4 bq[NI];
#pragma unroll
                    for (int i = 0; i < DB; ++i) bq[i] = rd(i);
                    __builtin_amdgcn_sched_group_barrier(0x100, DB, 0);
                    f32x4_t acc[2][2];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int s_ = i >> 2, G = (i >> 1) & 1, uu = i & 1;
                        if (i + DB < NI) bq[i + DB] = rd(i + DB);
                        if (s_ == 0) acc[G][uu] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                        acc[G][uu] = mfma16h(Wg[G][s_], bq[i], acc[G][uu]);
                        if (i + DB < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if constexpr (nvalu > 0) __builtin_amdgcn_sched_group_barrier(0x002, nvalu, 0);
                    }
#pragma unroll
                    for (int G = 0; G < 2; ++G)
#pragma unroll
                        for (int uu = 0; uu < 2; ++uu)                        // D row 4g + r = (oc = 4 (g & 1) + r, xp = g >> 1): pixel 4p + 2u + xp = N-tile 2u + xp
                            *(uint2*)(rs2 + (2 * uu + (g >> 1)) * 16 * PSR + 16 * G) =
                                make_uint2(cvt_pk_h2(acc[G][uu][0], acc[G][uu][1]), cvt_pk_h2(acc[G][uu][2], acc[G][uu][3]));
                };
                if (do1 && do2) { gemm2(0); gemm2(2); repconv(std::integral_constant<int, P1R_BV>{}); }
                else if (do1) { gemm2(0); gemm2(2); }
                else if (do2) repconv(std::integral_constant<int, 0>{});
                const int yo = Y0 - 9 + j;                                    // the g2 row gemm2 has just summed
                if (do1 && psums && (((yo + 1) & (P1R_RB - 1)) == 0 || yo == Y1 - 1)) {
                    // its row block is complete: the block's channel sums for CALayer2 (lane group g owns channels 16 q + 4 g + r) -> its pool row
                    const int ln = p1r_lane();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sm = row_sum16(psum[r]);
                        if ((ln & 15) == 0) sn_pool_store(psums + (size_t)(yo / P1R_RB) * C + 4 * (ln >> 4) + r, sm);
                        psum[r] = 0.f;
                    }
                }
                jm = jm == SH::GRING - 1 ? 0 : jm + 1;
                __syncthreads();
            }
            Ap = p1r_fresh(Ap);
            finish(u);
        }
    }
#undef A
}

int p1r_ncu() {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) return -1;
    return ncu;
}

}  // namespace

// Work decomposition of one launch over nfr frames of h x w (see P1RPlan); team: 0 = choose, else 1 / 2 / 4 / 8.  Host-callable from the tests
// (tests/test_host_logic.py walks every plan and checks that each row of each strip of each frame is produced exactly once).
extern "C" int sn_p1r_plan(int nfr, int h, int w, int ncu, int team, int* out7) {
    if (nfr < 1 || h < 1 || w < 1 || ncu < 1 || !out7) return SN_EINVAL;
    if (team != 0 && team != 1 && team != 2 && team != 4 && team != 8) return SN_EINVAL;
    P1RPlan P;
    if (w <= 64) { P.nsx = 1; P.sd = P.sr = 0; }
    else {
        int n = 2;
        while (122 + 58 * (n - 2) < w) ++n;
        const int slack = 122 + 58 * (n - 2) - w;
        P.nsx = n; P.sd = slack / n; P.sr = slack % n;
    }
    const int per_max = ncu / 8 > 0 ? ncu / 8 : 1;
    const int nbh = (h + P1R_RB - 1) / P1R_RB;                                // chunks are whole row blocks (pool rows are per block)
    const int qmin = nbh < 2 ? nbh : 2;                                       // no chunk much shorter than its own warm-up
    // team size: steps of the slowest workgroup ~ chunk rows + one warm-up per walk the chunk touches; among the sizes within 3 % of the best the
    // LARGEST wins (more frame pairs in lock step = fewer 128-byte lines fetched twice); a ragged last frame block idles its missing members
    long best = -1;
    long est[4], qs[4];
    for (int i = 0; i < 4; ++i) {
        const int F = 8 >> i;
        est[i] = -1;
        if ((team && F != team) || F > per_max || per_max % F) continue;
        const long blocks = (long)P.nsx * ((nfr + F - 1) / F) * nbh, tmax = 8L * per_max / F;
        if (blocks > 0x3fffffff) return SN_EINVAL;
        long q = (blocks + tmax - 1) / tmax;
        if (q < qmin) q = qmin;
        qs[i] = q;
        est[i] = q * P1R_RB + 10 * ((q + nbh - 1) / nbh + 1);
        if (best < 0 || est[i] < best) best = est[i];
    }
    if (best < 0) return SN_EINVAL;                                           // a fixed team size this device cannot place
    int pick = -1;
    for (int i = 0; i < 4 && pick < 0; ++i)
        if (est[i] >= 0 && 100 * est[i] <= 103 * best) pick = i;
    P.F = 8 >> pick; P.nfb = (nfr + P.F - 1) / P.F;
    const long blocks_all = (long)P.nsx * P.nfb * nbh, q = qs[pick];
    P.q = (int)q; P.nteam = (int)((blocks_all + q - 1) / q);
    out7[0] = P.nsx; out7[1] = P.sd; out7[2] = P.sr; out7[3] = P.F; out7[4] = P.nfb; out7[5] = P.q; out7[6] = P.nteam;
    return SN_OK;
}
// first own column of strip s of a plan (s == nsx: w): the tests' view of p1r_strip_begin
extern "C" int sn_p1r_strip_begin(const int* plan7, int s, int w) {
    P1RPlan P; P.nsx = plan7[0]; P.sd = plan7[1]; P.sr = plan7[2]; P.F = plan7[3]; P.nfb = plan7[4]; P.q = plan7[5]; P.nteam = plan7[6];
    return p1r_strip_begin(P.nsx, P.sd, P.sr, s, w);
}

namespace {

int p1r_plan(int nfr, int h, int w, int ncu, int team, P1RPlan& P) {
    int o[7];
    const int rc = sn_p1r_plan(nfr, h, w, ncu, team, o);
    if (rc != SN_OK) return rc;
    P.nsx = o[0]; P.sd = o[1]; P.sr = o[2]; P.F = o[3]; P.nfb = o[4]; P.q = o[5]; P.nteam = o[6];
    return SN_OK;
}

template <int C, bool HW, int ICA>
int p1r_launch1(P1RArgs& A, hipStream_t st) {
    using SH = P1RShape<C, HW>;
    if (hipFuncSetAttribute((const void*)cab_phase1r_kernel<C, HW, ICA>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::LDS) != hipSuccess) return SN_ELAUNCH;
    sn_clear_error();
    int per = (A.P.F * A.P.nteam + 7) / 8;
    per = (per + A.P.F - 1) / A.P.F * A.P.F;                                 // whole teams per XCD
    hipLaunchKernelGGL((cab_phase1r_kernel<C, HW, ICA>), dim3((unsigned)(8 * per)), dim3(p1r_threads<C, HW, ICA>()), SH::LDS, st, A);
    return sn_check_launch();
}
template <int C, bool HW>
int p1r_launch(P1RArgs& A, hipStream_t st) {
    if (A.g1_sums) return p1r_launch1<C, HW, 1>(A, st);
    if (A.g1_scale && A.g1_store) return p1r_launch1<C, HW, 3>(A, st);
    return A.g1_scale ? p1r_launch1<C, HW, 2>(A, st) : p1r_launch1<C, HW, 0>(A, st);
}

int cab_phase1(const sn_unit_src* s, const void* hw, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se,
               const sn_phase1_opts* opt, void* stream) {
    sn_clear_error();
    const bool sums = opt && opt->g1_sums;
    if (!s || !s->x || (s->C != 64 && s->C != 80) || s->mode < 0 || s->mode > 2 || s->T < 1 || s->h < 1 || s->w < 1 || !wt || !wt->wfrag1 || !wt->w3 ||
        !wt->wgrp || !wt->wfrag2 || (!g2 && !sums) || (sums && !pool) || (s->mode != 0 && !hw) || s->wrap < 0 || s->wrap > 2 || (s->wrap == 2 && s->mode != 0 && !s->halo)) return SN_EINVAL;
    const int ncu = p1r_ncu();
    if (ncu < 1) return SN_ELAUNCH;
    P1RArgs A;
    A.x = (const bf16_t*)s->x; A.halo = (const bf16_t*)s->halo; A.hwb = (const bf16_t*)hw; A.T = s->T; A.h = s->h; A.w = s->w; A.mode = s->mode; A.wrap = s->wrap;
    A.wfrag1 = (const uint4*)wt->wfrag1; A.w3 = (const uint4*)wt->w3; A.wgrp = (const uint4*)wt->wgrp; A.wfrag2 = (const uint4*)wt->wfrag2;
    A.g2 = (bf16_t*)g2; A.pool = pool;
    A.g1_scale = opt ? opt->g1_scale : nullptr; A.g1_sums = sums ? 1 : 0;
    A.g1_store = opt ? (char*)opt->g1_store : nullptr;
    if (A.g1_store && !sums && !A.g1_scale) return SN_EINVAL;               // a g1 store belongs to the two passes of the denoisers
    A.se.ca = nullptr; A.se.bad = nullptr;
    if (se) {
        if (!pool || !se->wa || !se->wb || !se->ticket || !se->ca || se->c != s->C || se->cr < 1 || se->cr > 128) return SN_EINVAL;
        A.se.wa = se->wa; A.se.wb = se->wb; A.se.ca = se->ca; A.se.ticket = se->ticket; A.se.bad = se->bad; A.se.inv_hw = 1.0f / ((float)s->h * (float)s->w);
        A.se.c = se->c; A.se.cr = se->cr;
    }
    SN_FRAME_RANGE(s, t0, nt);
    A.t0 = t0; A.nfr = nt;
    const int rc = p1r_plan(nt, s->h, s->w, ncu, opt ? opt->team : 0, A.P);      // the pool layout ([T][strips][row blocks][C]) does not depend on the frame range
    if (rc != SN_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (s->C == 80) return s->mode ? p1r_launch<80, true>(A, st) : p1r_launch<80, false>(A, st);
    return s->mode ? p1r_launch<64, true>(A, st) : p1r_launch<64, false>(A, st);
}

}  // namespace

extern "C" {

int sn_phase1_g1_store_bytes(int T, int h, int w, int C, long long* bytes) {
    if (T < 1 || h < 1 || w < 1 || (C != 64 && C != 80) || !bytes) return SN_EINVAL;
    int o[7];
    const int rc = sn_p1r_plan(1, h, w, 8, 1, o);                             // the strip count depends on w alone
    if (rc != SN_OK) return rc;
    *bytes = (long long)T * o[0] * h * (C == 80 ? p1r_g1_row_bytes<80>() : p1r_g1_row_bytes<64>());
    return SN_OK;
}

int sn_phase1_pool_blocks(int T, int h, int w) {
    if (T < 1 || h < 1 || w < 1) return SN_EINVAL;
    int o[7];
    const int rc = sn_p1r_plan(1, h, w, 8, 1, o);                             // the strip count depends on w alone
    return rc == SN_OK ? o[0] * ((h + P1R_RB - 1) / P1R_RB) : rc;
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
