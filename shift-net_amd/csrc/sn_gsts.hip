// Grouped spatial-temporal shift unit (channel_shift -> CAB2 -> CAB1) for gfx950.
//
// Production kernels of this file:
//   sn_gsts_shiftconv (K0): hw = dw3x3(spatial_shift(borrowed half))     LDS-staged gather, never materialises the shift
//   sn_gsts_cab2_phase2 / sn_cab1_phase2 (K4): y  = roll(x) + beta * W3 . (ca * g2)         per-pixel MFMA, rolled shortcut
//   sn_gsts_gather / sn_temporal_roll: validation op / Shift_CAB roll (pure index work)
// The temporal roll is only ever an address computation (frame/channel-offset pairs below).
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

namespace {

struct UnitK {
    const bf16_t* x; const bf16_t* halo;
    int T, h, w, C, mode, wrap, t0;
};
__device__ __forceinline__ SnSlabs<bf16_t> unit_slabs(const UnitK& U, int t) {
    return sn_unit_slabs<bf16_t>(U.x, U.halo, U.T, U.h * U.w, U.C, U.mode, U.wrap, t);
}

// ------------------------------------------------------------------------------------------------------------
// validation op: u = cat(y, shift(hw))
__global__ void gather_kernel(const UnitK U, const int8_t* offs, bf16_t* u, const int CU) {
    const int t = U.t0 + blockIdx.y, Ch = U.C >> 1, hw = U.h * U.w;
    const SnSlabs<bf16_t> s = unit_slabs(U, t);
    const size_t n = (size_t)hw * CU;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / CU), c = (int)(e - (size_t)i * CU);
        bf16_t v = 0;
        if (c < Ch) v = s.p0[(size_t)i * s.s0 + c];
        else if (c < U.C) v = s.p1[(size_t)i * s.s1 + c - Ch];
        else {
            const int k = c - U.C, y = i / U.w, x = i - y * U.w;
            const int sy = y + offs[2 * k], sx = x + offs[2 * k + 1];
            if (sy >= 0 && sy < U.h && sx >= 0 && sx < U.w) v = s.pb[((size_t)sy * U.w + sx) * s.sb + k];
        }
        u[(size_t)t * n + e] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K0: hw[t][p][k] = sum_tap w1[k][tap] * [p+tap in image] * shifted_k(p+tap),  shifted_k(q) = x[fb][q + off_k][ob + k] or 0
// PP = 8-channel chunks of the 34 x 34 window staged per pass = the whole borrowed half at once: 79 / 97 KB of LDS = two / one
// 4-wave workgroups per CU.  (Two chunks per pass -- 42 KB, three workgroups per CU -- measured slower in round 2: 12.4 vs 10.1 ms
// per window of config 2, 51.8 vs 48.9 ms of config 3.)
// Not VALU-bound either, although 38 % of its wave cycles issue VALU: a round-3 build with the staging loop's index arithmetic made
// incremental (~750 instead of ~1170 VALU per wave, bit-identical) measured 10.68 vs 10.30 ms per window on the same box.  What the
// kernel waits for is its own load -> LDS -> barrier -> compute chain at two workgroups per CU.
// A persistent, software-pipelined build (64 workgroups per XCD walking tile lists, the next tile's window in flight in 76 registers
// while the current one is computed; bit-identical) was slower as well: 11.88 vs 10.33 ms per window of config 2, 74.5 vs 49.4 ms of
// config 3 (C = 80: one workgroup per CU) -- the fourth register-prefetch pipeline on this path that lost to plain occupancy.  So was a
// ROW-WALKING build (strip of 32 columns, ring of 20 input rows in LDS: 1.56 instead of 4.5 window bytes per output byte): 11.3 vs
// 10.05 ms -- the kernel is not bound by the bytes it loads either.
// Round 5: channel pairs that share a displacement (8 of the 24 offsets of C = 64, 16 of C = 80 carry two channels) read as ONE 4-byte LDS word
// per tap with two v_dot2c (weight words with one non-zero half) -- 25 % / 40 % fewer LDS reads, bit-identical -- measured SLOWER, interleaved
// against the round-4 library on one device (tools/p1_ab.py): 374.8 vs 323.2 us (C = 64, 20 x 360 x 640), 1761.6 vs 1592.7 us (C = 80, 52 x 360 x
// 640): the wave-uniform pair test splits the unrolled tap loop into two bodies and the scheduler loses its order of LDS reads and dot
// products.  The sixth rebuild of this kernel that lost.
// And the seventh: the channel chunks of a tile split over TWO neighbouring workgroups (C = 64: 2 + 2 chunks, 46 KB of LDS, 76 VGPRs, three workgroups per
// CU instead of two; C = 80: 3 + 2 chunks, 63 KB, two instead of one), both halves adjacent in the XCD-aware walk so that the second finds the window in
// L2: 491.7 vs 317.6 us (C = 64), 2636.6 vs 1586.1 us (C = 80, 52 x 360 x 640), 1823.9 vs 1077.8 us (1080p): the loader then takes 32 / 48 bytes of every
// 64- / 80-byte pixel record -- half-used sectors and twice the staging index work per output byte cost far more than the extra residency buys.
template <int CH, int PP>
__global__ __launch_bounds__(256) void shiftconv_kernel(const UnitK U, const XcdTiles G, const int8_t* __restrict__ offs,
                                                      const uint32_t* __restrict__ w1d, bf16_t* hw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RW = 34, PSB = PP * 16 + 4;      // odd number of dwords per pixel: lanes = pixels hit distinct banks
    // ... and a row pitch of 16 (mod 32) dwords: a ds_read_u16 serves 32 lanes = 16 pixels of TWO tile rows, whose bank sets {17 px} and
    // {17 px + pitch} must not meet.  With the natural pitch (34 x 17 = 2 mod 32 dwords) half of the kernel's LDS cycles were bank
    // conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50, tools/lds_probe_k0.sh).
    constexpr int ROWP = RW * PSB + (((16 - (RW * PSB / 4) % 32) + 32) % 32) * 4;
    constexpr int PCS = CH / 8;
    int t, ty_, tx_;
    if (!sn_xcd_tile(G, t, ty_, tx_)) return;      // XCD-aware walk: the 34-wide windows of x-neighbours overlap by 18 columns
    t += U.t0;
    const int tid = threadIdx.x, y0 = ty_ * 16, x0 = tx_ * 16;
    const SnSlabs<bf16_t> s = unit_slabs(U, t);
    const bf16_t* src = s.pb;
    const int sstr = s.sb;
    const int px = tid & 15, py = tid >> 4, oy = y0 + py, ox = x0 + px;
    const bool valid = oy < U.h && ox < U.w;
    // interior tiles (every tap position p+tap of every output pixel is inside the image): no conv-padding masks at all,
    // and each tap is ONE v_dot2c on the zero-extended bf16 value with a packed weight word (bf16 weight in the low half)
    const bool interior = y0 >= 1 && x0 >= 1 && y0 + 16 < U.h && x0 + 16 < U.w;
#pragma unroll
    for (int pc0 = 0; pc0 < PCS; pc0 += PP) {
        const int np = PCS - pc0 < PP ? PCS - pc0 : PP;           // chunks of this pass (compile-time after unrolling)
        {   // staging: issue ALL global loads first (branch-free, clamped addresses), then mask + write to LDS: one memory
            // round trip per pass instead of one per loop iteration
            constexpr int NIT = (RW * RW * PP + 255) / 256;
            uint4 v[NIT];
            int lo[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * 256;
                const int pix = idx / np, pc = idx - pix * np;
                const int ry = pix / RW, rx = pix - ry * RW;
                const int gy = y0 - 9 + ry, gx = x0 - 9 + rx;
                const bool in = idx < RW * RW * np && gy >= 0 && gy < U.h && gx >= 0 && gx < U.w;
                lo[k] = idx < RW * RW * np ? (in ? ry * ROWP + rx * PSB + pc * 16 : -(ry * ROWP + rx * PSB + pc * 16) - 1) : 0x7fffffff;
                v[k] = *(const uint4*)(src + (in ? ((size_t)gy * U.w + gx) * sstr + (pc0 + pc) * 8 : 0));
            }
            if (pc0) __syncthreads();                             // every wave is done reading the previous pass
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                if (lo[k] == 0x7fffffff) continue;
                const bool in = lo[k] >= 0;
                uint32_t* d = (uint32_t*)(smem + (in ? lo[k] : -(lo[k] + 1)));
                d[0] = in ? v[k].x : 0u; d[1] = in ? v[k].y : 0u; d[2] = in ? v[k].z : 0u; d[3] = in ? v[k].w : 0u;
            }
        }
        __syncthreads();
        if (interior) {
            for (int kc = pc0; kc < pc0 + np; ++kc) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kc * 8 + j;
                    const int dy = offs[2 * k], dx = offs[2 * k + 1];
                    const char* base = smem + (py + 8 + dy) * ROWP + (px + 8 + dx) * PSB + (k - pc0 * 8) * 2;
                    float acc = 0.f;
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx)
                            acc = dot2bf((uint32_t)(*(const bf16_t*)(base + ty * ROWP + tx * PSB)), w1d[k * 9 + ty * 3 + tx], acc);
                    o[j] = acc;
                }
                *(uint4*)(hw + (((size_t)t * U.h + oy) * U.w + ox) * CH + kc * 8) = pack8(o);
            }
        } else {
            float m[9];
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int qy = oy + ty - 1, qx = ox + tx - 1;
                    m[ty * 3 + tx] = (qy >= 0 && qy < U.h && qx >= 0 && qx < U.w) ? 1.f : 0.f;
                }
            for (int kc = pc0; kc < pc0 + np; ++kc) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kc * 8 + j;
                    const int dy = offs[2 * k], dx = offs[2 * k + 1];
                    const char* base = smem + (py + 8 + dy) * ROWP + (px + 8 + dx) * PSB + (k - pc0 * 8) * 2;
                    float acc = 0.f;
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx) {
                            const float v = m[ty * 3 + tx] * bf_to_f(*(const bf16_t*)(base + ty * ROWP + tx * PSB));
                            acc = dot2bf(__float_as_uint(v) >> 16, w1d[k * 9 + ty * 3 + tx], acc);   // same bf16 weights as the fast path
                        }
                    o[j] = acc;
                }
                if (valid) *(uint4*)(hw + (((size_t)t * U.h + oy) * U.w + ox) * CH + kc * 8) = pack8(o);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// K0 ON THE MATRIX CORES (round 6; VERDICT r05 item 4 "K0 out of NHWC" -- inside the kernel: the tensors stay NHWC, the LDS window is planar).
// The seven rebuilds above all kept the arithmetic of shiftconv_kernel -- per output pixel and channel nine 2-byte LDS reads and nine half-empty
// v_dot2c, 288 + 288 per pixel, because a channel's displaced window is 2 bytes wide in a pixel-major layout.  A depthwise 3x3 has no reduction
// over channels to feed an MFMA, but it has one over SPACE: for one channel k of a 16 x 16 output tile
//     out[n][m] = sum_{ty} sum_{j = m .. m + 2} w[ty][j - m] * W_k[n + ty + 8 + sy][j + 8 + sx]          (W: the 34 x 34 window, origin - 9)
// is a GEMM with M = 16 output columns m, N = 16 output rows n, K = (ty, j): k-step ty = 32 slots j (18 used).  A[m][(ty, j)] = w[ty][j - m] is
// a banded (Toeplitz) matrix of the channel's nine weights; B[(ty, j)][n] is the window itself -- eight consecutive j are eight consecutive pixels
// of ONE row of ONE channel: 16 contiguous bytes in a CHANNEL-PLANAR window.  So the loader transposes on its way into LDS (a 16-byte piece =
// 8 channels of a pixel -> eight 2-byte stores into eight planes; planes skewed by 32 bytes per 8 so that the pieces of a pixel hit different
// banks), and a B fragment is one 8-byte-aligned 16-byte read (shifts are multiples of 4 pixels): 3 reads + 3 MFMAs per channel and 16 x 16 tile
// instead of 2304 + 2304 VALU-side operations per wave.
// The A fragments are never stored: with ty = the k-step, a lane's fragment word is two of {w[ty][0], w[ty][1], w[ty][2], 0} chosen by the lane's
// (m, j) alone -- ONE v_perm_b32 with a per-lane selector (4 registers for the whole kernel) on two wave-uniform words of the channel's weights.
// (The first version kept 24 host-built fragments per wave in 96 registers and had to stage the window in three batches, three exposed
// memory round trips per tile: 296 us at C = 64, 20 x 360 x 640, of which the loader 160 -- profiles/r06_k0_mfma_first_version_ablation.txt.)
// The accumulator holds out[x = 4 g + r][y = p] of the channel; a wave owns the 8 channels of one 16-byte piece, so after its eight channels a
// lane stores four whole pieces.  Workgroups are persistent (64 / 32 per XCD walk that XCD's tile list in order: x-neighbours run concurrently
// on one L2, as in sn_xcd_tile).  Zero padding of the conv ([p + tap in image]) = masks on the B fragment of border tiles; zero fill of the
// shift = the window's own zero fill.  Same bf16 products, fp32 accumulation in the MFMA's order: within rounding of shiftconv_kernel.
#ifndef K0M_SKIP         // measurement builds (tools/k0_ab.py): 1 no window loads / LDS writes, 2 no B reads / MFMAs, 4 no stores -- wrong results
#define K0M_SKIP 0
#endif
template <int CH>
__global__ __launch_bounds__(CH * 8, 2) void shiftconv_mfma_kernel(const UnitK U, const XcdTiles G, const int nfr, const int per_xcd,
                                                              const int8_t* __restrict__ offs, const uint32_t* __restrict__ w1d, bf16_t* hw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RW = 34, PB = 72, PLANE = RW * PB, NTH = CH * 8, PCS = CH / 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, n = lane & 15;
    // A-fragment selectors: word e of a lane's fragment = elements j = 8 g + 2 e, + 1 of row m = n: weight index tx = j - m in 0..2, else zero.
    // v_perm_b32 (S0 = w[ty][0] | w[ty][1] << 16, S1 = w[ty][2]): selector bytes 4..7 = S0, 0..3 = S1, 0x0c = 0x00
    unsigned asel[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int tx0 = 8 * g + 2 * e - n, tx1 = tx0 + 1;
        const unsigned lo = tx0 == 0 ? 0x0504u : tx0 == 1 ? 0x0706u : tx0 == 2 ? 0x0100u : 0x0c0cu;
        const unsigned hi = tx1 == 0 ? 0x0504u : tx1 == 1 ? 0x0706u : tx1 == 2 ? 0x0100u : 0x0c0cu;
        asel[e] = lo | (hi << 16);
    }
    // B fragment of k-step ty, lane group g: window row n + ty + 8 + sy, columns 8 g + 8 + sx .. + 7; g = 2: columns 16, 17 only; g = 3: no such slots
    const uint4 bmask = make_uint4(g < 3 ? 0xffffffffu : 0u, g < 2 ? 0xffffffffu : 0u, g < 2 ? 0xffffffffu : 0u, g < 2 ? 0xffffffffu : 0u);
    const int bcol = (g < 3 ? g : 0) * 16;
    // A thread stages PAIRS of horizontally adjacent window pixels (rx even), one 8-channel piece of each: two 16-byte loads, then per channel ONE
    // 4-byte LDS store of (pixel rx, pixel rx + 1) -- v_perm_b32 of the two pixels' words -- instead of two 2-byte ones (half the LDS store
    // instructions of the first version: they were ~40 % of the loader's time).  The pairs of a thread do not depend on the tile: geo = ry << 8 | rx,
    // lad = LDS address in plane 8 pc (-1: beyond the window), gpix = pixel offset from the window's first pixel; the piece index pc is the same for
    // all of them (NTH is a multiple of PCS).
    static_assert(NTH % PCS == 0 && RW % 2 == 0, "a thread's pieces share the piece index; pixel pairs do not straddle rows");
    constexpr int NPAIR = RW * (RW / 2) * PCS, NITP = (NPAIR + NTH - 1) / NTH;
    int geo[NITP], lad[NITP], gpix[NITP];
    const int mypc = tid % PCS;
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
        const int idx = tid + k * NTH, idc = idx < NPAIR ? idx : 0;
        const int pr = idc / PCS;
        const int ry = pr / (RW / 2), rx = (pr - ry * (RW / 2)) * 2;
        geo[k] = ry << 8 | rx;
        lad[k] = idx < NPAIR ? (mypc * 8) * PLANE + mypc * 32 + ry * PB + rx * 2 : -1;
        gpix[k] = ry * U.w + rx;
    }
    const int xcd = (int)blockIdx.x & 7, j0 = (int)blockIdx.x >> 3, nwg = (int)gridDim.x >> 3;
    const int ntile_x = per_xcd * G.ntx;                      // tiles of this XCD's run of row-frames
    for (int i = j0; i < ntile_x; i += nwg) {
        const int rfl = i / G.ntx, tx_ = i - rfl * G.ntx;
        const int rf = xcd * per_xcd + rfl;
        if (rf >= G.nrf) break;                               // workgroup-uniform: the last XCD's run is shorter
        int tc = rf / G.nty;
        const int ty_ = rf - tc * G.nty;
        tc += U.t0;
        const int yc = ty_ * 16, xc = tx_ * 16;
        const SnSlabs<bf16_t> sl = unit_slabs(U, tc);
        const bf16_t* src = sl.pb;
        const int sstr = sl.sb;
        if (!(K0M_SKIP & 1)) {   // window -> planar LDS: ALL global loads first (one memory round trip per tile), then the LDS stores.
            // (Issuing the NEXT tile's loads before this tile's MFMAs -- 80 registers live across the arithmetic -- spilled and ran 419 instead of
            // 286 us at C = 64, 20 x 360 x 640; issuing them between the MFMAs and the stores, so that the counted vmcnt leaves the stores outstanding,
            // still spilled 37-92 registers in hipcc's allocation and ran 349 us.  Two workgroups per CU overlap the phases instead.  The ablation of this
            // version: loader alone 188 us, stores alone 87, together 277, everything 300: loader and stores of a wave serialise on the in-order vmcnt.)
            uint4 v0[NITP], v1[NITP];
            const bool wfull = yc >= 9 && xc >= 9 && yc + 25 <= U.h && xc + 25 <= U.w;      // workgroup-uniform: no test at all inside the image
            const bf16_t* wsrc = src + ((ptrdiff_t)(yc - 9) * U.w + (xc - 9)) * sstr + mypc * 8;
            if (wfull) {
#pragma unroll
                for (int k = 0; k < NITP; ++k) {
                    const bf16_t* q = wsrc + (ptrdiff_t)gpix[k] * sstr;
                    v0[k] = *(const uint4*)q; v1[k] = *(const uint4*)(q + sstr);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NITP; ++k) {
                    const int gy = yc - 9 + (geo[k] >> 8), gx = xc - 9 + (geo[k] & 255);
                    const bool iny = gy >= 0 && gy < U.h, in0 = iny && gx >= 0 && gx < U.w, in1 = iny && gx + 1 >= 0 && gx + 1 < U.w;
                    const bf16_t* q = wsrc + (ptrdiff_t)gpix[k] * sstr;
                    const uint4 a = *(const uint4*)(in0 ? q : src), b = *(const uint4*)(in1 ? q + sstr : src);
                    v0[k] = in0 ? a : make_uint4(0, 0, 0, 0); v1[k] = in1 ? b : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < NITP; ++k) {
                if (k == NITP - 1 && lad[k] < 0) continue;                // only the last pair of a thread can be beyond the window
                char* d = smem + lad[k];
                const uint4 a = v0[k], b = v1[k];
                *(uint32_t*)(d + 0 * PLANE) = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u); *(uint32_t*)(d + 1 * PLANE) = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
                *(uint32_t*)(d + 2 * PLANE) = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u); *(uint32_t*)(d + 3 * PLANE) = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
                *(uint32_t*)(d + 4 * PLANE) = __builtin_amdgcn_perm(b.z, a.z, 0x05040100u); *(uint32_t*)(d + 5 * PLANE) = __builtin_amdgcn_perm(b.z, a.z, 0x07060302u);
                *(uint32_t*)(d + 6 * PLANE) = __builtin_amdgcn_perm(b.w, a.w, 0x05040100u); *(uint32_t*)(d + 7 * PLANE) = __builtin_amdgcn_perm(b.w, a.w, 0x07060302u);
            }
        }
        __syncthreads();
        const bool interior = yc >= 1 && xc >= 1 && yc + 16 < U.h && xc + 16 < U.w;      // every tap of every output pixel inside the image
        uint4 tmask[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) tmask[s] = bmask;
        if (!interior) {            // conv zero padding: the UNSHIFTED position (yc + n + ty - 1, xc + j - 1) of a B element must be inside the image
            unsigned cw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qx0 = xc + g * 8 + 2 * e - 1, qx1 = qx0 + 1;
                cw[e] = ((qx0 >= 0 && qx0 < U.w) ? 0x0000ffffu : 0u) | ((qx1 >= 0 && qx1 < U.w) ? 0xffff0000u : 0u);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int qy = yc + n + s - 1;
                const unsigned rowin = (qy >= 0 && qy < U.h) ? 0xffffffffu : 0u;
                tmask[s].x &= cw[0] & rowin; tmask[s].y &= cw[1] & rowin; tmask[s].z &= cw[2] & rowin; tmask[s].w &= cw[3] & rowin;
            }
        }
        float res[8][4];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = wv * 8 + c;
            const int sy = offs[2 * k], sx = offs[2 * k + 1];                       // multiples of 4: the 16-byte reads below are 8-byte aligned
            const char* rb = smem + k * PLANE + wv * 32 + (n + 8 + sy) * PB + (8 + sx) * 2 + bcol;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < ((K0M_SKIP & 2) ? 0 : 3); ++s) {
                const unsigned w01 = w1d[k * 9 + s * 3] | (w1d[k * 9 + s * 3 + 1] << 16), w2z = w1d[k * 9 + s * 3 + 2];      // wave-uniform (bf16 in the low halves)
                const uint4 a = make_uint4(__builtin_amdgcn_perm(w01, w2z, asel[0]), __builtin_amdgcn_perm(w01, w2z, asel[1]),
                                           __builtin_amdgcn_perm(w01, w2z, asel[2]), __builtin_amdgcn_perm(w01, w2z, asel[3]));
                const uint2 b0 = *(const uint2*)(rb + s * PB), b1 = *(const uint2*)(rb + s * PB + 8);
                const uint4 b = make_uint4(b0.x & tmask[s].x, b0.y & tmask[s].y, b1.x & tmask[s].z, b1.y & tmask[s].w);
                acc = mfma16(as_frag(a), as_frag(b), acc);
            }
            res[c][0] = acc[0]; res[c][1] = acc[1]; res[c][2] = acc[2]; res[c][3] = acc[3];
        }
        // lane (g, n): out[yc + n][xc + 4 g + r] of the wave's 8 channels = one 16-byte piece per r
        // (through an LDS tile and out as whole pixel records: measured no faster at C = 64 -- 294 vs 296 us -- and slower at C = 80, two more barriers)
        const int oy = yc + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ox = xc + 4 * g + r;
            if (oy < U.h && ox < U.w && !(K0M_SKIP & 4)) {
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = res[c][r];
                *(uint4*)(hw + (((size_t)tc * U.h + oy) * U.w + ox) * CH + wv * 8) = pack8(o);
            }
        }
        __syncthreads();                                     // every wave is done with the window before the next one is written
    }
}

// K0 on the matrix cores, WALKING form: a workgroup takes a SEGMENT of a tile column (S vertically adjacent tiles of one frame) and keeps the planar
// window as a RING of 34 rows: the windows of vertically adjacent tiles share 18 of their 34 rows, so every tile after a segment's first stages 16 new
// rows instead of 34 -- the loader (what the tile form waits for: 188 of 280 us at C = 64, 20 x 360 x 640) does 2.1x less per tile.  Window row wr of the
// segment's tile jj lives in ring row (16 jj + wr) mod 34; rows are staged in blocks of up to 17 (one table per thread: pairs of pixels of a 17-row block;
// a segment's first window = two blocks, a later one = 16 rows of one).
// Everything else -- selectors, B fragments, masks, accumulator layout, stores -- is shiftconv_mfma_kernel's: results are bit-identical to it.
template <int CH>
__global__ __launch_bounds__(CH * 8, 2) void shiftconv_mfma_walk_kernel(const UnitK U, const int ntx, const int nty, const int S, const int nseg, const int nfr,
                                                                   const int per_xcd, const int8_t* __restrict__ offs, const uint32_t* __restrict__ w1d, bf16_t* hw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RW = 34, PB = 72, PLANE = RW * PB, NTH = CH * 8, PCS = CH / 8, BR = 17;      // blocks of 17 rows: a window is two of them
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, n = lane & 15;
    unsigned asel[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int tx0 = 8 * g + 2 * e - n, tx1 = tx0 + 1;
        const unsigned lo = tx0 == 0 ? 0x0504u : tx0 == 1 ? 0x0706u : tx0 == 2 ? 0x0100u : 0x0c0cu;
        const unsigned hi = tx1 == 0 ? 0x0504u : tx1 == 1 ? 0x0706u : tx1 == 2 ? 0x0100u : 0x0c0cu;
        asel[e] = lo | (hi << 16);
    }
    const uint4 bmask = make_uint4(g < 3 ? 0xffffffffu : 0u, g < 2 ? 0xffffffffu : 0u, g < 2 ? 0xffffffffu : 0u, g < 2 ? 0xffffffffu : 0u);
    const int bcol = (g < 3 ? g : 0) * 16;
    // a thread's pixel pairs of a 16-row block: local row lr, column rx (even), pixel offset; the piece index is the same for all of them
    static_assert(NTH % PCS == 0, "a thread's pieces share the piece index");
    constexpr int NPB = BR * (RW / 2) * PCS, NB = (NPB + NTH - 1) / NTH;
    int geo[NB], gpix[NB];
    const int mypc = tid % PCS;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int idx = tid + k * NTH;
        const int pr = (idx < NPB ? idx : 0) / PCS;
        const int lr = idx < NPB ? pr / (RW / 2) : 255, rx = (pr % (RW / 2)) * 2;      // lr 255: no such pair
        geo[k] = lr << 8 | rx;
        gpix[k] = (idx < NPB ? pr / (RW / 2) : 0) * U.w + rx;
    }
    char* const myplane = smem + (mypc * 8) * PLANE + mypc * 32;
    // rows wr0 .. wr0 + nrows - 1 of the window at (y0, x0) -> ring rows (rbase + wr) mod 34: all their global loads first, then the LDS stores.
    // (Fetching the NEXT tile's 16 rows -- 40 registers -- right after this tile's window is complete and storing them to LDS after its MFMAs, so
    // that the round trip hides behind the arithmetic, spilled 35-47 registers in hipcc's allocation and ran 330 instead of 258 us at C = 64,
    // 20 x 360 x 640: the third prefetch ordering on this kernel that lost to the allocator.)
    auto stage = [&](const bf16_t* src, int sstr, int y0, int x0, int wr0, int nrows, int rbase) __attribute__((always_inline)) {
        if (K0M_SKIP & 1) return;
        uint4 v0[NB], v1[NB];
        const int gy0 = y0 - 9 + wr0, gx0 = x0 - 9;
        const bool full = gy0 >= 0 && gy0 + nrows <= U.h && gx0 >= 0 && gx0 + RW <= U.w;      // workgroup-uniform: every pixel of the block inside the image
        const bf16_t* wsrc = src + ((ptrdiff_t)gy0 * U.w + gx0) * sstr + mypc * 8;
        if (full) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const bf16_t* q = wsrc + (ptrdiff_t)((geo[k] >> 8) < nrows ? gpix[k] : 0) * sstr;      // (a row beyond the block: any valid address, never stored)
                v0[k] = *(const uint4*)q; v1[k] = *(const uint4*)(q + sstr);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int lr = geo[k] >> 8, gy = gy0 + lr, gx = gx0 + (geo[k] & 255);
                const bool iny = lr < nrows && gy >= 0 && gy < U.h, in0 = iny && gx >= 0 && gx < U.w, in1 = iny && gx + 1 >= 0 && gx + 1 < U.w;
                const bf16_t* q = wsrc + (ptrdiff_t)gpix[k] * sstr;
                const uint4 a = *(const uint4*)(in0 ? q : src), b = *(const uint4*)(in1 ? q + sstr : src);
                v0[k] = in0 ? a : make_uint4(0, 0, 0, 0); v1[k] = in1 ? b : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int lr = geo[k] >> 8;
            if (lr >= nrows) continue;
            int rr = rbase + wr0 + lr;
            rr -= rr >= RW ? RW : 0; rr -= rr >= RW ? RW : 0;
            char* d = myplane + rr * PB + (geo[k] & 255) * 2;
            const uint4 a = v0[k], b = v1[k];
            *(uint32_t*)(d + 0 * PLANE) = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u); *(uint32_t*)(d + 1 * PLANE) = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
            *(uint32_t*)(d + 2 * PLANE) = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u); *(uint32_t*)(d + 3 * PLANE) = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
            *(uint32_t*)(d + 4 * PLANE) = __builtin_amdgcn_perm(b.z, a.z, 0x05040100u); *(uint32_t*)(d + 5 * PLANE) = __builtin_amdgcn_perm(b.z, a.z, 0x07060302u);
            *(uint32_t*)(d + 6 * PLANE) = __builtin_amdgcn_perm(b.w, a.w, 0x05040100u); *(uint32_t*)(d + 7 * PLANE) = __builtin_amdgcn_perm(b.w, a.w, 0x07060302u);
        }
    };
    const int xcd = (int)blockIdx.x & 7, j0 = (int)blockIdx.x >> 3, nwg = (int)gridDim.x >> 3;
    // items of this XCD: (frame, segment) rows of the item list x tile columns, tile column fastest: x-neighbours run concurrently on one L2
    const int nitem_x = per_xcd * ntx, nrows_all = nfr * nseg;
    for (int i = j0; i < nitem_x; i += nwg) {
        const int rl = i / ntx, tx_ = i - rl * ntx, row = xcd * per_xcd + rl;
        if (row >= nrows_all) break;                          // workgroup-uniform
        int tc = row / nseg;
        const int sg = row - tc * nseg, ty0 = sg * S, ntile = min(S, nty - ty0);
        tc += U.t0;
        const SnSlabs<bf16_t> sl = unit_slabs(U, tc);
        const int xc = tx_ * 16;
        for (int jj = 0; jj < ntile; ++jj) {
            const int yc = (ty0 + jj) * 16;
            int rbase = (16 * jj) % RW;                       // ring row of this tile's window row 0
            // a segment's first window: two blocks of 17 rows; a later one: window rows 18 .. 33, the 16 rows the previous tile did not have
            // (one copy of the staging code: a loop the compiler must not unroll)
            const int nblk = jj == 0 ? 2 : 1;
#pragma unroll 1
            for (int b = 0; b < nblk; ++b) stage(sl.pb, sl.sb, yc, xc, jj == 0 ? b * BR : RW - 16, jj == 0 ? BR : 16, jj == 0 ? 0 : rbase);
            __syncthreads();
            const bool interior = yc >= 1 && xc >= 1 && yc + 16 < U.h && xc + 16 < U.w;
            uint4 tmask[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) tmask[s] = bmask;
            if (!interior) {
                unsigned cw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qx0 = xc + g * 8 + 2 * e - 1, qx1 = qx0 + 1;
                    cw[e] = ((qx0 >= 0 && qx0 < U.w) ? 0x0000ffffu : 0u) | ((qx1 >= 0 && qx1 < U.w) ? 0xffff0000u : 0u);
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int qy = yc + n + s - 1;
                    const unsigned rowin = (qy >= 0 && qy < U.h) ? 0xffffffffu : 0u;
                    tmask[s].x &= cw[0] & rowin; tmask[s].y &= cw[1] & rowin; tmask[s].z &= cw[2] & rowin; tmask[s].w &= cw[3] & rowin;
                }
            }
            float res[8][4];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = wv * 8 + c;
                const int sy = offs[2 * k], sx = offs[2 * k + 1];
                const char* cb = smem + k * PLANE + wv * 32 + (8 + sx) * 2 + bcol;
                const int r0 = rbase + n + 8 + sy;            // ring row of window row n + 8 + sy (k-step ty adds ty), < 2 * 34
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < ((K0M_SKIP & 2) ? 0 : 3); ++s) {
                    const unsigned w01 = w1d[k * 9 + s * 3] | (w1d[k * 9 + s * 3 + 1] << 16), w2z = w1d[k * 9 + s * 3 + 2];
                    const uint4 a = make_uint4(__builtin_amdgcn_perm(w01, w2z, asel[0]), __builtin_amdgcn_perm(w01, w2z, asel[1]),
                                               __builtin_amdgcn_perm(w01, w2z, asel[2]), __builtin_amdgcn_perm(w01, w2z, asel[3]));
                    int rr = r0 + s;
                    rr -= rr >= RW ? RW : 0;
                    const char* rb = cb + rr * PB;
                    const uint2 b0 = *(const uint2*)rb, b1 = *(const uint2*)(rb + 8);
                    const uint4 b = make_uint4(b0.x & tmask[s].x, b0.y & tmask[s].y, b1.x & tmask[s].z, b1.y & tmask[s].w);
                    acc = mfma16(as_frag(a), as_frag(b), acc);
                }
                res[c][0] = acc[0]; res[c][1] = acc[1]; res[c][2] = acc[2]; res[c][3] = acc[3];
            }
            const int oy = yc + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = xc + 4 * g + r;
                if (oy < U.h && ox < U.w && !(K0M_SKIP & 4)) {
                    float o[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] = res[c][r];
                    *(uint4*)(hw + (((size_t)tc * U.h + oy) * U.w + ox) * CH + wv * 8) = pack8(o);
                }
            }
            __syncthreads();                                 // every wave is done with the ring rows the next tile overwrites
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K4: y = shortcut + W3' . (ca * g2) (+ bias'), beta folded into W3'/bias'; shortcut = rolled x (CAB2) or x (CAB1)
// waves-per-SIMD target: see the register-budget note in sn_conv.hip (91 VGPRs + 80 AGPRs = 2 waves without it; 148 / 152 = 3 waves)
// NT = N-tiles (16 pixels) per wave = 4: 148 / 152 registers, 3 waves per SIMD, the shortcut is loaded after the MFMAs.  (NT = 2 with the
// shortcut prefetched, 6 / 4 waves per SIMD, measured 22.6 vs 20.5 ms per window for C = 64 and 69.0 vs 71.2 ms for C = 80 in round 2.)
// Round 5: the shortcut operand by LDS-DMA (global_load_lds_dwordx4, no VGPRs) issued BEFORE the g2 loads, so that a wave waits for memory once per
// chunk instead of twice (32 KB of LDS per workgroup, still three workgroups per CU, 150 VGPRs): 353.3 / 370.3 us against 354.2 / 368.5 us for
// this kernel (CAB1 / CAB2, C = 64, 20 x 360 x 640, interleaved on one device) -- no difference: the kernel streams at its rate of 5.0 - 5.2 TB/s
// whether the second latency is exposed or not.  Not kept.  NT = 2 WITHOUT the prefetch (88 VGPRs, five waves per SIMD instead of three, twice the
// weight-fragment fetches per pixel): 363.2 / 389.5 us against 352.4 / 369.1 us (C = 64), 1279.8 / 1368.7 against 1206.6 / 1280.7 (C = 80): slower.
#ifndef SN_K4_NT         // (measurement builds: -DSN_K4_NT=2)
#define SN_K4_NT 4
#endif
template <int C, int NT>
__global__ __launch_bounds__(256, NT == 4 ? 3 : 5)
void scale_gemm_res_kernel(const UnitK U, const bf16_t* __restrict__ g2, const float* __restrict__ ca,
                           const uint4* __restrict__ wfrag, const float* __restrict__ bias, bf16_t* y, const int nchunk, const int nfr) {
    constexpr int CH = C / 2, KS = (C + 31) / 32, MT = C / 16;
    const int lane = threadIdx.x & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    // Workgroup -> (pixel chunk, frame) with the FRAMES of one chunk back to back on ONE XCD (workgroup b runs on XCD b % 8): the rolled shortcut
    // of a CAB2 is the upper half-channels of frame t-1 and the lower ones of frame t, i.e. 64 / 80-byte halves of 128-byte lines whose other half
    // is read by the workgroup of the neighbouring frame -- with the natural (chunk fastest, frame slowest) order that workgroup ran a whole frame
    // later on whatever XCD, and every line came from HBM twice (PMC round 4: 1.17x the kernel's bytes; the kernel is HBM-bound).
    // (items = (chunk, frame), frame fastest; XCD k takes the k-th contiguous eighth of the item list: equal load)
    const int nitem = nchunk * nfr, per = (nitem + 7) >> 3, item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= nitem) return;
    const int chunk = item / nfr, t = U.t0 + (item - chunk * nfr), hw = U.h * U.w;
    const SnSlabs<bf16_t> sl = unit_slabs(U, t);
    const int ibase = chunk * (64 * NT) + wv * (16 * NT);
    const int c0 = g * 4 * MT;                   // lane (g,p) owns channels [c0, c0 + 4 MT): one contiguous 8*MT-byte run of the shortcut and of y
    const bf16_t* const sbase = c0 < CH ? sl.p0 + c0 : sl.p1 + c0 - CH;
    const int sstr = c0 < CH ? sl.s0 : sl.s1;                                  // pixel stride of this lane's half (C, or C/2 for a halo half-frame)
    bf16x8_t B[NT][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk0 = s * 32 + g * 8;
        float cs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] = kk0 < C ? ca[(size_t)t * C + kk0 + j] : 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int i = ibase + n * 16 + p, ii = i < hw ? i : hw - 1;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (kk0 < C) q = *(const uint4*)(g2 + ((size_t)t * hw + ii) * C + kk0);
            float v[8];
            unpack8(q, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= cs[j];
            B[n][s] = as_frag(pack8(v));
        }
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bb = *(const float4*)(bias + g * 4 * MT + m * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4_t){bb.x, bb.y, bb.z, bb.w};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf16x8_t a = as_frag(wfrag[(m * KS + s) * 64 + lane]);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = mfma16(a, B[n][s], acc[m][n]);
        }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int i = ibase + n * 16 + p;
        if (i >= hw) continue;
        uint32_t sc[2 * MT], o[2 * MT];
        const bf16_t* sp = sbase + (size_t)i * sstr;
#pragma unroll
        for (int m = 0; m + 1 < MT; m += 2) { const uint4 q = *(const uint4*)(sp + m * 4); sc[2 * m] = q.x; sc[2 * m + 1] = q.y; sc[2 * m + 2] = q.z; sc[2 * m + 3] = q.w; }
        if (MT & 1) { const uint2 q = *(const uint2*)(sp + (MT - 1) * 4); sc[2 * MT - 2] = q.x; sc[2 * MT - 1] = q.y; }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            o[2 * m] = pack_bf2(bf_lo(sc[2 * m]) + acc[m][n][0], bf_hi(sc[2 * m]) + acc[m][n][1]);
            o[2 * m + 1] = pack_bf2(bf_lo(sc[2 * m + 1]) + acc[m][n][2], bf_hi(sc[2 * m + 1]) + acc[m][n][3]);
        }
        bf16_t* yp = y + ((size_t)t * hw + i) * C + c0;
#pragma unroll
        for (int m = 0; m + 1 < MT; m += 2) *(uint4*)(yp + m * 4) = make_uint4(o[2 * m], o[2 * m + 1], o[2 * m + 2], o[2 * m + 3]);
        if (MT & 1) *(uint2*)(yp + (MT - 1) * 4) = make_uint2(o[2 * MT - 2], o[2 * MT - 1]);
    }
}

UnitK to_k(const sn_unit_src* s) {
    UnitK u; u.x = (const bf16_t*)s->x; u.halo = (const bf16_t*)s->halo; u.T = s->T; u.h = s->h; u.w = s->w; u.C = s->C; u.mode = s->mode; u.wrap = s->wrap;
    u.t0 = s->nt > 0 ? s->t0 : 0;
    return u;
}
bool unit_ok(const sn_unit_src* s) {
    return s && s->x && (s->C == 64 || s->C == 80) && s->T > 0 && s->h > 0 && s->w > 0 && s->mode >= 0 && s->mode <= 2 && s->wrap >= 0 &&
           s->wrap <= 2 && (s->wrap != 2 || s->mode == 0 || s->halo);
}

}  // namespace

extern "C" {

int sn_gsts_gather(const sn_unit_src* s, const int8_t* offs, void* u, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !offs || !u || s->mode == 0) return SN_EINVAL;
    SN_FRAME_RANGE(s, t0, nt);
    hipLaunchKernelGGL(gather_kernel, dim3(1024, nt), dim3(256), 0, (hipStream_t)stream, to_k(s), offs, (bf16_t*)u, s->C + s->C / 2);
    return sn_check_launch();
}

int sn_temporal_roll(const sn_unit_src* s, void* y, void* stream) {
    sn_clear_error();
    if (!s || !s->x || !y || y == s->x || (s->C & 1) || s->mode < 1 || s->mode > 2 || s->T < 1 || (s->wrap == 2 && !s->halo)) return SN_EINVAL;
    SN_FRAME_RANGE(s, t0, nt);
    hipLaunchKernelGGL(gather_kernel, dim3(1024, nt), dim3(256), 0, (hipStream_t)stream, to_k(s), (const int8_t*)nullptr, (bf16_t*)y, s->C);
    return sn_check_launch();
}

int sn_gsts_shiftconv(const sn_unit_src* s, const int8_t* offs, const uint32_t* w1, void* hw, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !offs || !w1 || !hw || s->mode == 0) return SN_EINVAL;
    SN_FRAME_RANGE(s, t0, nt);
    const XcdTiles G = sn_xcd_tiles((s->w + 15) / 16, (s->h + 15) / 16, nt);
    const dim3 grid = sn_xcd_grid(G);
    if (s->C == 64) {
        constexpr int PP = 4;
        const size_t lds = 34 * (34 * (PP * 16 + 4) + 56);           // rows padded to 16 (mod 32) dwords
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)shiftconv_kernel<32, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH;
        hipLaunchKernelGGL((shiftconv_kernel<32, PP>), grid, dim3(256), lds, (hipStream_t)stream, to_k(s), G, offs, w1, (bf16_t*)hw);
    } else {
        constexpr int PP = 5;
        const size_t lds = 34 * (34 * (PP * 16 + 4) + 24);
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)shiftconv_kernel<40, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH;
        hipLaunchKernelGGL((shiftconv_kernel<40, PP>), grid, dim3(256), lds, (hipStream_t)stream, to_k(s), G, offs, w1, (bf16_t*)hw);
    }
    return sn_check_launch();
}


}  // extern "C"

namespace {
int cab_phase2(const sn_unit_src* s, const void* g2, const float* ca, const void* wfrag, const float* bias, void* y, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !g2 || !ca || !wfrag || !y || y == s->x) return SN_EINVAL;
    const int npx = s->h * s->w;
    constexpr int PXWG = 64 * SN_K4_NT;
    SN_FRAME_RANGE(s, t0, nt);
    const int nchunk = (npx + PXWG - 1) / PXWG;
    dim3 grid((unsigned)(8 * ((nchunk * nt + 7) / 8)));
    if (s->C == 64) hipLaunchKernelGGL((scale_gemm_res_kernel<64, SN_K4_NT>), grid, dim3(256), 0, (hipStream_t)stream, to_k(s), (const bf16_t*)g2, ca, (const uint4*)wfrag, bias, (bf16_t*)y, nchunk, nt);
    else hipLaunchKernelGGL((scale_gemm_res_kernel<80, SN_K4_NT>), grid, dim3(256), 0, (hipStream_t)stream, to_k(s), (const bf16_t*)g2, ca, (const uint4*)wfrag, bias, (bf16_t*)y, nchunk, nt);
    return sn_check_launch();
}
}  // namespace

extern "C" {

// K0 on the matrix cores (shiftconv_mfma_kernel); operands as sn_gsts_shiftconv
int sn_gsts_shiftconv_mfma(const sn_unit_src* s, const int8_t* offs, const uint32_t* w1, void* hw, void* stream) {
    sn_clear_error();
    if (!unit_ok(s) || !offs || !w1 || !hw || s->mode == 0) return SN_EINVAL;
    SN_FRAME_RANGE(s, t0, nt);
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) {
        (void)hipGetLastError();
        return SN_ELAUNCH;
    }
#ifndef K0M_WALK         // 1: segments of tile columns with the window as a ring of rows (shiftconv_mfma_walk_kernel); 0: one window per tile
#define K0M_WALK 1
#endif
    if (K0M_WALK) {
        const int ntx = (s->w + 15) / 16, nty = (s->h + 15) / 16, CHw = s->C / 2;
        const int wgs_cu = s->C == 64 ? 2 : 1;
        const long slots = (long)wgs_cu * ncu;
        // segment length: the longest of 8, 6, 4, 3, 2 tiles (a segment's first tile stages 34 rows, the others 16) whose item count keeps the
        // workgroup slots busy: at least 4 items per slot, or whatever the launch has
        int S = 2;
        const int cand[5] = {8, 6, 4, 3, 2};
        for (int k = 0; k < 5; ++k) {
            const int Sk = cand[k] < nty ? cand[k] : nty, nsk = (nty + Sk - 1) / Sk;
            if ((long)nsk * ntx * nt >= 4 * slots || k == 4) { S = Sk; break; }
        }
        // short segments mostly pay first tiles: measured 67 vs 62 us at 20 x 180 x 320 (S = 2) -- those launches take the tile form below
        const bool walk = S >= 4;
        const int nseg = (nty + (S < 1 ? 1 : S) - 1) / (S < 1 ? 1 : S);
        const int rows = nt * nseg, per_x = (rows + 7) / 8;
        long wg = slots / 8;
        if (wg > (long)per_x * ntx) wg = (long)per_x * ntx;
        if (wg < 1) wg = 1;
        const size_t ldsw = (size_t)CHw * 34 * 72 + (CHw / 8) * 32 + 64;
        const dim3 gridw(8u * (unsigned)wg);
        if (!walk) {
        } else if (s->C == 64) {
            if (hipFuncSetAttribute((const void*)shiftconv_mfma_walk_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw) != hipSuccess) return SN_ELAUNCH;
            hipLaunchKernelGGL((shiftconv_mfma_walk_kernel<32>), gridw, dim3(256), ldsw, (hipStream_t)stream, to_k(s), ntx, nty, S, nseg, nt, per_x, offs, w1, (bf16_t*)hw);
        } else {
            if (hipFuncSetAttribute((const void*)shiftconv_mfma_walk_kernel<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw) != hipSuccess) return SN_ELAUNCH;
            hipLaunchKernelGGL((shiftconv_mfma_walk_kernel<40>), gridw, dim3(320), ldsw, (hipStream_t)stream, to_k(s), ntx, nty, S, nseg, nt, per_x, offs, w1, (bf16_t*)hw);
        }
        if (walk) return sn_check_launch();
    }
    XcdTiles G = sn_xcd_tiles((s->w + 15) / 16, (s->h + 15) / 16, nt);
    const int per_xcd = (G.nrf + 7) / 8;
    const int CH = s->C / 2;
    const size_t lds = (size_t)CH * 34 * 72 + (CH / 8) * 32 + 64;
#ifndef K0M_PERSIST      // measurement builds: 0 = one workgroup per tile (the dispatcher hands out an XCD's tiles in order)
#define K0M_PERSIST 1
#endif
    int wgs = (s->C == 64 ? 2 : 1) * ncu / 8;                                // persistent workgroups per XCD (LDS: 78 / 98 KB each)
    const long tiles_x = (long)per_xcd * G.ntx;
    // C = 64 on large maps: one workgroup per tile after all -- the dispatcher hands out an XCD's tiles in order, a compact frontier that keeps more
    // of the windows' overlap in L2 than 64 persistent workgroups drifting apart (read requests to the fabric 301 vs ~570 MB per launch at
    // 20 x 360 x 640; 284 vs 296 us).  Smaller maps and C = 80 (the per-workgroup tables rebuilt per tile cost more than the locality buys:
    // 70 vs 62 us at 20 x 180 x 320, 1320 vs 1121 us at C = 80) stay persistent.  Same results either way.
    const bool per_tile = s->C == 64 && tiles_x >= 1500;
    if (wgs > tiles_x || !K0M_PERSIST || per_tile) wgs = (int)tiles_x;
    if (wgs < 1) wgs = 1;
    const dim3 grid(8u * (unsigned)wgs);
    if (s->C == 64) {
        if (hipFuncSetAttribute((const void*)shiftconv_mfma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH;
        hipLaunchKernelGGL((shiftconv_mfma_kernel<32>), grid, dim3(256), lds, (hipStream_t)stream, to_k(s), G, nt, per_xcd, offs, w1, (bf16_t*)hw);
    } else {
        if (hipFuncSetAttribute((const void*)shiftconv_mfma_kernel<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SN_ELAUNCH;
        hipLaunchKernelGGL((shiftconv_mfma_kernel<40>), grid, dim3(320), lds, (hipStream_t)stream, to_k(s), G, nt, per_xcd, offs, w1, (bf16_t*)hw);
    }
    return sn_check_launch();
}

int sn_gsts_cab2_phase2(const sn_unit_src* s, const void* g2, const float* ca, const void* wfrag, const float* bias, void* y, void* stream) {
    if (!s || (s->mode != 1 && s->mode != 2)) return SN_EINVAL;
    return cab_phase2(s, g2, ca, wfrag, bias, y, stream);
}

int sn_cab1_phase2(const sn_unit_src* s, const void* g2, const float* ca, const void* wfrag, const float* bias, void* y, void* stream) {
    if (!s || s->mode != 0) return SN_EINVAL;
    return cab_phase2(s, g2, ca, wfrag, bias, y, stream);
}

}  // extern "C"
