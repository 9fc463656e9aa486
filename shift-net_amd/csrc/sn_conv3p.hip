// Streaming form of the workhorse 3x3 convolution (stride 1, pad 1, ONE NHWC input, NHWC output: both convs of every CAB, conv_trans).
// gfx950 only.
//
// conv3_fast_kernel (sn_conv.hip) launches one workgroup per 8 x 32 tile: load -> barrier -> MFMA -> store, latency hidden by 6-8 resident
// workgroups only, and ~700 executed instructions per wave for its 20-56 MFMAs (address arithmetic, bounds masks, tap-offset selects: counted
// on the ISA).  Round 6 measured that a pass of it that stores NOTHING (the statistics pass of the fused CAB) takes 270 us where the full conv
// takes 295: the kernel is bound by its own instruction stream and by the exposed latency chain of each tile, not by HBM.
//
// Here the workgroups are PERSISTENT (one chunk of the tile list each, the list ordered frame > tile column > tile row, so a workgroup walks
// DOWN a column and the two halo rows it shares with its previous tile are L2 hits) and ROLE-SPLIT:
//   * wave 4, the LOADER, moves the (TH+2) x 34 pixel region of tile i+2 HBM -> LDS with LDS-DMA
//     (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write pass) while tiles i and i+1 are being computed / are in flight; zero padding is
//     the buffer descriptor's range check (rows above / below the frame are out of range and read as 0) plus one select per piece on the
//     left / right tile columns; it never stores, so its counted vmcnt sees loads only (in-order);
//   * waves 0-3 COMPUTE: weights stay in registers for the whole chunk, B fragments are ds_read_b128 at precomputed per-lane addresses
//     (one VGPR per k-step, immediates per N-tile), stores go out through a scalar base + per-lane offset.  The residual operand of tile
//     i + 1 is fetched into registers during tile i (range-checked buffer loads: 8-16 bytes per lane and N-tile); those are the only loads of
//     a compute wave, older than the stores behind them, so hipcc's counted vmcnt waits for them without draining the stores.
//   One s_barrier per tile: loader "tile i landed", compute waves "tile i-1 consumed" (its buffer is the one tile i+2 goes into).
// Operand layouts, k-slot order, accumulation order and every rounding are conv3_fast_kernel's: results are bit-identical to it.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

struct C3P {
    const bf16_t* in; const bf16_t* res; bf16_t* out;
    const uint4* wfrag; const float* bias; const float* oscale; int oscale_stride;
    const uint4* w2; const float* b2;      // cabp_kernel: the CAB's second conv
    float* pool; int pool_rows;
    int act; float prelu;
    int T, h, w, ntx, nty;
    int dbg;                     // measurements only (sn_conv_desc.flags bits 12..14): 1 no DMA, 2 no B reads / MFMAs, 4 no stores -- wrong results
    int lines_len;               // MODE 3: `out` is the border-line buffer [T][4][lines_len][CS]
    int S, nseg, nsg, qs;        // a tile column is cut into nseg segments of S tiles (the last one shorter); nsg segments in all; qs per chunk (workgroup)
};

// v_max_f32 without the canonicalising v_max per operand that fmaxf adds (device-only helper: inline asm inside a __global__ body poisons
// the host-side stub of the kernel)
__device__ __forceinline__ float max_bare(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// round(a * b) + c with BOTH roundings (hipcc contracts the source form -- and __fmul_rn / __fadd_rn -- to one fma)
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
    float r;
    asm("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 64 lanes x 16 bytes, global (range-checked buffer) -> LDS at the wave-uniform address dst + lane * 16.  A device-only helper: the builtin (and its
// address-space cast) written more than once inside the __global__ body itself makes hipcc 7.2 silently drop the kernel's HOST stub.
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rs, char* dst, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, voffset, 0, 0, 0);
}

// pin(x): an empty asm statement that reads and "rewrites" a register value.  hipcc must have the value in registers there -- it waits for
// the load that produces it and can neither sink that load nor re-issue it later.  The compute waves pin every value they load from global
// memory (weights, bias, CALayer scale) BEFORE the tile loop: a load still pending at the loop head gives the first use of its register inside
// the loop a counted vmcnt that, in the steady state, waits for the previous tile's STORES (measured on the fused CAB kernel: 353 -> 513 us
// when its weights moved to registers; __builtin_amdgcn_s_waitcnt alone does not help, the loads are sunk past it).
template <typename V> __device__ __forceinline__ void pin(V& x) { asm volatile("" : "+v"(x)); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const bf16_t* base, int frame_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, frame_bytes, 0x00020000);      // raw buffer: offsets >= num_records read as 0
}

constexpr int c3p_waves(int mt) { return mt == 1 ? 4 : mt == 2 ? 3 : 2; }      // waves per SIMD (5-wave workgroups: k per CU need ceil(5 k / 4))

// D: tiles in flight ahead of the one being computed (D + 1 LDS buffers).  MODE: the epilogue, fixed at compile time (no wave-uniform branches
// per N-tile): 0 bias only (conv_trans); 1 PReLU with a slope in [0, 1] + per-wave channel sums (first conv of a CAB); 2 CALayer scale +
// residual (its second conv); 3 = 1 without the store: only the border lines of the result (statistics pass of the fused CAB).  Any other
// combination stays on the tile kernel.
// RL (MODE 2): the residual operand's tile is staged in LDS by the loader as well (no load at all in the compute waves); otherwise the compute
// waves fetch it into registers one tile ahead.
template <int MT, int CS, int TH, int D, int MODE, bool RL = false>
__global__ __launch_bounds__(320, c3p_waves(MT)) void conv3p_kernel(const C3P P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 32, RH = TH + 2, RW = TW + 2, NPB = CS / 8, PSB = CS * 2;
    constexpr int KTOT = 9 * CS, KS = (KTOT + 31) / 32;
    constexpr int ROWP = RW * NPB, NITEM = RH * ROWP, NDMA = (NITEM + 63) / 64, XBUF = NDMA * 1024;
    static_assert(!RL || MODE == 2, "residual staging belongs to MODE 2");
    constexpr int RDMA = RL ? (TH * TW * NPB) / 64 : 0;             // residual tile, [row][pixel][CS] linear: a tile row is one contiguous run of 32 * CS * 2 bytes
    constexpr int NBUF = D + 1, BUFSZ = XBUF + RDMA * 1024, NL = NDMA + RDMA;
    constexpr bool WLDS = MT * KS > 16;                            // weights: registers up to 64 per lane, else staged once into LDS (1 KB per fragment)
    constexpr int WBYTES = WLDS ? MT * KS * 1024 : 0;
    constexpr int RPW = TH / 4, NTW = RPW * 2;                     // rows / N-tiles (16 pixels) per compute wave
    constexpr int NH = MT == 1 ? (NTW < 4 ? NTW : 4) : 2;          // N-tiles per accumulation pass
    static_assert(NTW % NH == 0 && NTW <= 4 * NH, "passes");
    static_assert(TH % 4 == 0 && NL * (D - 1) <= 63 && D >= 1, "tile / prefetch shape");
    const int lane = threadIdx.x & 63, wv = wave_id();
    // chunk of the tile list; XCD k (= blockIdx.x & 7 as dispatched) takes the k-th contiguous eighth of the chunks
    const int G = (int)gridDim.x, c = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    // The work list is (frame, tile column, segment of the column) in that order; a chunk is qs consecutive segments, so its tiles are consecutive
    // in the (frame > column > row) order and every segment -- the unit the channel sums are flushed in -- belongs to exactly one workgroup.
    const int fseg = c * P.qs, nsegs = min(P.qs, P.nsg - fseg);
    if (nsegs <= 0) return;                                         // workgroup-uniform
    int t = fseg / (P.ntx * P.nseg), tx, ty, n = 0;
    {
        const int rem = fseg - t * (P.ntx * P.nseg);
        tx = rem / P.nseg;
        const int sg = rem - tx * P.nseg;
        ty = sg * P.S;
        const int tail = P.nty - (P.nseg - 1) * P.S;                // tiles of a column's last segment
        for (int k = 0, g2 = sg; k < nsegs; ++k) { n += g2 == P.nseg - 1 ? tail : P.S; if (++g2 == P.nseg) g2 = 0; }
    }
    n = __builtin_amdgcn_readfirstlane(n);
    t = __builtin_amdgcn_readfirstlane(t); tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty);
    const int rowpitch = P.w * PSB, frame_bytes = P.h * rowpitch;
    const size_t frame_elems = (size_t)P.h * P.w * CS;
    char* const xbufs = smem + WBYTES;
    if constexpr (WLDS) {                                           // all five waves, before the roles split: no DMA is in flight yet
        for (int i = (int)threadIdx.x; i < MT * KS * 64; i += 320) *(uint4*)(smem + i * 16) = P.wfrag[i];
        __syncthreads();
    }

    if (wv == 4) {
        // =================================================== LOADER ===================================================
        int voff[NDMA], pxk[NDMA];
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int i = k * 64 + lane, ic = i < NITEM ? i : NITEM - 1;
            const int r = ic / ROWP, j = ic - r * ROWP;
            voff[k] = r * rowpitch + j * 16;
            pxk[k] = j / NPB;
        }
        int rvoff[RL ? RDMA : 1];
        if constexpr (RL) {
#pragma unroll
            for (int k = 0; k < RDMA; ++k) {
                const int i = k * 64 + lane, row = i / (TW * NPB), rem = i - row * (TW * NPB);
                rvoff[k] = row * rowpitch + rem * 16;
            }
        }
        // tiles first .. first + D - 1 are requested before the first barrier, tile i + D right after barrier i (into the buffer tile i - 1 was computed from)
        int ft = t, ftx = tx, fty = ty;                             // the tile the next request fetches
        for (int i = -D; i < n; ++i) {
            if (i >= 0) {
                if (n - 1 - i >= D - 1) wait_vmcnt<NL * (D - 1)>(); else wait_vmcnt<0>();      // tile i has landed (the newer ones may be in flight)
                __builtin_amdgcn_s_barrier();
            }
            const int j = i + D;
            if (j < n && !(P.dbg & 1)) {
                const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P.in + (size_t)ft * frame_elems, frame_bytes);
                const int ix0 = ftx * TW - 1;
                const int o = ((fty * TH - 1) * P.w + ix0) * PSB;                   // may be negative: 32-bit wrap -> out of range -> 0
                const bool edge = ftx == 0 || ix0 + RW > P.w;                       // wave-uniform
                char* const dst = xbufs + (j % NBUF) * BUFSZ;
#pragma unroll
                for (int k = 0; k < NDMA; ++k) {
                    int vo = voff[k] + o;
                    if (edge) { const int gx = ix0 + pxk[k]; vo = (gx < 0 || gx >= P.w) ? (int)0x80000000 : vo; }      // left / right of the image: reads 0
                    dma16(rs, dst + k * 1024, vo);
                }
                if constexpr (RL) {                                                 // rows below the frame read 0; columns right of it land in pixels nobody stores
                    const __amdgpu_buffer_rsrc_t rr = frame_rsrc(P.res + (size_t)ft * frame_elems, frame_bytes);
                    const int o2 = ((fty * TH) * P.w + ftx * TW) * PSB;
#pragma unroll
                    for (int k = 0; k < RDMA; ++k)
                        dma16(rr, dst + XBUF + k * 1024, rvoff[k] + o2);
                }
                if (++fty == P.nty) { fty = 0; if (++ftx == P.ntx) { ftx = 0; ++ft; } }
            }
        }
        return;
    }

    // ====================================================== COMPUTE ======================================================
    const int g = lane >> 4, p = lane & 15;
    const int c0 = g * 4 * MT;
    // per-lane LDS address of the B operand of k-step s for the wave's first N-tile (row wv*RPW, pixels 0..15): lane group g reads
    // k-slots [(4s+g)*8, +8) = 8 channels starting at cc0 of tap (dy,dx)
    int addr_s[KS];
    const int lane_base = ((wv * RPW) * RW + p) * PSB;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        int toff = 0;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int kk0 = (s * 4 + gg) * 8;
            const int tap = kk0 / CS, cc0 = kk0 - tap * CS, dy = tap / 3, dx = tap - dy * 3;
            const int o = kk0 < KTOT ? (dy * RW + dx) * PSB + cc0 * 2 : 0;
            toff = g == gg ? o : toff;
        }
        addr_s[s] = lane_base + toff;
    }
    bf16x8_t A[WLDS ? 1 : MT][WLDS ? 1 : KS];
    if constexpr (!WLDS) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < KS; ++s) A[m][s] = as_frag(P.wfrag[(m * KS + s) * 64 + lane]);
    }
    const char* const wl = smem + lane * 16;                         // WLDS: fragment (m, s) at wl + (m * KS + s) * 1024
    f32x4_t biasv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float4 b4 = P.bias ? *(const float4*)(P.bias + c0 + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        biasv[m] = (f32x4_t){b4.x, b4.y, b4.z, b4.w};
    }
    const float slope = P.prelu;
    float4 osc[MT];
    auto load_osc = [&](int ft) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int m = 0; m < MT; ++m) osc[m] = *(const float4*)(P.oscale + (size_t)ft * P.oscale_stride + c0 + m * 4);
        }
    };
    load_osc(t);
    auto pin_loaded = [&]() __attribute__((always_inline)) {
        if constexpr (!WLDS) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int s = 0; s < KS; ++s) pin(A[m][s]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            pin(biasv[m]);
            if constexpr (MODE == 2) { pin(osc[m].x); pin(osc[m].y); pin(osc[m].z); pin(osc[m].w); }
        }
    };
    pin_loaded();
    // Byte offset of (wave row rr, pixel p, channel c0) from the tile's first output pixel.  Stores and residual loads go through a
    // range-checked buffer descriptor of the frame: an offset beyond it is dropped / reads 0, so rows below the frame need no mask, and a
    // lane whose channels are all padding (24 channels in 32 rows: lane group 3) carries an out-of-range offset for good.
    constexpr int OOR = (int)0x80000000;
    constexpr int PIECE = (MT == 2 || MT == 4) ? 16 : 8;             // bytes per store / residual piece; NP pieces per lane and N-tile
    constexpr int NP = (MT * 8) / PIECE;
    typedef unsigned rword_t __attribute__((ext_vector_type(PIECE / 4)));
    int ooff[RPW][NP];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int q = 0; q < NP; ++q)
            ooff[rr][q] = (c0 + q * (PIECE / 2) < CS) ? ((wv * RPW + rr) * P.w + p) * PSB + c0 * 2 + q * PIECE : OOR;
    // The residual of tile i + 1 is requested BEFORE tile i's MFMAs and stores: the loads are then older than the stores issued behind them, and
    // hipcc's counted vmcnt waits for them without draining those stores.
    // (two named sets used alternately by a tile loop unrolled twice: with one set plus a copy per tile hipcc hoists the copy -- and with it the
    // wait for loads issued a moment ago -- into the MFMA phase)
    typedef rword_t rset_t[(MODE == 2 && !RL) ? NTW : 1][(MODE == 2 && !RL) ? NP : 1];
    rset_t rA, rB;
    auto load_res = [&](rset_t& rnxt, int ft, int ftx, int fty) __attribute__((always_inline)) {
        if constexpr (MODE == 2 && !RL) {
            const __amdgpu_buffer_rsrc_t rr = frame_rsrc(P.res + (size_t)ft * frame_elems, frame_bytes);
            const int o2 = ((fty * TH) * P.w + ftx * TW) * PSB;
#pragma unroll
            for (int nn = 0; nn < NTW; ++nn)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int vo = o2 + ooff[nn >> 1][q] + (nn & 1) * 16 * PSB;
                    if constexpr (PIECE == 16) rnxt[nn][q] = __builtin_amdgcn_raw_buffer_load_b128(rr, vo, 0, 0);
                    else rnxt[nn][q] = __builtin_amdgcn_raw_buffer_load_b64(rr, vo, 0, 0);
                }
        }
    };
    load_res(rA, t, tx, ty);
    float psum[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) psum[m][r] = 0.f;
    // this wave's channel sums over ONE segment -> pool row (column, segment, wave) of the frame: the rows are a property of the image (not of
    // the frame's position in the window, the chunking or the device), so equal frames give bit-equal sums wherever they sit
    auto flush_pool = [&](int ft, int ftx, int fsg) {
        if constexpr (MODE == 1 || MODE == 3) {
            const int row = (ftx * P.nseg + fsg) * 4 + wv;
            float* dst = P.pool + ((size_t)ft * P.pool_rows + row) * (16 * MT);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sm = row_sum16(psum[m][r]);
                    if (p == 0) dst[c0 + m * 4 + r] = sm;
                    psum[m][r] = 0.f;
                }
        }
    };

    // one N-tile of the epilogue; MASKED: the tile crosses the right / bottom image border (pixels beyond it must reach neither a store nor the sums)
    // (plain lambdas called with literal arguments and force-inlined: generic lambdas in a __global__ template lose the host-side stub)
    const int raddr = XBUF + ((wv * RPW) * TW + p) * PSB + c0 * 2;   // RL: this lane's channels of pixel p of the wave's first row in the staged residual tile
    auto finish = [&](const f32x4_t (&acc)[MT][NH], const rset_t& rres, const char* xs, const int N0, const __amdgpu_buffer_rsrc_t ro, int o2, const bool MASKED,
                      int ylim, int xlim) __attribute__((always_inline)) {
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            const int nn = N0 + nh, rr = nn >> 1, xb = nn & 1;
            float v[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                v[m][0] = acc[m][nh][0]; v[m][1] = acc[m][nh][1]; v[m][2] = acc[m][nh][2]; v[m][3] = acc[m][nh][3];
                if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                    // max(x, a x), a in [0, 1]; the bare instruction: fmaxf adds a canonicalising v_max per operand
                        v[m][r] = max_bare(v[m][r], slope * v[m][r]);
                    }
                }
            }
            if constexpr (MODE == 2) {      // round(conv * scale) + x: the tile kernel's two steps sit in different blocks and are NOT contracted to an fma
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    rword_t qv;
                    if constexpr (RL) qv = *(const rword_t*)(xs + raddr + (rr * TW + xb * 16) * PSB + ((m * 8) / PIECE) * PIECE);
                    else qv = rres[nn][(m * 8) / PIECE];
                    const unsigned lo = qv[((m * 8) % PIECE) / 4], hi = qv[((m * 8) % PIECE) / 4 + 1];
                    v[m][0] = mul_then_add(v[m][0], osc[m].x, bf_lo(lo)); v[m][1] = mul_then_add(v[m][1], osc[m].y, bf_hi(lo));
                    v[m][2] = mul_then_add(v[m][2], osc[m].z, bf_lo(hi)); v[m][3] = mul_then_add(v[m][3], osc[m].w, bf_hi(hi));
                }
            }
            bool ok = true;
            if (MASKED) ok = (rr < ylim) && (xb * 16 < xlim);
            if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) psum[m][r] += (MASKED && !ok) ? 0.f : v[m][r];
            }
            if constexpr (MODE == 3) {
                // statistics pass of the fused CAB: nothing is stored but the first / last rows and columns of the result, bf16-rounded like the tensor
                // would be, into the line buffer P.out = [T][4][lines_len][CS] (row 0, row h-1, column 0, column w-1).  MASKED is set for every tile on
                // the image border (and for ragged ones).
                if (MASKED && ok && c0 < CS) {
                    const int oy = ty * TH + wv * RPW + rr, ox = tx * TW + xb * 16 + p;
                    bf16_t* const lb = P.out + (size_t)t * 4 * P.lines_len * CS + c0;
                    bf16_t* tgt[4] = {oy == 0 ? lb + (size_t)ox * CS : nullptr, oy == P.h - 1 ? lb + ((size_t)P.lines_len + ox) * CS : nullptr,
                                      ox == 0 ? lb + ((size_t)2 * P.lines_len + oy) * CS : nullptr, ox == P.w - 1 ? lb + ((size_t)3 * P.lines_len + oy) * CS : nullptr};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!tgt[e]) continue;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            if (c0 + m * 4 < CS) *(uint2*)(tgt[e] + m * 4) = make_uint2(pack_bf2(v[m][0], v[m][1]), pack_bf2(v[m][2], v[m][3]));
                    }
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                int vo = o2 + ooff[rr][q] + xb * 16 * PSB;
                if (MASKED) vo = ok ? vo : OOR;
                if (P.dbg & 4) vo = OOR;
                if constexpr (PIECE == 16) {
                    const int m = q * 2;
                    const rword_t d = {pack_bf2(v[m][0], v[m][1]), pack_bf2(v[m][2], v[m][3]), pack_bf2(v[m + 1][0], v[m + 1][1]), pack_bf2(v[m + 1][2], v[m + 1][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(d, ro, vo, 0, 0);
                } else {
                    const rword_t d = {pack_bf2(v[q][0], v[q][1]), pack_bf2(v[q][2], v[q][3])};
                    __builtin_amdgcn_raw_buffer_store_b64(d, ro, vo, 0, 0);
                }
            }
        }
    };

    int sgi = ty / P.S, sleft = min(P.S, P.nty - ty);                // current segment of the column and the tiles left in it (a chunk starts at a segment)
    int t2 = t, tx2 = tx, ty2 = ty;                                  // the tile after the current one
    auto advance2 = [&]() { if (++ty2 == P.nty) { ty2 = 0; if (++tx2 == P.ntx) { tx2 = 0; ++t2; } } };
    advance2();
    auto tile = [&](const int i, const rset_t& rres, rset_t& rnxt) __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();                                // tile i is in LDS (the loader waited for its DMA before arriving here)
        const char* const xs = xbufs + (i % NBUF) * BUFSZ;
        if (i + 1 < n) load_res(rnxt, t2, tx2, ty2);                 // residual of tile i + 1: older than this tile's stores
        // B fragments one k-step ahead of the MFMAs that consume them (two register sets): the scheduler on its own keeps two fragments in
        // flight and waits for each before its MFMA -- an LDS round trip per pair, ten per tile.  NH N-tiles per pass (registers).
        const int oy0 = ty * TH, ox0 = tx * TW;
        const __amdgpu_buffer_rsrc_t ro = frame_rsrc(P.out + (size_t)t * frame_elems, frame_bytes);
        const int o2 = (oy0 * P.w + ox0) * PSB;
        const bool full = (oy0 + TH <= P.h) && (ox0 + TW <= P.w);    // wave-uniform
#pragma unroll
        for (int ps = 0; ps < NTW / NH; ++ps) {
            const int N0 = ps * NH;                                  // a constant after unrolling
            f32x4_t acc[MT][NH];
            bf16x8_t bq[2][NH], aq[2][WLDS ? MT : 1];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) acc[m][nh] = biasv[m];
            if (!(P.dbg & 2)) {
#pragma unroll
            for (int nh = 0; nh < NH; ++nh)
                bq[0][nh] = as_frag(*(const uint4*)(xs + addr_s[0] + (((N0 + nh) >> 1) * RW + ((N0 + nh) & 1) * 16) * PSB));
            if constexpr (WLDS) {
#pragma unroll
                for (int m = 0; m < MT; ++m) aq[0][m] = as_frag(*(const uint4*)(wl + (m * KS) * 1024));
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh)
                        bq[(s + 1) & 1][nh] = as_frag(*(const uint4*)(xs + addr_s[s + 1] + (((N0 + nh) >> 1) * RW + ((N0 + nh) & 1) * 16) * PSB));
                    if constexpr (WLDS) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) aq[(s + 1) & 1][m] = as_frag(*(const uint4*)(wl + (m * KS + s + 1) * 1024));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh)
                        acc[m][nh] = mfma16(WLDS ? aq[s & 1][m] : A[WLDS ? 0 : m][WLDS ? 0 : s], bq[s & 1][nh], s == 0 ? biasv[m] : acc[m][nh]);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // ---- epilogue (arithmetic of conv3_fast_kernel): PReLU + channel sums | CALayer scale + residual; store ----
            const bool plain = full && !(MODE == 3 && (ty == 0 || ty == P.nty - 1 || tx == 0 || tx == P.ntx - 1));      // wave-uniform
            if (plain) finish(acc, rres, xs, N0, ro, o2, false, 0, 0);
            else finish(acc, rres, xs, N0, ro, o2, true, P.h - oy0 - wv * RPW, P.w - ox0 - p);
        }
        // next tile of the chunk (down the column, then the next column, then the next frame)
        if (--sleft == 0) {                                          // the segment ends with this tile (after S tiles or with the column)
            flush_pool(t, tx, sgi);
            ++sgi;
            sleft = min(P.S, P.nty - ty - 1);
        }
        if (++ty == P.nty) {
            ty = 0; sgi = 0; sleft = min(P.S, P.nty);
            if (++tx == P.ntx) {
                tx = 0;
                ++t;
                if (i + 1 < n) {                                     // once per frame of a chunk: the wait drains this wave's stores too
                    load_osc(t);
                    if constexpr (MODE == 2) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) { pin(osc[m].x); pin(osc[m].y); pin(osc[m].z); pin(osc[m].w); }
                    }
                }
            }
        }
        advance2();
    };
    for (int i = 0; i < n; i += 2) {
        tile(i, rA, rB);
        if (i + 1 < n) tile(i + 1, rB, rA);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Fused dense CAB on the streaming structure (pass 2 of the fused CAB; pass 1 is conv3p_kernel<.., MODE 3>, the closed-form CALayer sits between):
//   out = x + ca * conv2(PReLU(conv1(x) + b1)) + b2-scaled terms exactly as the two-launch form computes them, with `mid` only ever in LDS.
// Per TH x 32 output tile the loader brings the (TH+4) x 36 region of x; the compute waves run conv1 + PReLU on the (TH+2) x 34 ring conv2 needs
// (16-pixel N-tiles over the ring's pixels in row-major order, zero outside the image = conv2's zero padding) into one bf16 `mid` tile, meet at
// a second barrier, and run conv2 from it with the scale / + x (the region's centre) / store epilogue of MODE 2.  Two barriers per tile: "region
// landed and mid free" and "mid written".  Same operand layouts, k order and roundings as conv3_fast_kernel twice: bit-identical to the two-launch
// form.  conv1 runs 1.33x (TH = 8) redundantly -- on matrix cores that the ablations of round 6 show idle (the kernel's time is its memory traffic).
template <int MT, int CS, int TH, int D, int WR>
__global__ __launch_bounds__(320, c3p_waves(MT)) void cabp_kernel(const C3P P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 32, RH = TH + 4, RW = TW + 4, MH = TH + 2, MW = TW + 2, NPB = CS / 8, PSB = CS * 2;
    constexpr int KTOT = 9 * CS, KS = (KTOT + 31) / 32;
    constexpr int ROWP = RW * NPB, NITEM = RH * ROWP, NDMA = (NITEM + 63) / 64, XBUF = NDMA * 1024;
    constexpr int NBUF = D + 1, NL = NDMA;
    // Weight fragments of both convs in LDS, 1 KB per fragment (WR: measurement variants with conv1's (1), conv2's (2) or both sets (3) in registers.
    // Both sets at 16 channels = 144 registers: 513 us against 353 at 20 x 720 x 1280 -- the kernel is built for two 5-wave workgroups per CU, and at
    // three waves per SIMD the second one only fits when the dispatcher's starting SIMD happens to suit; at <= 128 registers any start fits.
    // conv1's set alone spills at 128; conv2's alone: 126 registers, 327 us)
    constexpr bool W1L = !(WR & 1), W2L = !(WR & 2);
    constexpr int WSET = MT * KS * 1024, W2OFF = W1L ? WSET : 0, WBYTES = W2OFF + (W2L ? WSET : 0);
    constexpr int NM = MH * MW, NT1 = (NM + 15) / 16, PER1 = (NT1 + 3) / 4;      // conv1: N-tiles of the ring, per wave (interleaved)
    constexpr int NH1 = MT == 1 ? 3 : 2;                            // N-tiles per accumulation pass of conv1 / conv2
    constexpr int RPW = TH / 4, NTW = RPW * 2, NH = MT == 1 ? (NTW < 4 ? NTW : 4) : 2;
    constexpr int MIDB = (NM * PSB + 15) / 16 * 16;
    static_assert(TH % 4 == 0 && NL * (D - 1) <= 63 && D >= 1 && PER1 % NH1 == 0 && NTW % NH == 0, "tile / prefetch shape");
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int G = (int)gridDim.x, c = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int fseg = c * P.qs, nsegs = min(P.qs, P.nsg - fseg);
    if (nsegs <= 0) return;                                         // workgroup-uniform
    int t = fseg / (P.ntx * P.nseg), tx, ty, n = 0;
    {
        const int rem = fseg - t * (P.ntx * P.nseg);
        tx = rem / P.nseg;
        const int sg = rem - tx * P.nseg;
        ty = sg * P.S;
        const int tail = P.nty - (P.nseg - 1) * P.S;
        for (int k = 0, g2 = sg; k < nsegs; ++k) { n += g2 == P.nseg - 1 ? tail : P.S; if (++g2 == P.nseg) g2 = 0; }
    }
    n = __builtin_amdgcn_readfirstlane(n);
    t = __builtin_amdgcn_readfirstlane(t); tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty);
    const int rowpitch = P.w * PSB, frame_bytes = P.h * rowpitch;
    const size_t frame_elems = (size_t)P.h * P.w * CS;
    char* const mids = smem + WBYTES;
    char* const xbufs = mids + MIDB;
    if constexpr (W1L || W2L) {
        for (int i = (int)threadIdx.x; i < MT * KS * 64; i += 320) {
            if constexpr (W1L) *(uint4*)(smem + i * 16) = P.wfrag[i];
            if constexpr (W2L) *(uint4*)(smem + W2OFF + i * 16) = P.w2[i];
        }
        __syncthreads();
    }

    if (wv == 4) {
        // =================================================== LOADER ===================================================
        int voff[NDMA], pxk[NDMA];
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int i = k * 64 + lane, ic = i < NITEM ? i : NITEM - 1;
            const int r = ic / ROWP, j = ic - r * ROWP;
            voff[k] = r * rowpitch + j * 16;
            pxk[k] = j / NPB;
        }
        int ft = t, ftx = tx, fty = ty;
        for (int i = -D; i < n; ++i) {
            if (i >= 0) {
                if (n - 1 - i >= D - 1) wait_vmcnt<NL * (D - 1)>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();                        // region i landed; mid is free (every compute wave is through tile i - 1)
            }
            const int j = i + D;
            if (j < n) {
                const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P.in + (size_t)ft * frame_elems, frame_bytes);
                const int ix0 = ftx * TW - 2;
                const int o = ((fty * TH - 2) * P.w + ix0) * PSB;
                const bool edge = ftx == 0 || ix0 + RW > P.w;
                char* const dst = xbufs + (j % NBUF) * XBUF;
#pragma unroll
                for (int k = 0; k < NDMA; ++k) {
                    int vo = voff[k] + o;
                    if (edge) { const int gx = ix0 + pxk[k]; vo = (gx < 0 || gx >= P.w) ? (int)0x80000000 : vo; }
                    dma16(rs, dst + k * 1024, vo);
                }
                if (++fty == P.nty) { fty = 0; if (++ftx == P.ntx) { ftx = 0; ++ft; } }
            }
            if (i >= 0) __builtin_amdgcn_s_barrier();                // (the compute waves' "mid written" barrier)
        }
        return;
    }

    // ====================================================== COMPUTE ======================================================
    const int g = lane >> 4, p = lane & 15;
    const int c0 = g * 4 * MT;
    int toff1[KS], addr2[KS];
    const int lane_base2 = ((wv * RPW) * MW + p) * PSB;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        int o1 = 0, o2 = 0;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int kk0 = (s * 4 + gg) * 8;
            const int tap = kk0 / CS, cc0 = kk0 - tap * CS, dy = tap / 3, dx = tap - dy * 3;
            const int a = kk0 < KTOT ? (dy * RW + dx) * PSB + cc0 * 2 : 0, b = kk0 < KTOT ? (dy * MW + dx) * PSB + cc0 * 2 : 0;
            o1 = g == gg ? a : o1; o2 = g == gg ? b : o2;
        }
        toff1[s] = o1;
        addr2[s] = lane_base2 + o2;
    }
    // conv1: this lane's pixel of each of the wave's ring N-tiles (tile-independent): region offset, mid offset, ring coordinates
    int pix1[PER1], mwr[PER1];
#pragma unroll
    for (int i = 0; i < PER1; ++i) {
        const int q = (wv + 4 * i) * 16 + p, qc = q < NM ? q : NM - 1;
        const int my = qc / MW, mx = qc - my * MW;
        pix1[i] = (my * RW + mx) * PSB;
        mwr[i] = (q < NM && c0 < CS) ? qc * PSB + c0 * 2 : -1;      // -1: nothing to write (beyond the ring, or a lane whose channels are all padding)
    }
    bf16x8_t A1[W1L ? 1 : MT][W1L ? 1 : KS], A2[W2L ? 1 : MT][W2L ? 1 : KS];
    if constexpr (!W1L) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < KS; ++s) A1[m][s] = as_frag(P.wfrag[(m * KS + s) * 64 + lane]);
    }
    if constexpr (!W2L) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < KS; ++s) A2[m][s] = as_frag(P.w2[(m * KS + s) * 64 + lane]);
    }
    const char* const wl = smem + lane * 16;
    f32x4_t bias1[MT], bias2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float4 a4 = P.bias ? *(const float4*)(P.bias + c0 + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b4 = P.b2 ? *(const float4*)(P.b2 + c0 + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bias1[m] = (f32x4_t){a4.x, a4.y, a4.z, a4.w};
        bias2[m] = (f32x4_t){b4.x, b4.y, b4.z, b4.w};
    }
    const float slope = P.prelu;
    float4 osc[MT];
    auto load_osc = [&](int ft) {
#pragma unroll
        for (int m = 0; m < MT; ++m) osc[m] = *(const float4*)(P.oscale + (size_t)ft * P.oscale_stride + c0 + m * 4);
    };
    load_osc(t);
    auto pin_osc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { pin(osc[m].x); pin(osc[m].y); pin(osc[m].z); pin(osc[m].w); }
    };
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        pin(bias1[m]); pin(bias2[m]);
        if constexpr (!W1L) {
#pragma unroll
            for (int s = 0; s < KS; ++s) pin(A1[m][s]);
        }
        if constexpr (!W2L) {
#pragma unroll
            for (int s = 0; s < KS; ++s) pin(A2[m][s]);
        }
    }
    pin_osc();
    constexpr int OOR = (int)0x80000000;
    constexpr int PIECE = (MT == 2 || MT == 4) ? 16 : 8, NP = (MT * 8) / PIECE;
    typedef unsigned rword_t __attribute__((ext_vector_type(PIECE / 4)));
    int ooff[RPW][NP];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int q = 0; q < NP; ++q)
            ooff[rr][q] = (c0 + q * (PIECE / 2) < CS) ? ((wv * RPW + rr) * P.w + p) * PSB + c0 * 2 + q * PIECE : OOR;
    const int raddr = ((wv * RPW + 2) * RW + p + 2) * PSB + c0 * 2;  // x at this lane's output pixel: the region's centre

    for (int i = 0; i < n; ++i) {
        __builtin_amdgcn_s_barrier();                                // region i is in LDS; mid is free
        const char* const xs = xbufs + (i % NBUF) * XBUF;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const bool inner = oy0 >= 1 && oy0 + TH + 1 <= P.h && ox0 >= 1 && ox0 + TW + 1 <= P.w;      // the whole ring lies inside the image (wave-uniform)
        // ---- conv1 + PReLU on the ring -> mid (bf16, [ring pixel][CS]) ----
#pragma unroll
        for (int ps = 0; ps < PER1 / NH1; ++ps) {
            f32x4_t acc[MT][NH1];
            bf16x8_t bq[2][NH1], aq[2][W1L ? MT : 1];
#pragma unroll
            for (int nh = 0; nh < NH1; ++nh) bq[0][nh] = as_frag(*(const uint4*)(xs + pix1[ps * NH1 + nh] + toff1[0]));
            if constexpr (W1L) {
#pragma unroll
                for (int m = 0; m < MT; ++m) aq[0][m] = as_frag(*(const uint4*)(wl + (m * KS) * 1024));
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
#pragma unroll
                    for (int nh = 0; nh < NH1; ++nh) bq[(s + 1) & 1][nh] = as_frag(*(const uint4*)(xs + pix1[ps * NH1 + nh] + toff1[s + 1]));
                    if constexpr (W1L) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) aq[(s + 1) & 1][m] = as_frag(*(const uint4*)(wl + (m * KS + s + 1) * 1024));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nh = 0; nh < NH1; ++nh)
                        acc[m][nh] = mfma16(W1L ? aq[s & 1][m] : A1[W1L ? 0 : m][W1L ? 0 : s], bq[s & 1][nh], s == 0 ? bias1[m] : acc[m][nh]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nh = 0; nh < NH1; ++nh) {
                const int i1 = ps * NH1 + nh;
                bool in = true;
                if (!inner) {                                        // (border tiles only: the ring coordinates back from the region offset)
                    const int pq = pix1[i1] / PSB, my = pq / RW, mx = pq - my * RW;
                    const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
                    in = gy >= 0 && gy < P.h && gx >= 0 && gx < P.w;
                }
                unsigned wd[MT][2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float v[4] = {acc[m][nh][0], acc[m][nh][1], acc[m][nh][2], acc[m][nh][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = max_bare(v[r], slope * v[r]);
                    wd[m][0] = in ? pack_bf2(v[0], v[1]) : 0u;
                    wd[m][1] = in ? pack_bf2(v[2], v[3]) : 0u;
                }
                if (mwr[i1] >= 0) {
                    char* dst = mids + mwr[i1];
                    if constexpr (MT == 2 || MT == 4) {
#pragma unroll
                        for (int m = 0; m < MT; m += 2)
                            if (c0 + m * 4 < CS) *(uint4*)(dst + m * 8) = make_uint4(wd[m][0], wd[m][1], wd[m + 1][0], wd[m + 1][1]);
                    } else {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            if (c0 + m * 4 < CS) *(uint2*)(dst + m * 8) = make_uint2(wd[m][0], wd[m][1]);
                    }
                }
            }
        }
        __builtin_amdgcn_s_barrier();                                // mid is complete
        // ---- conv2 from mid; epilogue: * ca, + x, store (MODE 2 of conv3p_kernel) ----
        const __amdgpu_buffer_rsrc_t ro = frame_rsrc(P.out + (size_t)t * frame_elems, frame_bytes);
        const int o2 = (oy0 * P.w + ox0) * PSB;
        const bool full = (oy0 + TH <= P.h) && (ox0 + TW <= P.w);
        const int ylim = P.h - oy0 - wv * RPW, xlim = P.w - ox0 - p;
#pragma unroll
        for (int ps = 0; ps < NTW / NH; ++ps) {
            const int N0 = ps * NH;
            f32x4_t acc[MT][NH];
            bf16x8_t bq[2][NH], aq[2][W2L ? MT : 1];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh)
                bq[0][nh] = as_frag(*(const uint4*)(mids + addr2[0] + (((N0 + nh) >> 1) * MW + ((N0 + nh) & 1) * 16) * PSB));
            if constexpr (W2L) {
#pragma unroll
                for (int m = 0; m < MT; ++m) aq[0][m] = as_frag(*(const uint4*)(wl + W2OFF + (m * KS) * 1024));
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh)
                        bq[(s + 1) & 1][nh] = as_frag(*(const uint4*)(mids + addr2[s + 1] + (((N0 + nh) >> 1) * MW + ((N0 + nh) & 1) * 16) * PSB));
                    if constexpr (W2L) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) aq[(s + 1) & 1][m] = as_frag(*(const uint4*)(wl + W2OFF + (m * KS + s + 1) * 1024));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh)
                        acc[m][nh] = mfma16(W2L ? aq[s & 1][m] : A2[W2L ? 0 : m][W2L ? 0 : s], bq[s & 1][nh], s == 0 ? bias2[m] : acc[m][nh]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                const int nn = N0 + nh, rr = nn >> 1, xb = nn & 1;
                float v[MT][4];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const rword_t qv = *(const rword_t*)(xs + raddr + (rr * RW + xb * 16) * PSB + ((m * 8) / PIECE) * PIECE);
                    const unsigned lo = qv[((m * 8) % PIECE) / 4], hi = qv[((m * 8) % PIECE) / 4 + 1];
                    v[m][0] = mul_then_add(acc[m][nh][0], osc[m].x, bf_lo(lo)); v[m][1] = mul_then_add(acc[m][nh][1], osc[m].y, bf_hi(lo));
                    v[m][2] = mul_then_add(acc[m][nh][2], osc[m].z, bf_lo(hi)); v[m][3] = mul_then_add(acc[m][nh][3], osc[m].w, bf_hi(hi));
                }
                const bool ok = full || ((rr < ylim) && (xb * 16 < xlim));
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    int vo = o2 + ooff[rr][q] + xb * 16 * PSB;
                    vo = ok ? vo : OOR;
                    if constexpr (PIECE == 16) {
                        const int m = q * 2;
                        const rword_t d = {pack_bf2(v[m][0], v[m][1]), pack_bf2(v[m][2], v[m][3]), pack_bf2(v[m + 1][0], v[m + 1][1]), pack_bf2(v[m + 1][2], v[m + 1][3])};
                        __builtin_amdgcn_raw_buffer_store_b128(d, ro, vo, 0, 0);
                    } else {
                        const rword_t d = {pack_bf2(v[q][0], v[q][1]), pack_bf2(v[q][2], v[q][3])};
                        __builtin_amdgcn_raw_buffer_store_b64(d, ro, vo, 0, 0);
                    }
                }
            }
        }
        if (++ty == P.nty) {
            ty = 0;
            if (++tx == P.ntx) { tx = 0; ++t; if (i + 1 < n) { load_osc(t); pin_osc(); } }
        }
    }
}

// Segment length S of a launch's tile columns -- the unit the channel sums of a CAB's first conv are flushed in, i.e. what a pool row covers.
// It is a function of the IMAGE (h, w) and the kernel instance only, never of the frame count, the device or a measurement override: the
// sums of equal frames are then bit-equal whatever window they sit in (a temporally split window must equal the long one bit for bit,
// tests/test_temporal_split.py), on whatever device.  Chosen on a MODEL machine (256 CUs x the instance's default workgroups per CU) over the
// window lengths of the BASELINE configs: of 8, 6, 4, 3, 2, 1 tiles the one with the smallest modelled time, (1 + F / S) / balance averaged
// over those lengths -- balance = mean tiles per workgroup / the busiest one's (chunks are whole segments), F = 0.25 = a flush of the channel
// sums (~64 VALU instructions per wave) relative to a tile (measured: the 24-channel conv at 20 x 360 x 640 runs 92 us with S = 3, 106 with S = 1).
int c3p_segment(int h, int w, int th, int wgs_model) {
    const int ntx = (w + 31) / 32, nty = (h + th - 1) / th;
    const int cand[6] = {8, 6, 4, 3, 2, 1}, Ts[5] = {12, 16, 20, 36, 52};
    double best = 1e30;
    int bestS = 1;
    for (int k = 0; k < 6; ++k) {
        const int S = cand[k] < nty ? cand[k] : nty;
        const int nseg = (nty + S - 1) / S;
        double cost = 0.0;
        for (int j = 0; j < 5; ++j) {
            const long ntiles = (long)ntx * nty * Ts[j];
            long g = 256L * wgs_model;
            if (g > (ntiles + 7) / 8) g = (ntiles + 7) / 8;
            if (g < 1) g = 1;
            const long nsg = (long)nseg * ntx * Ts[j], qs = (nsg + g - 1) / g;
            cost += (1.0 + 0.25 / S) * (double)(qs * S) * (double)g / (double)ntiles;
        }
        if (cost < best - 1e-9) { best = cost; bestS = S; }
    }
    return bestS;
}

struct C3PPlan { int ntx, nty, S, nseg, nsg, qs, grid, pool_rows; };

// `wgs` persistent workgroups per CU on `ncu` CUs; S from c3p_segment.  A chunk (workgroup) is qs consecutive segments of the
// (frame > tile column > segment) list.
C3PPlan c3p_plan(int T, int h, int w, int th, int ncu, int wgs, int S) {
    C3PPlan p;
    p.ntx = (w + 31) / 32; p.nty = (h + th - 1) / th;
    const long ntiles = (long)p.ntx * p.nty * T;
    long g = (long)ncu * wgs;
    if (g > (ntiles + 7) / 8) g = (ntiles + 7) / 8;                  // every workgroup at least 8 tiles to amortise its prologue
    if (g < 1) g = 1;
    p.S = S < p.nty ? S : p.nty;
    p.nseg = (p.nty + p.S - 1) / p.S;
    const long nsg = (long)p.nseg * p.ntx * T;
    p.nsg = (int)nsg;
    p.qs = (int)((nsg + g - 1) / g);
    const int chunks = (p.nsg + p.qs - 1) / p.qs;
    p.grid = (chunks + 7) / 8 * 8;
    p.pool_rows = 4 * p.ntx * p.nseg;
    return p;
}

template <int MT, int CS, int TH, int D, int MODE, bool RL>
int launch_conv3p_mode(const C3P& K, const C3PPlan& pl, hipStream_t st) {
    constexpr int NPB = CS / 8, NDMA = ((TH + 2) * 34 * NPB + 63) / 64, KS = (9 * CS + 31) / 32, RDMA = RL ? (TH * 32 * NPB) / 64 : 0;
    C3P P = K;
    P.ntx = pl.ntx; P.nty = pl.nty; P.S = pl.S; P.nseg = pl.nseg; P.nsg = pl.nsg; P.qs = pl.qs; P.pool_rows = pl.pool_rows;
    const size_t lds = (size_t)(D + 1) * (NDMA + RDMA) * 1024 + (MT * KS > 16 ? MT * KS * 1024 : 0);
    if (lds > 160 * 1024) return SN_EINVAL;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv3p_kernel<MT, CS, TH, D, MODE, RL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SN_ELAUNCH;
    hipLaunchKernelGGL((conv3p_kernel<MT, CS, TH, D, MODE, RL>), dim3(pl.grid), dim3(320), lds, st, P);
    return sn_check_launch();
}

// D2 / RL2: prefetch depth and residual staging of the MODE 2 instance (the residual tile costs LDS: fewer buffers or fewer workgroups per CU)
template <int MT, int CS, int TH, int D, int D2, bool RL2>
int launch_conv3p(const C3P& K, const C3PPlan& pl, int mode, hipStream_t st) {
    switch (mode) {
        case 0: return launch_conv3p_mode<MT, CS, TH, D, 0, false>(K, pl, st);
        case 1: return launch_conv3p_mode<MT, CS, TH, D, 1, false>(K, pl, st);
        case 2: return launch_conv3p_mode<MT, CS, TH, D2, 2, RL2>(K, pl, st);
        case 3: return launch_conv3p_mode<MT, CS, TH, D, 3, false>(K, pl, st);
        default: return SN_EINVAL;
    }
}

template <int MT, int CS, int TH, int D, int WR = 0>
int launch_cabp(const C3P& K, const C3PPlan& pl, hipStream_t st) {
    constexpr int NPB = CS / 8, NDMA = ((TH + 4) * 36 * NPB + 63) / 64, KS = (9 * CS + 31) / 32;
    C3P P = K;
    P.ntx = pl.ntx; P.nty = pl.nty; P.S = pl.S; P.nseg = pl.nseg; P.nsg = pl.nsg; P.qs = pl.qs; P.pool_rows = 0;
    const size_t lds = (size_t)(D + 1) * NDMA * 1024 + ((TH + 2) * 34 * CS * 2 + 15) / 16 * 16 + ((WR & 1) ? 0 : MT * KS * 1024) + ((WR & 2) ? 0 : MT * KS * 1024);
    if (lds > 160 * 1024) return SN_EINVAL;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)cabp_kernel<MT, CS, TH, D, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SN_ELAUNCH;
    hipLaunchKernelGGL((cabp_kernel<MT, CS, TH, D, WR>), dim3(pl.grid), dim3(320), lds, st, P);
    return sn_check_launch();
}

// epilogue mode of a descriptor (conv3p_kernel), -1: a combination the streaming kernel does not implement
// (want_pool: sn_conv_pool_blocks is asked BEFORE the caller has a pool buffer to put into the descriptor)
int c3p_mode(const sn_conv_desc* d, bool want_pool = false) {
    if (d->res2) return -1;
    const bool pool = d->pool != nullptr || want_pool;
    if (d->act == 0 && !pool && !d->oscale && !d->res) return 0;
    if (d->act == 1 && d->prelu >= 0.f && d->prelu <= 1.f && pool && !d->oscale && !d->res) return 1;
    if (d->act == 0 && !pool && d->oscale && d->res) return 2;
    return -1;
}

}  // namespace

// ---- entry points used by sn_conv.hip (sn_conv2d / sn_conv_pool_blocks) -------------------------------------------------------------
// key = M-tiles * 1000 + storage channels of a conv the streaming kernel has an instance for, else 0
int sn_conv3p_key(const sn_conv_desc* d, bool want_pool) {
    if (!(d->k == 3 && d->stride == 1 && d->pad == 1 && d->in_mode == 0 && d->out_mode == 0 && d->n_in == 1 && d->cs_in == d->cs_out &&
          d->ks == (9 * d->cs_in + 31) / 32 && d->h_in == d->h_out && d->w_in == d->w_out) || c3p_mode(d, want_pool) < 0) return 0;
    if ((size_t)d->h_out * d->w_out * d->cs_out * 2 >= 0x7fffffffull) return 0;       // a frame is addressed with 32-bit byte offsets
    const int key = d->mt * 1000 + d->cs_in;
    // (the 16-channel conv with scale + residual is the one case the tile kernel still wins: 348 vs 395 us at 20 x 720 x 1280; bit 8 of flags forces it here)
    if (key == 1016 && c3p_mode(d, want_pool) == 2 && !(d->flags & 256)) return 0;
    return (key == 1016 || key == 2024 || key == 3040 || key == 3048 || key == 4064) ? key : 0;
}

static int c3p_ncu() {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) {
        (void)hipGetLastError();
        return 0;
    }
    return ncu;
}
// persistent workgroups per CU (LDS: 3 buffers of 11 / 16 KB; registers: 5-wave workgroups); bits 4..7 of sn_conv_desc.flags override it (measurements)
// MODE 2 may stage the residual tile in LDS (RL); bit 9 of flags asks for the register form everywhere (measurements)
// (measured, conv2 of a CAB, LDS vs registers: 24 channels 141 vs 153 us at 20 x 360 x 640 and 1327 vs 1347 at 52 x 720 x 1280; 40 channels 591 vs 576;
// 48 channels 184 vs 182: staged for <= 24 channels, registers above, where the tile would also cost a prefetch buffer)
static bool c3p_rl(const sn_conv_desc* d, int key, int mode) { return mode == 2 && (key == 1016 || key == 2024) && !(d->flags & 512); }
// workgroups per CU of an instance's first-conv / statistics modes: the MODEL occupancy c3p_segment balances for (no override, no device query)
static int c3p_wgs_default(int key) { return key == 1016 ? 3 : key == 2024 ? 2 : 1; }
static int c3p_wgs(const sn_conv_desc* d, int key, int mode) {
    const int o = (d->flags >> 4) & 15;
    if (o) return o;
    // 16 channels: 3 (measured best of 1-4), 2 with the residual tile in LDS (3 x 19 KB per workgroup); 24 channels: 2 (one: +6 % time);
    // >= 40 channels: weights (36-72 KB) + buffers fill the LDS of a CU
    if (key == 1016) return c3p_rl(d, key, mode) ? 2 : 3;
    return key == 2024 ? 2 : 1;
}

// rows of `pool` per frame when sn_conv2d runs this descriptor on the streaming kernel; 0: it will not (no instance, or no device to plan for)
int sn_conv3p_pool_rows(const sn_conv_desc* d) {
    const int key = sn_conv3p_key(d, true), ncu = key ? c3p_ncu() : 0;
    if (!key || !ncu) return 0;
    return c3p_plan(d->T, d->h_out, d->w_out, 8, ncu, c3p_wgs(d, key, 1), c3p_segment(d->h_out, d->w_out, 8, c3p_wgs_default(key))).pool_rows;
}

// lines_len > 0: the statistics pass of the fused CAB (sn_cab_stats): d is a MODE-1 descriptor whose `out` is the border-line buffer
int sn_conv3p_launch(const sn_conv_desc* d, int lines_len, void* stream) {
    const int key = sn_conv3p_key(d, false), ncu = key ? c3p_ncu() : 0;
    if (!key || !ncu) return SN_EINVAL;
    C3P K;
    K.in = (const bf16_t*)d->in[0]; K.res = (const bf16_t*)d->res; K.out = (bf16_t*)d->out;
    K.wfrag = (const uint4*)d->wfrag; K.bias = d->bias; K.oscale = d->oscale; K.oscale_stride = d->oscale_stride;
    K.pool = d->pool; K.pool_rows = 0; K.act = d->act; K.prelu = d->prelu;
    K.T = d->T; K.h = d->h_out; K.w = d->w_out; K.lines_len = lines_len; K.dbg = (d->flags >> 12) & 7;
    const int mode = lines_len > 0 ? 3 : c3p_mode(d);
    if (lines_len > 0 && c3p_mode(d) != 1) return SN_EINVAL;
    const C3PPlan pl = c3p_plan(d->T, d->h_out, d->w_out, 8, ncu, c3p_wgs(d, key, mode), c3p_segment(d->h_out, d->w_out, 8, c3p_wgs_default(key)));
    hipStream_t st = (hipStream_t)stream;
    const bool rl = c3p_rl(d, key, mode);
    // <M-tiles, channels, tile rows, prefetch depth, prefetch depth of MODE 2, MODE 2 with the residual in LDS>
    switch (key) {
        case 1016:
            if (((d->flags >> 10) & 3) == 1) return launch_conv3p<1, 16, 8, 3, 3, false>(K, pl, mode, st);      // bits 10..11: deeper prefetch (measurements)
            if (((d->flags >> 10) & 3) == 2) return launch_conv3p<1, 16, 8, 4, 4, false>(K, pl, mode, st);
            return rl ? launch_conv3p<1, 16, 8, 2, 2, true>(K, pl, mode, st) : launch_conv3p<1, 16, 8, 2, 2, false>(K, pl, mode, st);
        case 2024: return rl ? launch_conv3p<2, 24, 8, 2, 1, true>(K, pl, mode, st) : launch_conv3p<2, 24, 8, 2, 2, false>(K, pl, mode, st);
        case 3040: return launch_conv3p<3, 40, 8, 2, 2, false>(K, pl, mode, st);
        case 3048: return launch_conv3p<3, 48, 8, 2, 2, false>(K, pl, mode, st);
        case 4064: return launch_conv3p<4, 64, 8, 1, 1, false>(K, pl, mode, st);      // 72 KB of weights: two buffers of 43 KB
        default: return SN_EINVAL;
    }
}

// pass 2 of the fused CAB on the streaming structure (sn_cab_fused routes here unless the tile form is asked for); SN_EINVAL: no instance / no device
int sn_cabp_launch(const sn_conv_desc* a, const sn_conv_desc* b, void* stream) {
    const int key = a->mt * 1000 + a->cs_in, ncu = c3p_ncu();
    if (!ncu || !(key == 1016 || key == 2024) || a->act != 1 || !(a->prelu >= 0.f && a->prelu <= 1.f) || !b->oscale || b->res2) return SN_EINVAL;
    if ((size_t)a->h_out * a->w_out * a->cs_out * 2 >= 0x7fffffffull) return SN_EINVAL;
    C3P K;
    K.in = (const bf16_t*)a->in[0]; K.res = nullptr; K.out = (bf16_t*)b->out;
    K.wfrag = (const uint4*)a->wfrag; K.bias = a->bias; K.w2 = (const uint4*)b->wfrag; K.b2 = b->bias;
    K.oscale = b->oscale; K.oscale_stride = b->oscale_stride; K.pool = nullptr; K.pool_rows = 0; K.act = 1; K.prelu = a->prelu;
    K.T = a->T; K.h = a->h_out; K.w = a->w_out; K.lines_len = 0; K.dbg = 0;
    // LDS per workgroup (region buffers + mid + both weight sets): 16 channels 63 KB with three region buffers (two workgroups per CU), 49 KB with
    // two (three per CU); 24 channels 107 / 86 KB (one per CU).  Bits 4..7 / 10..11 of conv1's flags override the count / ask for two buffers (measurements).
    const int o = (a->flags >> 4) & 15;
    const bool shallow = ((a->flags >> 10) & 3) == 3;      // (codes 1 and 2 mean deeper prefetch to the statistics pass that shares these flags)
    const C3PPlan pl = c3p_plan(a->T, a->h_out, a->w_out, 8, ncu, o ? o : (key == 1016 ? (shallow ? 3 : 2) : 1), c3p_segment(a->h_out, a->w_out, 8, c3p_wgs_default(key)));
    hipStream_t st = (hipStream_t)stream;
    if (shallow) return key == 1016 ? launch_cabp<1, 16, 8, 1>(K, pl, st) : launch_cabp<2, 24, 8, 1>(K, pl, st);
    // 16 channels: conv2's weight set in registers (126 of them), 327 against 341 us at 20 x 720 x 1280; bit 12 of CONV2's flags asks for both sets in LDS (measurements)
    if (key == 1016 && !((b->flags >> 12) & 1)) return launch_cabp<1, 16, 8, 2, 2>(K, pl, st);
    return key == 1016 ? launch_cabp<1, 16, 8, 2>(K, pl, st) : launch_cabp<2, 24, 8, 2>(K, pl, st);
}
