// Shared device helpers for the Shift-Net gfx950 kernels (CDNA4, wave64).
//
// Conventions used by every kernel in this directory
//   * activations live in HBM as NHWC bf16: [T][H][W][Cs], Cs a multiple of 8 (16 B), pad channels are zero;
//   * every 1x1 / dense conv is an MFMA GEMM with the WEIGHTS as the A operand (M = out channels) and the
//     ACTIVATIONS as the B operand (N = 16 pixels): v_mfma_f32_16x16x32_bf16.  For that instruction
//       A: lane l holds A[m = l&15][k-slot (l>>4, j)], j = 0..7      (8 bf16 = 16 B)
//       B: lane l holds B[k-slot (l>>4, j)][n = l&15]
//       D: lane l, reg r holds D[m = (l>>4)*4 + r][n = l&15]
//     The hardware pairs A's slot (g, j) with B's slot (g, j); which logical k a slot means is OUR choice, so
//     a lane's B operand is simply 8 consecutive channels (16 B) of "its" pixel, straight from NHWC memory.
//   * the host prepacks weights into A-fragment order [mt][ks][lane][8] (shiftnet_amd/prep.py), including any
//     row permutation that makes a lane's D registers contiguous channels, so kernels never shuffle channels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint16_t bf16_t;  // storage type

#define SN_OK 0
#define SN_EINVAL (-22)
#define SN_ELAUNCH (-5)

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_to_f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even fp32 -> bf16 pair (v_cvt_pk_bf16_f32 on gfx950); lo goes to bits 0..15
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, h);
}
// fp32 pair -> packed fp16, round to nearest even (two v_cvt_f16_f32 + v_pack_b32_f16); lo goes to bits 0..15
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const h2_t h = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ bf16_t f_to_bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ void unpack8(const uint4 q, float* v) {
    v[0] = bf_lo(q.x); v[1] = bf_hi(q.x); v[2] = bf_lo(q.y); v[3] = bf_hi(q.y);
    v[4] = bf_lo(q.z); v[5] = bf_hi(q.z); v[6] = bf_lo(q.w); v[7] = bf_hi(q.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    uint4 q;
    q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]); q.z = pack_bf2(v[4], v[5]); q.w = pack_bf2(v[6], v[7]);
    return q;
}
__device__ __forceinline__ bf16x8_t as_frag(const uint4 q) { return __builtin_bit_cast(bf16x8_t, q); }

__device__ __forceinline__ f32x4_t mfma16(const bf16x8_t a, const bf16x8_t b, const f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): the IEEE division sequence costs ~10 VALU instructions per element
// acc + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16).  With a weight word that has ONE non-zero half
// this is "fma on one bf16 lane of a packed pair" without unpacking the data: the stencil kernels use it that way.
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: four full-rate v_add_f32_dpp.
// __shfl_xor lowers to ds_bpermute_b32 + s_waitcnt per step, ~100 cycles each and serialised by the compiler.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// integer DPP move, zero where the source lane is outside the row; 0x100 + n = row_shl:n (lane i reads lane i + n)
template <int CTRL> __device__ __forceinline__ int dpp_movi(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_mov<0xB1>(x);     // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);     // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);    // row_half_mirror
    x += dpp_mov<0x140>(x);    // row_mirror
    return x;
}

// Sum over the four DPP rows (lanes l, l^16, l^32, l^48), result in every lane: v_permlane16_swap / v_permlane32_swap
// (gfx950) exchange odd/even rows and the wave halves inside the VALU, no LDS round trip (ds_bpermute) per step.
__device__ __forceinline__ float sum_rows4(float x) {
    // Inline asm: the instruction swaps IN PLACE between two distinct registers; through the builtin hipcc (ROCm 7.2) adds
    // result 0 to itself when both inputs carry the same value.  s_nop 1: wait states between a VALU write and the swap.
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    const float s = a + b;
    float c = s, d = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
}

// Where the half-channel slabs of the virtual 1.5 C-channel input u of a GSTS unit come from (gshift_deblur1.py:504-528
// keep, gshift_deblur2.py:499-519 circular).  Every slab is C/2 consecutive channels of one frame: element (pixel i, channel c) of
//   u[:, :C/2]  = p0[i * s0 + c],   u[:, C/2:C] = p1[i * s1 + c],   borrowed half (input of the spatial shift) = pb[i * sb + c].
// mode 0: CAB1, u = x[t].  mode 1 / 2: forward / reverse unit.  wrap 0: the window's boundary frame is kept un-rolled; 1: circular roll
// inside the tensor; 2: the boundary frame's neighbour belongs to the adjacent rank of a temporally split window
// (shiftnet_amd/temporal_split.py) and only its borrowed half exists here, as the contiguous [h][w][C/2] buffer `halo` (pixel stride C/2).
// frame sub-range of an sn_unit_src: (first frame, count); false when it does not lie inside [0, T)
#define SN_FRAME_RANGE(s, T0, NT) const int T0 = (s)->nt > 0 ? (s)->t0 : 0, NT = (s)->nt > 0 ? (s)->nt : (s)->T; \
    if (T0 < 0 || NT < 1 || T0 + NT > (s)->T) return SN_EINVAL
template <typename E> struct SnSlabs { const E* p0; const E* p1; const E* pb; int s0, s1, sb; };
template <typename E>
__device__ __forceinline__ SnSlabs<E> sn_unit_slabs(const E* x, const E* halo, int T, int hw, int C, int mode, int wrap, int t) {
    const int Ch = C >> 1;
    const E* xt = x + (ptrdiff_t)t * hw * C;
    SnSlabs<E> s;
    s.p0 = xt; s.p1 = xt + Ch; s.pb = xt; s.s0 = s.s1 = s.sb = C;
    if (mode == 1) {                                   // forward: u[:, :Ch] = x[t-1][Ch:], u[:, Ch:] = x[t][:Ch], borrowed = x[t-1][Ch:]
        if (t > 0 || wrap == 1) { s.p0 = x + (ptrdiff_t)(t > 0 ? t - 1 : T - 1) * hw * C + Ch; s.p1 = xt; s.pb = s.p0; }
        else if (wrap == 2) { s.p0 = halo; s.s0 = Ch; s.p1 = xt; s.pb = halo; s.sb = Ch; }
        else s.pb = xt;                                // kept boundary frame: borrowed = its own lower half
    } else if (mode == 2) {                            // reverse: u[:, :Ch] = x[t][Ch:], u[:, Ch:] = x[t+1][:Ch], borrowed = x[t+1][:Ch]
        if (t < T - 1 || wrap == 1) { s.p0 = xt + Ch; s.p1 = x + (ptrdiff_t)(t < T - 1 ? t + 1 : 0) * hw * C; s.pb = s.p1; }
        else if (wrap == 2) { s.p0 = xt + Ch; s.p1 = halo; s.s1 = Ch; s.pb = halo; s.sb = Ch; }
        else s.pb = xt + Ch;                           // kept boundary frame: borrowed = its own upper half
    }
    return s;
}

// ---- squeeze-excite folded into its producer (shiftnet_hip.h: sn_se_fold) -------------------------------------------------------
// CALayer2 needs the global average pool of g2: every workgroup of a frame stores its partial channel sums, and the LAST workgroup of
// the frame to finish (a ticket counter per frame) reduces them in a FIXED order -- so the result does not depend on which workgroup
// was last: bit-reproducible -- and runs the C -> C/r -> C MLP, instead of a separate sn_ca_mlp launch between the two phases.
// Inter-workgroup visibility (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by other CUs' stores;
// cdna_hip_programming.md Guideline 16, counter form): partial sums leave as sc1 (write-through) stores -> every wave waits for
// its stores (vmcnt(0)) -> workgroup barrier -> lane 0 takes the ticket with a relaxed agent-scope fetch_add; the last arriver issues an
// agent-scope ACQUIRE fence, re-arms the counter and reads the partial sums with sc1 loads.
struct SeFold { const float* wa; const float* wb; float* ca; unsigned* ticket; unsigned* bad; float inv_hw; int c, cr; };
// the range guard of the half-precision intermediates: a channel sum that is not finite (an fp16 `a`, g1 or r overflowed upstream) raises *bad
__device__ __forceinline__ void sn_flag_nonfinite(unsigned* bad, float v) {
    if (bad && !(fabsf(v) <= 3.0e38f)) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sn_pool_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// pool_t: [nblk][cpad] partial sums of frame t (every row written by exactly one workgroup of the launch, with sn_pool_store);
// narrive: how many calls are made for frame t by the whole launch (each after storing its rows); lds: >= 16 + (nthreads / (cpad / 4)) * cpad
// + 2 * 128 floats, no longer read by anybody in the workgroup once its first barrier is passed; cpad % 4 == 0, cpad <= 128, cr <= 128.
// Called by ALL threads.  The reduction order depends on (nblk, nthreads, cpad) only.
__device__ __forceinline__ void sn_se_tail(const SeFold& S, const float* pool_t, int nblk, int narrive, int cpad, int t, float* lds, int tid, int nthreads) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's partial-sum stores are acknowledged by memory
    __syncthreads();
    unsigned* flag = (unsigned*)lds;
    if (tid == 0) *flag = __hip_atomic_fetch_add(S.ticket + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(narrive - 1) ? 1u : 0u;
    __syncthreads();
    if (!*flag) return;                                              // workgroup-uniform
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(S.ticket + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch on this stream
    }
    __syncthreads();
    // thread = (row group `part`, channel quad): the nblk rows are split over nthreads / (cpad / 4) groups, each walks its rows in order with
    // four channels per thread; the groups are then added in order
    const int nq = cpad >> 2, parts = nthreads / nq, cq = tid % nq, part = tid / nq;
    float* acc = lds + 16; float* mean = acc + parts * cpad; float* hid = mean + 128;
    if (part < parts) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
        for (int b = part; b < nblk; b += parts) {
            const float* r = pool_t + (size_t)b * cpad + 4 * cq;
            s0 += __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s1 += __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s2 += __hip_atomic_load(r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s3 += __hip_atomic_load(r + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float* a = acc + part * cpad + 4 * cq;
        a[0] = s0; a[1] = s1; a[2] = s2; a[3] = s3;
    }
    __syncthreads();
    if (tid < cpad) {
        float m = 0.f;
        for (int q = 0; q < parts; ++q) m += acc[q * cpad + tid];
        sn_flag_nonfinite(S.bad, m);
        mean[tid] = m * S.inv_hw;
    }
    __syncthreads();
    if (tid < S.cr) {
        float hsum = 0.f;
        for (int j = 0; j < S.c; ++j) hsum += S.wa[tid * S.c + j] * mean[j];
        hid[tid] = hsum > 0.f ? hsum : 0.f;
    }
    __syncthreads();
    if (tid < cpad) {
        float o = 0.f;
        if (tid < S.c) {
            for (int j = 0; j < S.cr; ++j) o += S.wb[tid * S.cr + j] * hid[j];
            o = sigmoidf_(o);
        }
        S.ca[(size_t)t * cpad + tid] = o;                            // read by the NEXT kernel on the stream: the kernel boundary publishes it
    }
}

// wave-uniform wave index (threadIdx-derived values are "divergent" to the compiler; make it provably uniform)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- XCD-aware tile walk for the halo-reading tile kernels (K0, K12, dense convs) ------------------------------------------------
// The dispatcher hands workgroups to the 8 XCDs round-robin in linear order (x fastest), and every XCD has its own 4 MiB L2.  With
// the natural grid (tiles_x, tiles_y, T) horizontally adjacent tiles -- which share their halo columns: 18 of K0's 34 -- never meet
// in one L2, so the halo is fetched from the fabric once per tile (PMC round 2: K0 2.4x, K12 1.37x its algorithmic bytes).  Here
// the grid is (8 * tiles_x, chunk): blockIdx.x & 7 IS the XCD, blockIdx.x >> 3 the tile column, and XCD k owns the contiguous run
// [k * chunk, (k + 1) * chunk) of "row-frames" (frame-major tile rows), walked in order: neighbours in x run concurrently on the
// same XCD, neighbours in y one tile row (tiles_x workgroups) apart, both inside one L2.  A device with another XCD count
// still gets a bijection, just not the locality.  SN_XCD_TILES=0 restores the natural order (A/B measurements).
#ifndef SN_XCD_TILES
#define SN_XCD_TILES 1
#endif
struct XcdTiles { int ntx, nty, nrf, chunk; uint32_t inv_nty; };
static inline XcdTiles sn_xcd_tiles(int ntx, int nty, int T) {
    XcdTiles g; g.ntx = ntx; g.nty = nty; g.nrf = nty * T;
    g.chunk = SN_XCD_TILES ? (g.nrf + 7) / 8 : g.nrf;
    g.inv_nty = nty > 1 ? (uint32_t)((0x100000000ull + (uint32_t)nty - 1) / (uint32_t)nty) : 0u;      // exact quotients while nrf * nty < 2^32
    return g;
}
static inline dim3 sn_xcd_grid(const XcdTiles& g) { return SN_XCD_TILES ? dim3(8u * g.ntx, g.chunk, 1) : dim3(g.ntx, g.nrf, 1); }
// (bx, by) = block index -> (t, ty, tx); false: this workgroup is padding of the last XCD's chunk.  Host-callable so that
// tests/test_host_logic.py can check, without a GPU, that every tile of every grid shape is produced exactly once.
__host__ __device__ inline bool sn_xcd_decode(const XcdTiles& g, uint32_t bx, uint32_t by, int& t, int& ty, int& tx) {
#if SN_XCD_TILES
    tx = (int)(bx >> 3);
    const uint32_t rf = (bx & 7) * (uint32_t)g.chunk + by;
#else
    tx = (int)bx;
    const uint32_t rf = by;
#endif
    if (rf >= (uint32_t)g.nrf) return false;
    t = g.inv_nty ? (int)(((uint64_t)rf * g.inv_nty) >> 32) : (int)rf;       // = __umulhi(rf, inv_nty): s_mul_hi_u32 on the device
    ty = (int)rf - t * g.nty;
    return true;
}
// returns before any barrier when false
__device__ __forceinline__ bool sn_xcd_tile(const XcdTiles& g, int& t, int& ty, int& tx) {
    return sn_xcd_decode(g, blockIdx.x, blockIdx.y, t, ty, tx);
}

// hipGetLastError() is per-thread sticky state that other libraries in the process (PyTorch's allocator polling
// events, ...) may leave set; every entry point therefore clears it BEFORE launching and reads it right after.
static inline void sn_clear_error() { (void)hipGetLastError(); }
static inline int sn_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SN_OK : SN_ELAUNCH;
}
