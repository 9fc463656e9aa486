// Fused dense CAB (gfx950) for the narrow CABs of the encoder-decoder (c <= 24: every CAB of Shift-Net-s' TFR_UNets, the full-resolution ones of "+"):
//     out = x + ca * conv2(PReLU(conv1(x)))  [+ extra]                                   (CAB, gshift_deblur1.py:141-156)
// The two-kernel path (csrc/sn_conv.hip) is five tensor passes per CAB -- conv1 reads x and writes mid, conv2 reads mid and x and writes out -- for two
// algorithmic ones, and in the "+" model the dense convs move more bytes than the GSTS path (988 of 1767 GB per config-3 window).  Here `mid` lives in
// LDS: a workgroup walks a 62-column strip top to bottom, one image row per step:
//   waves 0-3 ("conv1"): wave n = N-tile n of the 64 mid columns x0-1 .. x0+62: 3x3 conv on the x ring (MFMA, weights resident in registers) -> PReLU
//               -> zero outside the image -> mid ring (bf16, as the two-kernel path stores it).  They also prefetch the input rows (two steps ahead).
//   waves 4-7 ("conv2"): wave n = N-tile n of the 64 output columns x0 .. x0+63 (62 owned): 3x3 conv on the mid ring -> x CALayer scale -> + x (+ extra)
//               -> NHWC stores.
// The CALayer scale must be known before conv2; its pooled input is linear in `mid` (sn_cab_ca), so a STATISTICS pass comes first: the same kernel with
// the conv1 role only, reducing the sums of mid (total, first / last row, first / last column, corners) -- sn_cabf_ca turns them into the scale.
// Per CAB: read x (statistics), read x, write out: three passes; conv1 is computed twice (the MFMA pipe is idle enough: these convs are HBM-bound).
// Operands: the same fragments as sn_conv2d (prep.pack_conv: k index = tap * CS + channel, rows in 'conv' order: lane (g, p) holds channels
// [4 MT g, 4 MT (g + 1)) of pixel p), so results follow the two-kernel path up to summation order.
#include "sn_common.h"
#include "../../include/shiftnet_hip.h"
#include <type_traits>

namespace {

constexpr __host__ __device__ int cabf_lds_slots(int npb) {      // = sn_lds_slots of csrc/sn_conv.hip: pixel stride in 16-byte slots
    const int k = npb <= 2 ? 2 : 4 * ((npb - 2 + 3) / 4) + 2;
    return k <= npb + 1 ? k : ((npb & 1) ? npb : npb + 1);
}

struct CabfK {
    const bf16_t* x; const bf16_t* res2; bf16_t* out;
    const uint4* wfrag1; const uint4* wfrag2; const float* bias1; const float* bias2;
    float prelu; const float* ca; float* part;
    int T, h, w, nsx, nsy, seg, vw;
};

#ifndef CABF_WPS
#define CABF_WPS 4
#endif
constexpr int CABF_OWN = 62, CABF_XW = 66, CABF_MW = 66;      // own columns per strip; x ring columns (x0-2 .. x0+63); mid ring columns (64 + 2 pad)

template <int MT, int CS, int MODE>      // MODE 0: statistics pass (conv1 role only), 1: the fused CAB
__global__ __launch_bounds__(MODE ? 512 : 256, CABF_WPS) void cabf_kernel(const CabfK P) {
    constexpr int NPB = CS / 8, PS = 16 * cabf_lds_slots(NPB), KTOT = 9 * CS, KS = (KTOT + 31) / 32;
    constexpr int XROW = CABF_XW * PS, MROW = CABF_MW * PS, CP = 16 * MT;
    constexpr int NTW = MT;              // a role's four waves = MT M-tiles x (4 / MT) groups of NTW = MT N-tiles: ONE M-tile's fragments resident per wave
    static_assert(MT == 1 || MT == 2, "wave layout: four waves per role");
    __shared__ __attribute__((aligned(16))) char lds_x[4 * XROW];
    __shared__ __attribute__((aligned(16))) char lds_m[MODE ? 4 * MROW : 16];
    __shared__ float red[MODE ? 1 : 4 * 5 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), g = lane >> 4, p = lane & 15;
    const int b = blockIdx.x, sx = b % P.nsx, sy = (b / P.nsx) % P.nsy, t = b / (P.nsx * P.nsy);
    const int x0 = sx * P.vw, Y0 = sy * P.seg, Y1 = Y0 + P.seg < P.h ? Y0 + P.seg : P.h;
    if (Y0 >= P.h) return;                                                    // workgroup-uniform
    const int h = P.h, w = P.w, seg = Y1 - Y0;
    const int NS = (seg + 3 + 3) & ~3;                                        // steps, a multiple of the ring size (the step loop is unrolled by four)
    const bool role1 = wv < 4;                                                // conv1 waves (the only ones of the statistics pass)
    const int wr = wv & 3, mt_ = wr % MT, nh = wr / MT;                       // this wave's M-tile and its group of N-tiles nh NTW .. nh NTW + NTW - 1
    const int c0 = g * 4 * MT + mt_ * 4;                                      // the lane's 4 channels of its pixel ('conv' row order of prep.pack_conv)

    for (int e = tid; e < 4 * XROW / 16; e += blockDim.x) ((uint4*)lds_x)[e] = make_uint4(0u, 0u, 0u, 0u);      // unused k-slots read row 0: keep it finite
    if constexpr (MODE != 0) for (int e = tid; e < 4 * MROW / 16; e += blockDim.x) ((uint4*)lds_m)[e] = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (!MODE) for (int e = tid; e < 4 * 5 * 16; e += blockDim.x) red[e] = 0.f;

    // ---- resident weights of this wave: one M-tile of its role's conv ----
    bf16x8_t W[KS];
    {
        const uint4* wf = role1 ? P.wfrag1 : P.wfrag2;
#pragma unroll
        for (int s = 0; s < KS; ++s) W[s] = as_frag(wf[(mt_ * KS + s) * 64 + lane]);
    }
    const float* const bp = role1 ? P.bias1 : P.bias2;
    const float4 bias = bp ? *(const float4*)(bp + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- input rows: thread i < 66 NPB of the conv1 waves moves the i-th 16-byte piece of a row (column x0 - 2 + i / NPB) ----
    const bf16_t* const xt = P.x + (size_t)t * h * w * CS;
    const int lpx = tid / NPB, lblk = tid - lpx * NPB, lgx = x0 - 2 + lpx;
    const bool loader = role1 && tid < CABF_XW * NPB;
    const bool lcol = loader && lgx >= 0 && lgx < w;
    const int lgoff = lcol ? lgx * CS + lblk * 8 : 0, lloff = lpx * PS + lblk * 16;
    auto load_row = [&](int y) -> uint4 {                                     // branch-free (clamped); the mask is applied at the LDS write
        const int yc = (y >= 0 && y < h) ? y : 0;
        return *(const uint4*)(xt + (size_t)yc * w * CS + lgoff);
    };
    auto put_row = [&](int slot, const uint4 v, int y) {
        if (loader) *(uint4*)(lds_x + slot * XROW + lloff) = (lcol && y >= 0 && y < h) ? v : make_uint4(0u, 0u, 0u, 0u);
    };
    // Rows in flight: FOUR steps ahead, in registers (X[j & 3] holds input row Y0 + 1 + j from step j - 4 on).  A step is ~300 cycles of MFMA work, a
    // load from HBM takes 1500 - 2500 under load: with the two-step distance of the first version the whole walk ran at memory latency (3000 cycles per
    // step, 1.9 TB/s), and so did the conv2 waves, which fetched their residual operands in the step that consumed them.
    uint4 X[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) X[k] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();                                                          // rings zeroed
    if (role1) {
        // ring slot of input row r = (r - (Y0 - 2)) & 3
        const uint4 r0 = load_row(Y0 - 2), r1 = load_row(Y0 - 1), r2 = load_row(Y0);
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = load_row(Y0 + 1 + k);
        put_row(0, r0, Y0 - 2); put_row(1, r1, Y0 - 1); put_row(2, r2, Y0);
    }
    // ---- per-lane constants ----
    // conv1: mid column index mc = 16 n + p  <->  image column x0 - 1 + mc; tap dx reads x ring column mc + dx
    // conv2: out column index oc = 16 n + p  <->  image column x0 + oc;      tap dx reads mid ring column oc + dx
    const int cidx0 = 16 * nh * NTW + p;                                      // + 16 i for N-tile i of the wave
    const float slope = P.prelu;
    float st[MODE ? 1 : 3][4];                                                // statistics: total, row 0, row h-1 (the two border columns go through LDS)
    if constexpr (!MODE) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[k][r] = 0.f;
    }
    float4 osc = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (MODE != 0) osc = *(const float4*)(P.ca + (size_t)t * CP + c0);
    const int nblk = P.nsx * P.nsy, blk = sy * P.nsx + sx;
    float* const partb = MODE ? nullptr : P.part + ((size_t)t * nblk + blk) * 9 * CP;
    // conv2 waves: residual operands of output row Y0 - 3 + j in RX[j & 3] / RR[j & 3], fetched four steps ahead
    bool ok2[NTW];
    int eoff2[NTW];
    uint2 RX[4][NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int oc = cidx0 + 16 * i, gx2 = x0 + oc;
        ok2[i] = oc < P.vw && gx2 < w && c0 < CS;
        eoff2[i] = ok2[i] ? gx2 * CS + c0 : 0;
    }
    const bool has_r2 = P.res2 != nullptr;                                    // workgroup-uniform
    auto load_res = [&](int yo, uint2 (&rx)[NTW]) {
        const int yc = (yo >= 0 && yo < h) ? yo : 0;
        const size_t rowoff = ((size_t)t * h + yc) * w * CS;
#pragma unroll
        for (int i = 0; i < NTW; ++i) rx[i] = *(const uint2*)(P.x + rowoff + eoff2[i]);
    };
    if (MODE != 0 && !role1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_res(Y0 - 3 + k, RX[k]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                       // vmcnt(0): weights and the first rows have landed
    __syncthreads();

    // 3x3 MFMA conv of one row for this wave's N-tiles and M-tile: ring = x ring (conv1) or mid ring (conv2), rows in slots S0, S0 + 1, S0 + 2 (mod 4)
    auto conv_row = [&](const char* ring, const int rowb, auto s0tag, f32x4_t (&acc)[NTW]) {
        constexpr int S0 = decltype(s0tag)::value;
        int gl = g;
        asm volatile("" : "+v"(gl));       // opaque: the 4 x KS tap offsets of the unrolled steps are recomputed (3 selects each) instead of kept in 28 registers
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = (f32x4_t){bias.x, bias.y, bias.z, bias.w};
        // lane group g reads k-slots [(4 s + g) 8, + 8) = 8 channels from cc0 of tap (dy, dx); fragments in batches of two k-steps
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += 2) {
            uint4 bq[2][NTW];
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const int s = s0 + si;
                if (s >= KS) continue;
                int toff = 0;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int kk0 = (s * 4 + gg) * 8;
                    const int tap = kk0 / CS, cc0 = kk0 - tap * CS, dy = tap / 3, dx = tap - dy * 3;
                    const int o = kk0 < KTOT ? ((S0 + dy) & 3) * rowb + dx * PS + cc0 * 2 : 0;
                    toff = gl == gg ? o : toff;
                }
#pragma unroll
                for (int i = 0; i < NTW; ++i) bq[si][i] = *(const uint4*)(ring + (cidx0 + 16 * i) * PS + toff);
            }
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const int s = s0 + si;
                if (s >= KS) continue;
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = mfma16(W[s], as_frag(bq[si][i]), acc[i]);
            }
        }
    };

    auto step1 = [&](const int j, auto jtag) {                                // a step of a conv1 wave
        constexpr int J = decltype(jtag)::value;                              // j & 3
        {
            const int ym = Y0 - 1 + j;                                        // mid row of this step: input rows ym - 1 .. ym + 1 = ring slots J, J + 1, J + 2
            if (j <= seg + 1) {
                f32x4_t acc[NTW];
                conv_row(lds_x, XROW, std::integral_constant<int, J>{}, acc);
                const bool rin = ym >= 0 && ym < h;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int gx1 = x0 - 1 + cidx0 + 16 * i;
                    const bool keep = rin && gx1 >= 0 && gx1 < w;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = acc[i][r];
                        const float q = (slope >= 0.f && slope <= 1.f) ? fmaxf(a, slope * a) : fmaf(slope, fminf(a, 0.f), fmaxf(a, 0.f));
                        v[r] = keep ? q : 0.f;                                // conv2's zero padding: mid is zero outside the image
                    }
                    if constexpr (MODE != 0) {
                        if (c0 < CS) *(uint2*)(lds_m + J * MROW + (cidx0 + 16 * i) * PS + c0 * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    } else {
                        const float rown = (ym >= Y0 && ym < Y1 && gx1 >= x0 && gx1 < x0 + P.vw && gx1 < w) ? 1.f : 0.f;      // every mid pixel counted by exactly one workgroup
                        const float f0 = ym == 0 ? rown : 0.f, f1 = ym == h - 1 ? rown : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            st[0][r] = fmaf(v[r], rown, st[0][r]); st[1][r] = fmaf(v[r], f0, st[1][r]); st[2][r] = fmaf(v[r], f1, st[2][r]);
                        }
                        // first / last image column: ONE lane group of ONE wave per (workgroup, M-tile) holds such a pixel; it adds its values row by
                        // row (program order: reproducible) to its own slots of `red`
                        if (rown != 0.f && (gx1 == 0 || gx1 == w - 1)) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (gx1 == 0) red[(wr * 5 + 3) * 16 + g * 4 + r] += v[r];
                                if (gx1 == w - 1) red[(wr * 5 + 4) * 16 + g * 4 + r] += v[r];
                            }
                            if (ym == 0 || ym == h - 1) {                     // a corner pixel (a one-row / one-column frame has coinciding corners)
#pragma unroll
                                for (int ck = 0; ck < 4; ++ck)
                                    if (ym == ((ck & 2) ? h - 1 : 0) && gx1 == ((ck & 1) ? w - 1 : 0)) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) partb[(5 + ck) * CP + c0 + r] = v[r];
                                    }
                            }
                        }
                    }
                }
            }
            // input row Y0 + 1 + j (loaded four steps ago) -> slot (j + 3) & 3; refill the register with row Y0 + 5 + j
            put_row((J + 3) & 3, X[J], Y0 + 1 + j);
            X[J] = load_row(Y0 + 5 + j < Y1 + 2 ? Y0 + 5 + j : Y1 + 1);
        }
        __syncthreads();
    };
    auto step2 = [&](const int j, auto jtag) {                                // a step of a conv2 wave
        constexpr int J = decltype(jtag)::value;
        if constexpr (MODE != 0) {
            const int yo = Y0 - 3 + j;                                        // output row: mid rows yo - 1 .. yo + 1 = ring slots J + 1, J + 2, J + 3
            if (yo >= Y0 && yo < Y1) {
                const size_t rowoff = ((size_t)t * h + yo) * w * CS;
                uint2 r2v[NTW];                                               // the second residual (only the last CAB of a TFR_UNet has one): fetched in place
#pragma unroll
                for (int i = 0; i < NTW; ++i) r2v[i] = has_r2 ? *(const uint2*)(P.res2 + rowoff + eoff2[i]) : make_uint2(0u, 0u);
                f32x4_t acc[NTW];
                conv_row(lds_m, MROW, std::integral_constant<int, (J + 1) & 3>{}, acc);
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const uint2 rx = RX[J][i], rr2 = r2v[i];
                    const float o0 = acc[i][0] * osc.x + bf_lo(rx.x) + bf_lo(rr2.x), o1 = acc[i][1] * osc.y + bf_hi(rx.x) + bf_hi(rr2.x);
                    const float o2 = acc[i][2] * osc.z + bf_lo(rx.y) + bf_lo(rr2.y), o3 = acc[i][3] * osc.w + bf_hi(rx.y) + bf_hi(rr2.y);
                    if (ok2[i]) *(uint2*)(P.out + rowoff + eoff2[i]) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                }
            }
            load_res(yo + 4 < Y1 ? yo + 4 : Y1 - 1, RX[J]);            // (on every path: the compiler's s_waitcnt counts assume the fewest younger operations)
        }
        __syncthreads();
    };
    // one loop per role: the prefetch registers of one role are not live in the other's loop (both loops pass the same NS barriers)
    if (role1) {
#pragma unroll 1
        for (int j = 0; j < NS; j += 4) {
            step1(j, std::integral_constant<int, 0>{});
            step1(j + 1, std::integral_constant<int, 1>{});
            step1(j + 2, std::integral_constant<int, 2>{});
            step1(j + 3, std::integral_constant<int, 3>{});
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < NS; j += 4) {
            step2(j, std::integral_constant<int, 0>{});
            step2(j + 1, std::integral_constant<int, 1>{});
            step2(j + 2, std::integral_constant<int, 2>{});
            step2(j + 3, std::integral_constant<int, 3>{});
        }
    }
    if constexpr (!MODE) {
        // sums of this (frame, strip, segment): 16 lanes of a DPP row -> the waves of an M-tile -> partial [k][channel]
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sm = row_sum16(st[k][r]);
                if (p == 0) red[(wr * 5 + k) * 16 + g * 4 + r] = sm;
            }
        __syncthreads();
        if (tid < 5 * CP) {                                                   // channel ch = 4 MT g + 4 m + r  <-  waves wr = m + MT nh, slot 4 g + r
            const int k = tid / CP, ch = tid - k * CP, gq = ch / (4 * MT), m = (ch / 4) % MT, r = ch & 3;
            float sm = 0.f;
#pragma unroll
            for (int q = 0; q < 4 / MT; ++q) sm += red[((m + MT * q) * 5 + k) * 16 + gq * 4 + r];
            partb[k * CP + ch] = sm;
        }
    }
}

// CALayer of a CAB from the partial sums of `mid` (cf. cab_ca_kernel of csrc/sn_conv.hip, which reads the border lines from the stored tensor):
// pooled res[co] = (1 / hw) sum_ci sum_tap w2[ci][tap][co] * S_tap[ci], S_tap = total - excluded row - excluded column + excluded corner.
__global__ __launch_bounds__(1024) void cabf_ca_kernel(const float* part, int nsx, int nsy, int cpad, int c, int cr, int h, int w,
                                                       const float* w2, int w2pad, const float* wa, const float* wb, float* ca) {
    __shared__ float acc[1024];
    __shared__ float S[9][128];      // 0 total, 1 row 0, 2 row h-1, 3 column 0, 4 column w-1, 5..8 corners (0,0) (0,w-1) (h-1,0) (h-1,w-1)
    __shared__ float mean[128];
    __shared__ float hid[128];
    const int t = blockIdx.x, tid = threadIdx.x, nblk = nsx * nsy;
    const float* pt = part + (size_t)t * nblk * 9 * cpad;
    if (tid < 5 * cpad) {                                                     // fixed summation order: bit-reproducible
        const int k = tid / cpad, ch = tid - k * cpad;
        float m = 0.f;
        for (int bb = 0; bb < nblk; ++bb) m += pt[((size_t)bb * 9 + k) * cpad + ch];
        S[k][ch] = m;
    } else if (tid >= 512 && tid < 512 + 4 * cpad) {                          // corners: owned by the workgroups of the four corner (strip, segment) pairs
        const int k = (tid - 512) / cpad, ch = (tid - 512) - k * cpad;
        const int bb = (k & 2 ? (nsy - 1) * nsx : 0) + (k & 1 ? nsx - 1 : 0);
        S[5 + k][ch] = pt[((size_t)bb * 9 + 5 + k) * cpad + ch];
    }
    __syncthreads();
    {
        const int nsplit = 1024 / w2pad, co = tid % w2pad, part_ = tid / w2pad;
        float r = 0.f;
        if (part_ < nsplit)
            for (int cin = part_; cin < c; cin += nsplit) {
                const float tot = S[0][cin], r0 = S[1][cin], r1 = S[2][cin], cc0 = S[3][cin], cc1 = S[4][cin];
                const float k00 = S[5][cin], k01 = S[6][cin], k10 = S[7][cin], k11 = S[8][cin];
                const float* wk = w2 + (size_t)cin * 9 * w2pad + co;
                // tap (ky, kx) reads mid(p + (ky-1, kx-1)): dy = +1 cannot reach row 0, dy = -1 cannot reach row h-1, same for columns
                const float rowex[3] = {r1, 0.f, r0}, colex[3] = {cc1, 0.f, cc0};
                const float cor[3][3] = {{k11, 0.f, k10}, {0.f, 0.f, 0.f}, {k01, 0.f, k00}};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) r += wk[(ky * 3 + kx) * w2pad] * (tot - rowex[ky] - colex[kx] + cor[ky][kx]);
            }
        acc[tid] = r;
        __syncthreads();
        if (tid < w2pad) {
            float m = 0.f;
            for (int q = 0; q < nsplit; ++q) m += acc[q * w2pad + tid];
            if (tid < 128) mean[tid] = m / ((float)h * (float)w);
        }
    }
    __syncthreads();
    if (tid < cr) {
        float hh = 0.f;
        for (int j = 0; j < c; ++j) hh += wa[tid * c + j] * mean[j];
        hid[tid] = hh > 0.f ? hh : 0.f;
    }
    __syncthreads();
    if (tid < cpad) {
        float o = 0.f;
        if (tid < c) {
            for (int j = 0; j < cr; ++j) o += wb[tid * cr + j] * hid[j];
            o = sigmoidf_(o);
        }
        ca[(size_t)t * cpad + tid] = o;
    }
}

// strips of <= 62 own columns, row segments so that ~two workgroups per CU are busy for as few (segment + 3)-step rounds as possible
void cabf_partition(int T, int h, int w, int ncu, int& nsx, int& vw, int& nsy, int& seg) {
    nsx = (w + CABF_OWN - 1) / CABF_OWN;
    vw = (w + nsx - 1) / nsx;
    long best = -1;
    nsy = 1;
    for (int cand = 1; cand <= (h + 7) / 8; ++cand) {
        const int sg = (h + cand - 1) / cand;
        if ((sg * (cand - 1)) >= h) continue;
        const long items = (long)T * nsx * cand, rounds = (items + 2L * ncu - 1) / (2L * ncu);
        const long cost = rounds * (sg + 3);
        if (best < 0 || cost < best) { best = cost; nsy = cand; }
    }
    seg = (h + nsy - 1) / nsy;
}

int cabf_ncu() {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) return -1;
    return ncu;
}

template <int MODE>
int cabf_launch(const CabfK& K, int mt, int cs, hipStream_t st) {
    const dim3 grid((unsigned)(K.T * K.nsx * K.nsy)), blk(MODE ? 512 : 256);
    sn_clear_error();
    if (mt == 1 && cs == 16) hipLaunchKernelGGL((cabf_kernel<1, 16, MODE>), grid, blk, 0, st, K);
    else if (mt == 2 && cs == 24) hipLaunchKernelGGL((cabf_kernel<2, 24, MODE>), grid, blk, 0, st, K);
    else return SN_EINVAL;
    return sn_check_launch();
}

}  // namespace

extern "C" {

int sn_cabf_supported(int c) { return (c >= 9 && c <= 16) || (c >= 17 && c <= 24) ? 1 : 0; }

int sn_cabf_blocks(int T, int h, int w) {
    const int ncu = cabf_ncu();
    if (ncu < 1 || T < 1 || h < 1 || w < 1) return SN_EINVAL;
    int nsx, vw, nsy, seg;
    cabf_partition(T, h, w, ncu, nsx, vw, nsy, seg);
    return nsx * nsy;
}

static int cabf_fill(CabfK& K, const sn_cabf_desc* d) {
    if (!d || !d->x || d->T < 1 || d->h < 1 || d->w < 1 || !d->wfrag1 || (d->cs != 16 && d->cs != 24) || d->mt != d->cs / 16 + (d->cs % 16 ? 1 : 0)) return SN_EINVAL;
    const int ncu = cabf_ncu();
    if (ncu < 1) return SN_ELAUNCH;
    K.x = (const bf16_t*)d->x; K.res2 = (const bf16_t*)d->res2; K.out = (bf16_t*)d->out;
    K.wfrag1 = (const uint4*)d->wfrag1; K.wfrag2 = (const uint4*)d->wfrag2; K.bias1 = d->bias1; K.bias2 = d->bias2;
    K.prelu = d->prelu; K.ca = d->ca; K.part = d->part; K.T = d->T; K.h = d->h; K.w = d->w;
    cabf_partition(d->T, d->h, d->w, ncu, K.nsx, K.vw, K.nsy, K.seg);
    return SN_OK;
}

int sn_cabf_stats(const sn_cabf_desc* d, void* stream) {
    CabfK K;
    const int rc = cabf_fill(K, d);
    if (rc) return rc;
    if (!d->part) return SN_EINVAL;
    return cabf_launch<0>(K, d->mt, d->cs, (hipStream_t)stream);
}

int sn_cabf_ca(const float* part, int T, int h, int w, int cpad, int c, int cr, const float* w2, int w2pad, const float* wa, const float* wb,
               float* ca, void* stream) {
    sn_clear_error();
    const int ncu = cabf_ncu();
    if (!part || !w2 || !wa || !wb || !ca || T < 1 || cpad < 16 || cpad > 32 || c < 1 || c > cpad || cr < 1 || cr > 128 || w2pad < cpad || w2pad > 128 || ncu < 1)
        return SN_EINVAL;
    int nsx, vw, nsy, seg;
    cabf_partition(T, h, w, ncu, nsx, vw, nsy, seg);
    hipLaunchKernelGGL(cabf_ca_kernel, dim3(T), dim3(1024), 0, (hipStream_t)stream, part, nsx, nsy, cpad, c, cr, h, w, w2, w2pad, wa, wb, ca);
    return sn_check_launch();
}

int sn_cabf(const sn_cabf_desc* d, void* stream) {
    CabfK K;
    const int rc = cabf_fill(K, d);
    if (rc) return rc;
    if (!d->wfrag2 || !d->ca || !d->out || d->out == d->x) return SN_EINVAL;
    return cabf_launch<1>(K, d->mt, d->cs, (hipStream_t)stream);
}

}  // extern "C"
