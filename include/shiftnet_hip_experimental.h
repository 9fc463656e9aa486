/* shiftnet_hip_experimental.h -- entry points that exist ONLY in a -DSN_EXPERIMENTAL build of the library
 * (python shift-net_amd/build.py --experimental -> lib/libshiftnet_hip_exp.so).  None of them is on the production path:
 * they are earlier generations of the GSTS kernels kept for A/B measurements, an opt-in matrix-core variant that is not
 * faster yet, and process-global profiling switches (the production library is stateless, shiftnet_hip.h).
 */
#ifndef SHIFTNET_HIP_EXPERIMENTAL_H
#define SHIFTNET_HIP_EXPERIMENTAL_H
#include "shiftnet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- round-1 five-kernel chain (every intermediate NHWC bf16 in HBM) ---- */
/* a = body[0](norm(cat(shortcut, hw))): LayerNorm2d over 3C/2 (CAB2) or C (CAB1) channels, eps 1e-6, affine folded
 * into the 1x1 weights, then the 1x1 conv to 2C (gshift_deblur1.py:19-28,190,225,252).  a:[T][h][w][2C] in
 * "gate-paired" position order (prep.py).  hw may be NULL for mode 0. */
int sn_ln_gemm(const sn_unit_src* s, const void* hw, const void* wfrag, const float* bias, void* a, void* stream);

/* g1 = SimpleGate(RepConv2(a)) (gshift_deblur1.py:166-178): (a1 + dw3x3(a1)) * (a2 + dw3x3(a2)); w:[9][2C] f32 in
 * a's position order with the identity folded into the centre tap.  g1:[T][h][w][C] natural order.
 * pool: NULL, or [T][sn_dwgate_blocks][C] per-workgroup sums of g1 (denoise CALayer2, gshift_denoise1.py:224). */
int sn_dw_gate(const void* a, const float* w, void* g1, float* pool, int T, int h, int w_, int C, void* stream);
int sn_dwgate_blocks(int h, int w);

/* g2 = SimpleGate2(body[4](RepConv(g1))) (gshift_deblur2.py:159-168,182-185,201): depthwise 5x5 + 3x3 + identity
 * (folded into one 5x5, w5:[25][C] f32), 1x1 C->2C on MFMA, x1*sigmoid(x2); plus per-workgroup channel sums for
 * the CALayer2 that follows.  ca_in: NULL or [T][C] f32 scale applied to g1 first (denoise).  g2:[T][h][w][C]. */
int sn_dw_gemm_gate(const void* g1, const float* ca_in, const float* w5, const void* wfrag, void* g2, float* pool,
                    int T, int h, int w, int C, void* stream);
int sn_dwgemm_blocks(int h, int w);


/* ---- K3': LDS-staged VALU 5x5, superseded by sn_dw5m_gemm_gate; profiling switches ---- */
/* sn_dw_gemm_gate for the depthwise variants (C = 64) with the channel-blocked g1 tile staged through LDS; pool: [T][sn_dw5_blocks][C]. */
int sn_dw5_blocks(int h, int w);
/* Profiling aids (tools/ only; process-global, default 0 / NULL = production behaviour):
 *   sn_debug_set(mask): bits 1,2,4,8 (sn_dw5_gemm_gate / sn_dw5m_gemm_gate) and 8,16,32 (sn_ln_gemm_gate) skip a phase of
 *     the kernel (results are then wrong) for ablation timing; bit 256 / 512 make sn_ln_gemm_gate(_m) / sn_dw5m_gemm_gate
 *     write per-wave s_memtime phase accumulators ([workgroup][8 waves][8 slots] u64) to the buffer set below.
 *   sn_debug_buf_set(dev_ptr): device buffer for those accumulators (tools/prof_k12.py, tools/prof_k3m.py). */
int sn_debug_set(int v);
int sn_debug_get(void);
int sn_debug_buf_set(void* dev_ptr);
void* sn_debug_buf_get(void);
int sn_dw5_gemm_gate(const void* g1, const float* ca_in, const uint32_t* w5, const void* wfrag, void* g2, float* pool,
                     int T, int h, int w, int C, void* stream);


/* ---- K12m: sn_ln_gemm_gate with the depthwise 3x3 on the matrix cores ---- */
/* Same operator as sn_ln_gemm_gate (LayerNorm2d -> body[0] 1x1 -> RepConv2 -> SimpleGate, gshift_deblur1.py:19-28,190-198)
 * for C = 64 with the depthwise 3x3 as Toeplitz MFMAs and g1 written channel-planar [T][h][C][sn_planar_pitch(w)].
 * wfrag / bias: as for sn_ln_gemm_gate; ttab3: bf16 [C/16][32][3][2][20] band records (prep.pack_toeplitz_dw3_chunks).
 * pool: NULL or [T][sn_lngatem_blocks(h,w)][C] per-workgroup sums of g1 (denoise CALayer2). */
int sn_lngatem_blocks(int h, int w);
int sn_ln_gemm_gate_m(const sn_unit_src* s, const void* hw, const void* wfrag, const float* bias, const void* ttab3,
                      void* g1p, float* pool, void* stream);


/* ---- fused CAB (mid in LDS): parity green, slower than two sn_conv2d launches on MI355X (note in csrc/sn_conv.hip) ---- */
/* Fused CAB, pass B (gshift_deblur1.py:141-156): out = x + ca * conv2(PReLU(conv1(x))) [+ res2] with mid kept in LDS.  d describes
 * conv1 (in[0] = x, wfrag = conv1 fragments, act = 1 / prelu, bias NULL, oscale = ca from sn_cab_ca, res2 optional, out); wfrag2 =
 * conv2 fragments (same mt / ks).  Storage widths 16, 24, 40, 48; wider CABs run as two sn_conv2d calls.  Three tensor passes per
 * CAB (pass A reads x, pass B reads x and writes out) instead of five. */
int sn_cab_fused(const sn_conv_desc* d, const void* wfrag2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
