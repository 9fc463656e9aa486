/* shiftnet_hip.h -- C ABI of libshiftnet_hip.so: the MI355X (gfx950) kernels behind GShiftNet.forward.
 *
 * The reference has no native code and no FFI: its "operators" are the ATen calls made by
 * basicsr/models/archs/gshift_{deblur,denoise}{1,2}.py.  This header is the boundary a maintainer binds instead
 * (ctypes stub: INTEGRATION.md).  Each entry point names the reference expression it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless said otherwise; the caller owns every buffer (outputs and workspaces
 *     included), the library never allocates or frees, reads no environment variable and keeps no global mutable
 *     state (thread compatible).  Kernels launch on the calling thread's CURRENT HIP device: make the device that owns
 *     `stream` and the buffers current first (the Python engine wraps every forward in torch.cuda.device(dev));
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and returns immediately;
 *   - return 0 on success, negative errno-style code otherwise (-22 bad argument, -5 launch failure); nothing throws;
 *   - activations: NHWC bf16 [T][H][W][Cs], Cs a multiple of 8, pad channels must be (and are kept) zero;
 *   - "wfrag" arguments are weights prepacked by the host into MFMA A-fragment order (shiftnet_amd/prep.py):
 *     bf16 [MT][KS][64 lanes][8], see csrc/sn_common.h for the lane/slot convention.
 */
#ifndef SHIFTNET_HIP_H
#define SHIFTNET_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_ABI_VERSION 17     /* bump on ANY change of a struct, signature or operand encoding (shiftnet_amd/lib.py checks it) */

/* element types of NCHW tensors exchanged with the PyTorch side */
#define SN_F32 0
#define SN_F16 1
#define SN_BF16 2

int sn_abi_version(void);

/* MFMA lane-layout self test: D = A.B on one 16x16x32 tile with asymmetric operands written in the slot
 * convention of sn_common.h.  a:[16][32] f32, b:[32][16] f32 row major, d:[16][16] f32 out (all device). */
int sn_selftest_mfma(const float* a, const float* b, float* d, void* stream);

/* x = x[0]; torch.cat((x, noise_map), 1) and the implicit NCHW->kernel layout change
 * (gshift_deblur1.py:784-787, gshift_denoise1.py:828-831).  src:[T][C][H][W] of `src_dtype`, noise:[T][1][H][W]
 * or NULL, dst:[T][H][W][8] bf16 (channels C(+1)..7 zeroed). */
int sn_ingest(const void* src, int src_dtype, const void* noise, void* dst, int T, int C, int H, int W, void* stream);

/* One dense nn.Conv2d (+ what the reference applies around it), as an implicit GEMM on MFMA. */
typedef struct sn_conv_desc {
    const void* in[3];   /* 1..3 NHWC inputs of equal shape, concatenated along channels (torch.cat(...,1)):
                            rconcat gshift_deblur1.py:772, conv_hr0 :640 */
    int n_in;
    int cs_in;           /* storage channels of each input */
    int T, h_in, w_in;   /* spatial size the convolution sees */
    int in_mode;         /* 0: as is; 1: inputs are [T][h_in/2][w_in/2] and are bilinearly upsampled x2,
                            align_corners=False, while staging (SkipUpSample.up[0], gshift_deblur1.py:344) */
    int k, stride, pad;  /* k in {1,2,3,5}; Conv2d zero padding */
    int h_out, w_out;
    const void* wfrag;   /* [MT][KS][64][8] bf16 */
    int mt, ks;
    const float* bias;   /* [16*mt] natural channel order, zero padded; NULL = no bias */
    int act;             /* 0 none, 1 PReLU with the single shared slope `prelu` (nn.PReLU(), :551,576) */
    float prelu;
    const void* res;     /* NHWC [T][h_out][w_out][cs_out] added after the activation, or NULL
                            ("x = self.up(x); x = x + y" :348-349, "+ self.skip_conv(shortcut)" gshift_deblur2.py:611) */
    void* out;
    int cs_out;
    int out_mode;        /* 0: NHWC; 1: F.pixel_shuffle(.,2) store -> [T][2h][2w][cs_out] (PixelShufflePack :277) -- the weight ROWS (and bias) of such
                            a conv are ordered [sub-pixel 2 i + j][output channel c < cs_out] (reference row 4 c + 2 i + j; zero rows for the storage
                            padding), cs_out = 4 mt, so lane group g of the accumulator layout holds every channel of sub-pixel g (ABI 16; prep.pack_conv);
                            2: NCHW [T][c_out][h][w] of `nchw_dtype` plus the NCHW shortcut `sc`
                               ("return output_features + shortcut[...]" :791) */
    int c_out;           /* logical out channels (mode 2 only) */
    int nchw_dtype;      /* SN_F32 with a half-precision module = the restored frame straight from the fp32 accumulators: what the CLIs'
                            metrics and PNG writer consume (inference/test_deblur.py:137-143 calls .float() on the module output) */
    const void* sc;      /* mode 2: NCHW tensor of `sc_dtype`, same shape as out */
    int sc_dtype;
    float* pool;         /* NULL or [T][gridDim.y*gridDim.x][16*mt] f32 per-workgroup channel sums of the output
                            (first half of AdaptiveAvgPool2d(1), CALayer :69); rows per frame = sn_conv_pool_blocks(d) */
    const float* oscale; /* NULL or [T][oscale_stride] f32: out = conv * oscale[t][c] (+ res) -- the CALayer scale of a CAB
                            applied in the epilogue of its second conv ("res = self.CA(res); res += x", :155-157) */
    int oscale_stride;
    const void* res2;    /* NULL or a second NHWC residual of the same shape as `res`, added after it: the "+ shortcut" that
                            follows the last TFR_UNet of a stage (gshift_deblur1.py:769,779) rides on that UNet's last conv */
    int flags;           /* SN_CONV_TILE_KERNEL: run single-input 3x3 stride-1 convs on the one-workgroup-per-tile kernel instead of the
                            persistent streaming kernel (csrc/sn_conv3p.hip) -- A/B measurements; results are bit-identical.
                            Bits 4..7: persistent workgroups per CU of the streaming kernel, 0 = the library's choice; bit 8: the streaming kernel also
                            where the library prefers the tile kernel; bit 9: its residual operand through registers instead of LDS; bits 10..11:
                            prefetch depth code (1 / 2: three / four tiles ahead at 16 channels; 3: the streaming fused CAB with two region buffers);
                            bits 12..14: MEASUREMENTS ONLY, WRONG RESULTS -- the streaming conv without its DMA (1), B reads / MFMAs (2), stores (4);
                            on conv2 of sn_cab_fused(rows = 0) bit 12 means "both weight sets in LDS" (results unchanged); bit 15: stride-2 convs on
                            4 x 16 tiles whatever their width (measurements; results unchanged) */
} sn_conv_desc;
#define SN_CONV_TILE_KERNEL 1
int sn_conv2d(const sn_conv_desc* d, void* stream);   /* d is a HOST pointer, read during the call */
/* number of workgroups per frame sn_conv2d launches for this descriptor (= rows of `pool` per frame); host only */
int sn_conv_pool_blocks(const sn_conv_desc* d);

/* SkipUpSample's tail (gshift_deblur1.py:341-350: x = up(x); x = x + y with up = bilinear x2 -> 1x1 conv).  The 1x1 is linear and the interpolation
 * weights sum to one, so conv(up(x)) = up(conv(x)): the caller runs the 1x1 at LOW resolution with sn_conv2d (in_mode 0) and this pass computes
 *   out[T][2 hs][2 ws][cs] = bilinear_x2(lo[T][hs][ws][cs]) + res[T][2 hs][2 ws][cs]      (bf16 NHWC, fp32 arithmetic, cs a multiple of 8;
 * nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)).  sn_conv2d's in_mode 1 (interpolation in the conv's loader) stays available. */
int sn_upsample2_add(const void* lo, const void* res, void* out, int T, int hs, int ws, int cs, void* stream);

/* CALayer / CALayer2 squeeze-excite: mean -> 1x1 -> ReLU -> 1x1 -> sigmoid (gshift_deblur1.py:61-70,84-87).
 * partial:[T][nblk][cpad] f32 sums, wa:[cr][c], wb:[c][cr] f32, ca:[T][cpad] f32 out (pad entries 0).
 * bad: NULL, or one u32 that is set to 1 when a channel sum is not finite -- the range guard of the half-precision intermediates of the
 * producer (an overflowed fp16 value upstream turns its channel sum into inf / NaN); the caller zeroes it and reads it back when it likes. */
int sn_ca_mlp(const float* partial, int nblk, int cpad, int c, int cr, float inv_hw,
              const float* wa, const float* wb, float* ca, int T, unsigned* bad, void* stream);

/* CALayer of a CAB computed BEFORE its second conv runs: the pooled mean of res = conv2(mid) is linear in mid,
 *   sum_p res[co](p) = sum_ci sum_tap W2[co][ci][tap] * S_tap[ci],   S_tap = sum of mid over the pixels the tap can reach
 *   (total - excluded border row / column + corner), so it follows from the channel sums of mid (partial, from conv1's
 * epilogue) and the first/last rows and columns of mid.  Then the usual 1x1 -> ReLU -> 1x1 -> sigmoid (:61-70).
 * mid:[T][h][w][cs] bf16, w2:[cin=c][9][cpad] f32 (bias-free 3x3, zero beyond c), ca:[T][cpad] out.  Lets sn_conv2d apply the scale and the
 * residual in conv2's epilogue (no separate pass over res).
 * scratch: sn_cab_ca_scratch_floats(T) floats of workspace (partial sums of the split reduction). */
int sn_cab_ca_scratch_floats(int T);
int sn_cab_ca(const float* partial, int nblk, int cpad, const void* mid, int cs, int c, int cr, int h, int w,
              const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream);
/* ---- fused dense CAB (csrc/sn_cabf.hip): CAB.forward, gshift_deblur1.py:141-156, in three tensor passes instead of five ----
 *   res = body(x) = conv3x3(PReLU(conv3x3(x)));  res = CA(res);  res += x
 * The two-launch form (sn_conv2d with `pool`, sn_cab_ca, sn_conv2d with oscale / res) writes `mid` = PReLU(conv1(x)) to memory and reads it
 * back.  The fused form never stores it:
 *   1. sn_cab_stats(conv1, lines_len): conv1 exactly as sn_conv2d would run it -- same tiles, same per-workgroup channel sums into conv1->pool --
 *      but conv1->out is the LINE buffer [T][4][lines_len][cs] bf16 (lines_len >= max(h, w)): row 0, row h-1, column 0, column w-1 of mid;
 *   2. sn_cab_ca_lines: sn_cab_ca reading those lines instead of the tensor;
 *   3. sn_cab_fused(conv1, conv2, tile_rows): per (tile_rows x 32)-pixel tile conv1 + PReLU on the tile's 1-pixel ring into LDS, conv2 from there,
 *      * conv2->oscale, + x (conv2->res must be conv1->in[0]), + conv2->res2, stored to conv2->out.  tile_rows: 8 or 16 = one workgroup per
 *      tile (csrc/sn_cabf.hip); 0 = the STREAMING form (csrc/sn_conv3p.hip: cabp_kernel -- persistent workgroups, a loader wave that moves the next
 *      (8 + 4) x 36 pixel regions HBM -> LDS with LDS-DMA, four compute waves: conv1 on the ring -> mid in LDS -> conv2 -> scale + x -> store;
 *      no res2; its statistics pass is sn_cab_stats on a descriptor WITHOUT SN_CONV_TILE_KERNEL, i.e. the streaming conv's own pool rows).
 * Both descriptors are those of the two-launch form (3x3, stride 1, pad 1, one input, cs_in == cs_out, NHWC); conv1->pool / conv1->out are only
 * read by sn_cab_stats.  Results are bit-identical to the two-launch form.  sn_cab_fused_supported: 1 when an instance for the pair exists
 * (16- and 24-channel storage), else 0 -- the caller then uses the two-launch form. */
int sn_cab_fused_supported(const sn_conv_desc* conv1, const sn_conv_desc* conv2);
int sn_cab_stats(const sn_conv_desc* conv1, int lines_len, void* stream);
int sn_cab_ca_lines(const float* partial, int nblk, int cpad, const void* lines, int lines_len, int cs, int c, int cr, int h, int w,
                    const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream);
int sn_cab_fused(const sn_conv_desc* conv1, const sn_conv_desc* conv2, int tile_rows, void* stream);

/* the same for the fp32 engine: mid [T][h][w][c] float32 (pixel stride c), partial = sn32_chan_sum(mid), w2 [c][9][cpad] f32: the CAB's scale and
 * residual then ride on the second sn32_conv2d (oscale / res) instead of a pass of their own (sn32_scale_residual). */
int sn32_cab_ca(const float* partial, int nblk, int cpad, const float* mid, int c, int cr, int h, int w,
                const float* w2, const float* wa, const float* wb, float* scratch, float* ca, int T, void* stream);

/* ---- grouped spatial-temporal shift unit: channel_shift -> CAB2 -> CAB1 (gshift_deblur1.py:504-547) ---- */
typedef struct sn_unit_src {
    const void* x;       /* [T][h][w][C] the unit's input */
    int T, h, w, C;      /* C in {64, 80} */
    int mode;            /* 0: CAB1 (no shift, u = x[t]); 1: CAB2 of a forward unit; 2: CAB2 of a reverse unit */
    int wrap;            /* 0: boundary frame kept (gshift_deblur1.py:513,517); 1: circular temporal roll (gshift_deblur2.py:504-505);
                            2: x is a frame range of a temporally split window and the neighbour of its boundary frame (frame -1 for a
                            forward unit, T for a reverse unit) belongs to the adjacent rank: only the half the unit borrows exists
                            here, in `halo` */
    const void* halo;    /* wrap == 2, mode 1 / 2: the neighbour frame's borrowed half-channels, CONTIGUOUS [h][w][C/2] (same element type
                            as x): forward units the upper half of the previous rank's last frame, reverse units the lower half of the
                            next rank's first frame -- the receive buffer of the halo exchange itself, no copy into a strided slot */
    int t0, nt;          /* frames [t0, t0 + nt) of x are processed (nt == 0: all T).  Outputs, pool rows and the neighbour rule are indexed
                            by the absolute frame, so a unit can be launched in pieces: the frames that need no halo while the exchange is
                            in flight, the boundary frame after it */
} sn_unit_src;

/* validation op: materialise u = cat(y, spatial_shift2(hw)) : [T][h][w][3C/2] exactly as channel_shift returns it
 * (gshift_deblur1.py:504-528).  offs: int8 [C/2][2] (dy,dx) of the source pixel.  Pure index work: bit exact. */
int sn_gsts_gather(const sn_unit_src* s, const int8_t* offs, void* u, void* stream);

/* y = channel_shift(x) of Shift_CAB (gshift_denoise1.py:167-179): the temporal half-channel roll alone, materialised
 * (the 24-/80-channel Shift_CAB encoders feed a dense 3x3 CAB, which reads y as an ordinary tensor).
 * C any even channel count whose halves are whole elements; mode 1 forward / 2 reverse. */
int sn_temporal_roll(const sn_unit_src* s, void* y, void* stream);

/* hw = CAB2.conv1(spatial_shift2(borrowed half)) (gshift_deblur1.py:470-503,223,251): depthwise 3x3 of the
 * zero-padded displaced neighbour-frame channels, never materialising the shifted tensor.
 * w1:[C/2][9] u32 words: the bf16 weight in the LOW half, high half zero (operand of v_dot2c_f32_bf16). */
int sn_gsts_shiftconv(const sn_unit_src* s, const int8_t* offs, const uint32_t* w1, void* hw, void* stream);
/* The same operator on the matrix cores (round 6), same operands: per channel and 16 x 16 output tile the depthwise 3x3 is the banded GEMM
 *   out[n][m] = sum_{ty, j} A[m][(ty, j)] * W[n + ty + 8 + dy_k][j + 8 + dx_k],   A[m][(ty, j)] = w[ty][j - m] for 0 <= j - m <= 2,
 * over a CHANNEL-PLANAR 34 x 34 LDS window (the loader transposes; hw and x stay NHWC): three 16-byte LDS reads + three MFMAs per channel and tile
 * instead of 9 two-byte reads + 9 v_dot2c per pixel and channel; the A fragments are built in registers from the nine weights (one v_perm_b32 per
 * word).  Every displacement in offs must be a multiple of 4 pixels (spec.shift_table: they are).  Same products, another accumulation order:
 * within rounding of sn_gsts_shiftconv, not bit-identical to it. */
int sn_gsts_shiftconv_mfma(const sn_unit_src* s, const int8_t* offs, const uint32_t* w1, void* hw, void* stream);

/* g1 = SimpleGate(RepConv2(body[0](norm(cat(shortcut, hw))))): LayerNorm2d over 3C/2 (CAB2) or C (CAB1) channels (eps 1e-6,
 * affine folded into the 1x1 weights), the 1x1 conv to 2C on MFMA, depthwise 3x3 + identity and the gate, with the 2C-channel
 * intermediate kept in LDS (gshift_deblur1.py:19-28,190-198,225-233).  wfrag / bias: prep.pack_ln_gemm (gate-paired rows).
 * hw may be NULL for mode 0.
 * wdw: [9][C] u32: word k of a tap row holds the fp16 weights of positions (2k, 2k+1) of a's storage order (v_pk_fma_f16
 * operand, prep.pk_f16_words of the [9][2C] table with the identity folded into the centre tap).
 * g1_blocked = 0: g1 natural NHWC [T][h][w][C] (sn_grp5_gemm_gate);
 * g1_blocked = 2 (C = 64 only): channel-planar [T][h][C][sn_planar_pitch(w)], zeros in the pad columns (sn_dw5m_gemm_gate).
 * pool: NULL or [T][sn_lngate_blocks][C]. */
int sn_ln_gemm_gate(const sn_unit_src* s, const void* hw, const void* wfrag, const float* bias, const uint32_t* wdw,
                    void* g1, float* pool, int g1_blocked, void* stream);
int sn_lngate_blocks(int h, int w);

/* "+" variants: RepConv with groups = C/8 (gshift_deblur1.py:157-165) as a block-diagonal MFMA GEMM, then body[4]
 * (1x1 C->2C), SimpleGate2 and the channel sums.  g1:[T][h][w][C] natural NHWC, wgrp: prep.pack_grouped_frag
 * [C/16][13][64][8] bf16 (3x3 and identity folded), C = 80.  pool: [T][sn_grp5_blocks][C]. */
int sn_grp5_gemm_gate(const void* g1, const float* ca_in, const void* wgrp, const void* wfrag, void* g2, float* pool,
                      int T, int h, int w, int C, void* stream);
int sn_grp5_blocks(int h, int w);

/* ---- matrix-core stencils (csrc/sn_gsts3.hip) ------------------------------------------------------------------
 * A depthwise k x k conv is a GEMM over the x axis: 16 outputs x 32 input columns of ONE channel per MFMA (banded
 * Toeplitz A operand), N = 16 places of that channel.  Its input is channel-planar: [T][h][C][wr] bf16,
 * wr = sn_planar_pitch(w) (w rounded up to 8), columns >= w are ZERO. */
int sn_planar_pitch(int w);
int sn_nhwc_to_planar(const void* x, void* xp, int T, int h, int w, int C, void* stream);   /* x:[T][h][w][C] -> xp planar */

/* g2 = SimpleGate2(body[4](RepConv(g1))) for the depthwise variants (gshift_deblur2.py:159-168,182-185,201): 5x5 + 3x3 +
 * identity folded into one 5x5 and run as Toeplitz MFMAs on the channel-planar g1p, the 1x1 C -> 2C on MFMA, x1 * sigmoid(x2),
 * per-workgroup channel sums for the CALayer2 that follows.  ca_in: NULL or [T][C] f32 scale applied first (denoise).  C = 64.
 * ttab: bf16 [C][5][2][20] padded bands (prep.pack_toeplitz), wfrag: body 1x1 fragments, gate-paired rows, natural K.
 * pool: [T][sn_dw5m_blocks(h,w)][C] partial sums of g2. */
int sn_dw5m_blocks(int h, int w);
int sn_dw5m_gemm_gate(const void* g1p, const float* ca_in, const void* ttab, const void* wfrag, void* g2, float* pool,
                      int T, int h, int w, int C, void* stream);

/* ---- the two phases of a GSTS unit's blocks (sn_gsts_cab2_phase1 / phase2, sn_cab1_phase1 / phase2) --------------------
 * The global average pool of CALayer2 (gshift_deblur1.py:76-87) is the one grid-wide dependency inside CAB2 / CAB1, so a block is two
 * passes over the frame: phase 1 up to g2 and its channel sums, [sn_ca_mlp on the sums], phase 2 from g2 to the block's output.
 *
 * PHASE 1, fused, every variant (csrc/sn_phase1r.hip):
 *   g2 = SimpleGate2(body[4](RepConv(SimpleGate(RepConv2(body[0](norm(u)))))))   (gshift_deblur1.py:183-211 CAB1, :212-255 CAB2; gshift_deblur2.py:186-258)
 * in ONE kernel: u is read once, g2 written once; `a`, g1 and r never reach HBM.  (The denoisers, whose inner CALayer2 needs the global pool of g1,
 * run it twice: sn_phase1_opts.  sn_ln_gemm_gate + sn_dw5m_gemm_gate / sn_grp5_gemm_gate above are the same phase 1 in two kernels with g1 in bf16 in HBM:
 * the engine's fallback for a checkpoint whose activations leave the fp16 range of `a`, g1 and r inside the fused kernel.)
 * sn_gsts_cab2_phase1: s->mode 1 / 2, hw = sn_gsts_shiftconv's output; sn_cab1_phase1: s->mode 0.  C = 64 or 80, RepConv depthwise or grouped 8 -> 8 ON THE
 * MATRIX CORES; weights prep.pack_phase1r --
 *     wfrag1  bf16 [C/8][KS1][64][8]: body[0] with the LayerNorm scale folded, wave-paired rows (M-tiles 2q, 2q+1 = channels 16q.. and their gate
 *             partners), the folded bias as bf16 hi + lo in the columns of k-slots K, K+1 (the kernel feeds the normalised input and a constant 1);
 *     w3      uint32 [C/16][4][9][4] packed-fp16 taps of RepConv2 (+identity) per (wave, lane group, tap, packed register);
 *     wgrp    fp16 [C/16][2][8][64][8]: RepConv (5x5 + 3x3 + identity) of every group as an x-pair Toeplitz GEMM (row = output channel + 8 * pixel of a
 *             pair; k = kernel row, input column 0..5 relative to the pair, input channel);
 *     wfrag2  fp16 [C/8][KS2][64][8]: body[4], wave-paired rows, sigmoid rows times -log2 e.
 * g2: [T][h][w][C] NHWC bf16.  pool: NULL or [T][sn_phase1_pool_blocks(T,h,w)][C] f32 partial channel sums of g2 (CALayer2, finished by sn_ca_mlp or by
 * the sn_se_fold tail): one row per (column strip, block of 8 image rows) -- a property of the image, not of the launch, so the sums and every
 * reduction over them are bit-identical whatever frame range (s->t0, s->nt), team size or device the launch runs with. */
typedef struct sn_phase1_weights {
    const void* wfrag1;
    const uint32_t* w3;
    const void* wgrp;
    const void* wfrag2;
} sn_phase1_weights;
/* Optional fold of CALayer2's squeeze-excite MLP (gshift_deblur1.py:76-87) into phase 1: the LAST workgroup of a frame to finish reduces
 * the frame's partial sums in a fixed order (bit-reproducible whichever workgroup that is) and writes ca[t][C] = sigmoid(wb relu(wa mean)),
 * so no sn_ca_mlp launch sits between the phases.  wa:[cr][c], wb:[c][cr] f32 (conv_du.0 / conv_du.2), c == C.
 * ticket: [>= T] u32 counters, ZERO before the first use; every launch leaves them zero again.  NULL: partial sums only. */
typedef struct sn_se_fold {
    const float* wa;
    const float* wb;
    int c, cr;
    unsigned* ticket;
    float* ca;
    unsigned* bad;       /* NULL, or the range guard of sn_ca_mlp: set to 1 when a channel sum of the frame is not finite */
} sn_se_fold;
/* The denoisers' inner CALayer2 on g1 = SimpleGate(...) (gshift_denoise1.py:224,257; gshift_denoise2.py:194,227).  Its global average
 * pool sits INSIDE phase 1, so phase 1 runs twice over the input and g1 still never reaches HBM:
 *   pass 1, g1_sums = 1: LayerNorm -> 1x1 -> dw3x3 -> gate only; pool receives the partial channel sums of g1 (finish them with sn_ca_mlp, or pass
 *           the inner CALayer2's weights as `se`: se->ca is then that layer's scale); g2 is not touched (may be NULL);
 *   pass 2, g1_scale = that scale [T][C] f32: the whole phase 1 with g1 multiplied by it before the RepConv.
 *   g1_store (optional, both passes): a scratch buffer of the size sn_phase1_g1_store_bytes gives for (T, h, w, C).  Pass 1 then also writes every g1 row it
 *           computes (fp16, unscaled, in the kernel's own strip / ring order: opaque), and pass 2 reads those rows back, scales them and runs only the
 *           RepConv -> 1x1 -> gate half: the stagers' LayerNorm and the first 1x1 + 3x3 run once per block instead of twice.  Bit-identical to the
 *           two passes without a store.  Both passes must see the same buffer, frame range and geometry.
 * NULL / zeros: the deblur models (no inner CALayer2).
 * team: 0 = the library chooses how many workgroups walk consecutive frames of the same rows in lock step (1, 2, 4 or 8: csrc/sn_phase1r.hip,
 * P1RPlan); a fixed value is for measurements only.  Results do not depend on it beyond the summation order of the pool rows. */
typedef struct sn_phase1_opts {
    const float* g1_scale;
    int g1_sums;
    int team;
    void* g1_store;
} sn_phase1_opts;
int sn_phase1_pool_blocks(int T, int h, int w);
int sn_phase1_g1_store_bytes(int T, int h, int w, int C, long long* bytes);
int sn_gsts_cab2_phase1(const sn_unit_src* s, const void* hw, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se,
                        const sn_phase1_opts* opt, void* stream);
int sn_cab1_phase1(const sn_unit_src* s, const sn_phase1_weights* wt, void* g2, float* pool, const sn_se_fold* se, const sn_phase1_opts* opt, void* stream);
/* The work decomposition of a phase-1 launch over nfr frames of h x w on a device with ncu compute units (host only, no device access): out7 =
 * {strips, slack / strip, slack remainder, team size F, frame blocks, rows per team chunk, teams}; sn_p1r_strip_begin: first own column of strip s
 * (s == strips: w).  Exposed for the host-logic tests and for tools that size measurements; the launch uses exactly this plan. */
int sn_p1r_plan(int nfr, int h, int w, int ncu, int team, int* out7);
int sn_p1r_strip_begin(const int* plan7, int s, int w);

/* PHASE 2, all variants (C = 64 / 80): y = shortcut + beta * body[7](ca * g2) (gshift_deblur1.py:201,210,254): beta and the optional bias
 * are folded into wfrag / bias; the shortcut is the ROLLED tensor for CAB2 (s->mode 1 / 2: sn_gsts_cab2_phase2) and x itself for CAB1
 * (s->mode 0: sn_cab1_phase2).  ca: [T][C] f32 from sn_ca_mlp. */
int sn_gsts_cab2_phase2(const sn_unit_src* s, const void* g2, const float* ca, const void* wfrag, const float* bias, void* y, void* stream);
int sn_cab1_phase2(const sn_unit_src* s, const void* g2, const float* ca, const void* wfrag, const float* bias, void* y, void* stream);


/* ---- fp32-storage path (csrc/sn_f32.hip) -----------------------------------------------------------------------
 * The arithmetic type upstream runs the "+" denoiser in (inference/test_denoise.py:83-85: the .half() is commented out)
 * and the validation build of the engine: activations fp32 NHWC [T][H][W][C] with an explicit pixel stride `cs`
 * (elements) so that channel slices of a wider tensor can be read / written in place, weights = the checkpoint's fp32
 * values re-ordered to [k][k][cin/groups][cout].  Direct fp32 FMA kernels (no MFMA): correctness first. */
typedef struct sn32_conv_desc {
    const float* in[3];  /* 1..3 inputs concatenated along channels (groups == 1), pre-offset to their first channel */
    int c_in[3];         /* logical channels taken from each input */
    int cs_in[3];        /* pixel stride of each input, elements */
    int n_in;
    int T, h_in, w_in;   /* spatial size the convolution sees */
    int in_mode;         /* 0 as is; 1 inputs are [T][h_in/2][w_in/2] upsampled x2 bilinearly while reading (gshift_deblur1.py:344) */
    int k, stride, pad, groups;   /* nn.Conv2d(.., groups): 1, C/8 (RepConv "+", :160-161) or C (depthwise) */
    int h_out, w_out, c_out;
    const float* w;      /* [k][k][cin_total/groups][c_out] */
    const float* bias;   /* [c_out] or NULL */
    int act; float prelu;                      /* as sn_conv_desc */
    const float* oscale; int oscale_stride;    /* NULL or [T][oscale_stride] (stride 0: one row for all frames): out = (conv+bias) * oscale (+ res) */
    const float* res; int cs_res;              /* NULL or NHWC tensor added last */
    void* out; int cs_out;
    int out_mode;        /* 0 NHWC fp32; 1 pixel_shuffle(2) NHWC fp32; 2 NCHW of nchw_dtype + shortcut sc (as sn_conv_desc) */
    int nchw_dtype; const void* sc;
    const void* wsplit;  /* NULL: exact fp32 products (v_mfma_f32_16x16x4_f32).  Else the weights as bf16 hi / lo A fragments
                          * (prep.pack_conv32_split): every product is wh xh + wh xl + wl xh on the bf16 matrix cores with fp32 accumulation
                          * (~2^-16 relative per product), taken for single-input stride-1 dense k = 1 / 3 and grouped-by-8 k = 3 / 5 convs */
    const float* iscale; int iscale_stride;   /* NULL or [T][iscale_stride] f32: the input is multiplied by iscale[t][ci] while it is loaded (the
                          * CALayer scale of the producer, gshift_deblur1.py:69-70, without a pass of its own); single-input dense / grouped-by-8 convs */
    const float* rscale; int rscale_stride;   /* NULL or [T][rscale_stride] f32: res is multiplied by rscale[t][co] before it is added (RepConv of the
                          * denoisers: conv(g1 ca1) + g1 ca1 with iscale = rscale = ca1, gshift_denoise1.py RepConv after CALayer2); grouped-by-8 convs */
    const float* ln_w; const float* ln_b;     /* NULL or [cin]: LayerNorm2d (gshift_deblur1.py:19-28, eps 1e-6) over the cin channels of every input pixel while
                          * it is loaded; the split-precision 1x1 path only (wsplit, k 1, cin <= 128, h_out w_out >= 64), SN_EINVAL otherwise */
    float* csum; int csum_cpad;               /* NULL or [T][sn32_conv_csum_tiles(h_out, w_out)][csum_cpad] f32: channel sums of the stored output per workgroup
                          * (AdaptiveAvgPool2d(1) of the CALayer behind a conv, finished by sn_ca_mlp / sn32_cab_ca); split-precision dense 3x3 only */
} sn32_conv_desc;
int sn32_conv_csum_tiles(int h_out, int w_out);
int sn32_conv2d(const sn32_conv_desc* d, void* stream);
/* channel_shift (gshift_deblur1.py:504-528) materialised in fp32: offs != NULL: u [T][h][w][3C/2] = cat(roll(x), shift(borrowed));
 * offs == NULL: the temporal roll alone, [T][h][w][C] (Shift_CAB, gshift_denoise1.py:167-179).  s->x is a float tensor.
 * u2: NULL, or (offs != NULL) a second [T][h][w][3C/2] tensor whose first C channels receive roll(x) too (CAB2's LayerNorm input is built in it). */
int sn32_gsts_gather(const sn_unit_src* s, const int8_t* offs, float* u, float* u2, void* stream);
/* channel_shift (gshift_deblur1.py:504-528) written where CAB2 consumes it: vin:[T][h][w][3C/2] receives roll(x) in its first C channels
 * (LayerNorm input and shortcut -- one copy, not the two of sn32_gsts_gather), and
 *   w == NULL, u != NULL: u:[T][h][w][C/2] = shift(borrowed half) for a separate conv1 (sn32_conv2d depthwise into vin[:, C:]);
 *   w != NULL, u == NULL: conv1 (depthwise 3x3, no bias, :206-212; w:[9][C/2] tap-major) is applied here, vin[:, C:] = conv1(shift(borrowed))
 *                         (bit-identical, slower: scattered 4-byte taps).
 * C % 8 == 0.  Frame range as sn32_gsts_gather. */
int sn32_gsts_shiftconv(const sn_unit_src* s, const int8_t* offs, const float* w, float* vin, float* u, void* stream);
/* LayerNorm2d (gshift_deblur1.py:19-28,44-53) over K channels per pixel. */
int sn32_layernorm(const float* x, int cs_x, int K, const float* w, const float* b, float* out, int cs_out, long long npix, void* stream);
/* SimpleGate (mode 0, :175-178) / SimpleGate2 (mode 1, :179-182): a:[npix][2C] -> out:[npix][C]. */
int sn32_gate(const float* a, int C, int mode, float* out, long long npix, void* stream);
/* SimpleGate / SimpleGate2 plus the channel sums of the result in one pass (the CALayer2 after it): out:[T][hw][C], partial as
 * sn32_chan_sum would compute from out. */
int sn32_gate_sum(const float* a, int C, int cpad, int mode, float* out, int T, int hw, int nblk, float* partial, void* stream);
/* The second 1x1 of phase 1 (C -> 2C, no bias) + SimpleGate2 (gshift_deblur1.py:179-182, x1 * sigmoid(x2)) + the channel sums of the result for
 * CALayer2, split-precision arithmetic (wsplit as in sn32_conv_desc): x:[T][hw][cs_x >= cin], out:[T][hw][C], partial:[T][hw / 64][cpad]
 * (finished by sn_ca_mlp with nblk = hw / 64).  hw % 64 == 0, C % 16 == 0, cin % 4 == 0, cin <= 128. */
int sn32_conv1x1_gate2(const float* x, int cs_x, int cin, const void* wsplit, int C, int cpad, float* out, int T, int hw, float* partial, void* stream);
/* RepConv2 (depthwise 3x3 + identity, gshift_deblur1.py:143-157) and SimpleGate (:175-178) in one pass: a:[T][h][w][cs_a >= 2C] f32,
 * w:[9][2C] (the depthwise weight, tap-major), out:[T][h][w][C] = a'[c] * a'[C + c]; partial NULL, or [T][nblk][cpad] channel sums of out
 * (what sn32_gate_sum returns) for the CALayer2 of the denoisers.  Bit-identical to sn32_conv2d(depthwise, res = a) + sn32_gate. */
int sn32_dw_gate(const float* a, int cs_a, const float* w, int C, int cpad, float* out, int T, int h, int wd, int nblk, float* partial, void* stream);
/* AdaptiveAvgPool2d(1) first half: partial:[T][nblk][cpad] sums (cpad >= C, <= 256), finished by sn_ca_mlp. */
int sn32_chan_sum(const float* x, int cs, int C, int cpad, int T, int hw, int nblk, float* partial, void* stream);
/* out = r * ca[t][c] (+ x if x != NULL). */
int sn32_scale_residual(const float* r, const float* x, const float* ca, int ca_stride, float* out, int T, int hw, int C, void* stream);
/* NCHW (src_dtype) [+ noise map] -> NHWC fp32 [T][H][W][C(+1)]. */
int sn32_ingest(const void* src, int src_dtype, const void* noise, float* dst, int T, int C, int H, int W, void* stream);


/* ---- I/O edges of the CLIs (csrc/sn_io.hip) ---------------------------------------------------------------------
 * numpy2tensor + .to(device) + .half() (inference/test_deblur.py:191-200,128,134) with the uint8 frames crossing PCIe:
 * src:[T][H][W][3] u8 (device) -> dst:[T][3][H][W] of dst_dtype, value = round(float(v) * (1/255)). */
int sn_ingest_u8(const uint8_t* src, void* dst, int dst_dtype, int T, int H, int W, void* stream);
/* clamp(0,1) * 255 of the network output (test_deblur.py:140-141): img:[T][H][W][3] u8 rounded to nearest even (what
 * cv2.imwrite stores, :152) or NULL; gt:[T][H][W][3] u8 or NULL; sse:[T][sn_egress_blocks()] f32 partial sums of the
 * squared error of the UNROUNDED value vs gt (skimage PSNR, data_range 255, :142). */
int sn_egress_blocks(void);
int sn_egress_u8(const void* out, int out_dtype, const uint8_t* gt, uint8_t* img, float* sse, int T, int H, int W, void* stream);
/* The CLIs' own SSIM (inference/test_deblur.py:25-49: Gaussian statistics, sd 1.5, over the (C,H,W) volume of clamp(out,0,1) and
 * gt / 255, scipy 'reflect' boundaries on all three axes): scratch:[T][15][H][W] f32 workspace, partial:[T][sn_ssim_blocks()]
 * f32 sums of the SSIM map; SSIM of frame t = sum(partial[t]) / (3 H W). */
int sn_ssim_blocks(void);
int sn_ssim_u8(const void* out, int out_dtype, const uint8_t* gt, float* scratch, float* partial, int T, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
