/*
 * speinet_hip.h — C-ABI of the MI355X (gfx950) kernels behind SPEINet's per-sequence forward pass.
 *
 * The reference (yangt1013/SPEINet) is pure PyTorch: it has no FFI for this path, so this header *defines*
 * the boundary underneath the drop-in Python class (last row).  Each entry point names the
 * reference code it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; the library never
 *     allocates, frees or synchronises), every call is asynchronous on `stream`;
 *   - feature maps are NHWC fp32 ("pixel rows": [H][W][C], C contiguous) with an explicit row stride `ld*`
 *     in floats so channel-concatenated buffers are addressed in place; frames are NCHW fp32 planes;
 *   - packed weights are [tap][Cout][Cin] fp32 (speinet_amd/pack.py builds them once per checkpoint);
 *   - return 0 on success, <0 on error: -1 bad argument / unsupported shape, -2 launch failure; the text
 *     is available from spei_last_error() (thread local).
 */
#ifndef SPEINET_HIP_H
#define SPEINET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spei_stream_t; /* hipStream_t */

#define SPEI_ACT_NONE 0
#define SPEI_ACT_RELU 1
#define SPEI_ACT_GELU 2 /* exact erf GELU (nn.GELU default, reference model/swinir.py:14-19) */

/* Element formats.  16-bit entry points take `fmt` = SPEI_BF16 or SPEI_F16: the operand type of the matrix pipe
 * (v_mfma_f32_32x32x16_bf16 / _f16, same rate, fp32 accumulation; half keeps 11 significand bits instead of 8 at a
 * range of +-65504).  `*_fmt` arguments say how a tensor is stored in HBM: SPEI_F32 or the call's 16-bit format. */
#define SPEI_F32 0
#define SPEI_BF16 1
#define SPEI_F16 2

#define SPEI_CONV 0
#define SPEI_CONV_TRANSPOSED 1 /* ConvTranspose2d(k, stride, pad=k/2, output_padding=stride-1) */

int spei_version(void);
const char* spei_last_error(void);
const char* spei_arch(void); /* "gfx950" */

/* K15 — routing test `torch.all(x[:,3]==0)` (model/speinet.py:70-73).  flag[0] := 1 if any element != 0. */
int spei_any_nonzero(const float* x, int64_t n, int32_t* flag, spei_stream_t stream);

/* K1 — r_l_per_channel(img, box5/25, iters, lam) (model/rcl.py:22-51).  img/out: [C][H][W]; scratch same size. */
int spei_rl_prior(const float* img, float* out, float* scratch, int C, int H, int W, int iters, float lam,
                  spei_stream_t stream);

/* K2 head — first conv 5x5 pad 2, Cin=3 NCHW planes -> NHWC [H][W][Cout], +bias, ReLU
 * (model/recons_video_ori.py:28-32).  w: [25][Cout][3] packed, Cout == 32. */
int spei_conv5_in(const float* img_chw, const float* w, const float* bias, float* out_hwc, int H, int W,
                  int Cout, spei_stream_t stream);

/* K2 tail — last conv 5x5 pad 2, NHWC [H][W][32] -> NCHW [3][H][W], +bias, no activation
 * (model/recons_video_ori.py:75-77).  w: [25][3][Cin] packed. */
int spei_conv5_out(const float* in_hwc, int ldi, const float* w, const float* bias, float* out_chw, int H,
                   int W, int Cin, spei_stream_t stream);

/* K2/K4/K5/K6/K9 — implicit-GEMM convolution / linear on f32 MFMA:
 *   out[m][n] = epi( sum_t sum_k A[src(m,t)][k] * w[t][n][k] ),  m = oy*Wout+ox, k over cat(a0[:, :k0], a1[:, :k1])
 *   epi(v) = (act(v + bias[n])) * rowscale[m] + residual[m][n]      (rowscale / residual optional)
 * Replaces nn.Conv2d / nn.ConvTranspose2d / nn.Linear call sites: model/block.py:26-47, model/swinir.py:18-29,
 * 105-108,467,667,716,742, model/speinet.py:55-66,93-119, model/recons_video_ori.py:44-71.
 * Constraints: k0, k1 multiples of 32; N multiple of 32; lda*, ldo, ldr multiples of 4. */
int spei_igemm_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                   const float* bias, float* out, int ldo, const float* residual, int ldr,
                   const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize, int stride,
                   int pad, int mode, int act, spei_stream_t stream);

/* The same on `batch` equally sized maps stored one after the other (a0 / a1 / out / residual rows and rowscale entries of sample b
 * start at b * map size): one launch, blockIdx.z = sample (the training step's batches of 200x200 crops give single maps of only
 * 2 500 .. 40 000 pixels — a launch per sample leaves most of the chip idle). */
int spei_igemm_f32_batched(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                           const float* bias, float* out, int ldo, const float* residual, int ldr,
                           const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize, int stride,
                           int pad, int mode, int act, int batch, spei_stream_t stream);

/* Same contract on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 activations in HBM rounded to
 * bf16 while staged to LDS).  w_hi/w_lo: [tap][N][K] bf16.  w_lo == NULL: one bf16 product per MAC ("bf16");
 * w_lo != NULL: split product al*wh + ah*wl + ah*wh ("bf16x3", f32-grade at 3/16 of the f32 MFMA cost). */
int spei_igemm_bf16(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi,
                    const void* w_lo, const float* bias, float* out, int ldo, const float* residual, int ldr,
                    const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize, int stride,
                    int pad, int mode, int act, spei_stream_t stream);

/* Slab-resident variant of spei_igemm_bf16 for SPEI_CONV (stride 1/2) and linears, bf16 or half operands (`fmt`): the
 * input tile + halo is staged once into LDS as 16-bit and the weights stream from HBM/L2 in MFMA fragment order
 * (wfrag: [N/32][tap][K/16][64][8] 16-bit, see speinet_amd/pack.py).  Linears: pass Hin = Hout = M, Win = Wout = 1.
 * a_fmt / out_fmt: the activations (both sources) / the output are stored as fp32 or as `fmt` in HBM — 16-bit for
 * tensors that only feed the next GEMM or the attention kernel; residual, rowscale and bias stay fp32.
 * wfrag_lo != NULL: the split "bf16x3" product (SPEI_BF16 with fp32 activations only).
 * ln_input: LayerNorm(256) without affine (model/swinir.py:244, affine folded into the weights) is applied to each
 * 256-wide fp32 input row while it is staged, so the normalised tokens never exist in HBM. */
int spei_conv_slab16(int fmt, const void* a0, int lda0, int k0, const void* a1, int lda1, int k1, int a_fmt,
                     const void* wfrag_hi, const void* wfrag_lo, const float* bias, void* out, int ldo, int out_fmt,
                     const float* residual, int ldr, const float* rowscale, int Hin, int Win, int Hout, int Wout,
                     int N, int ksize, int stride, int pad, int act, int ln_input, spei_stream_t stream);

/* The same convolution on `batch` dense maps in ONE launch (gridDim.y = map): the frame's 7 encoder passes go through every layer of
 * model/recons_video_ori.py:26-56 with the same weights (model/speinet.py:82-83,125-131).  a0 [batch][Hin*Win][k0], out
 * [batch][Hout*Wout][N], fp32 or `fmt` each; wfrag_lo != NULL: the split (bf16x3) form on fp32 maps (the training step's forward and
 * data-gradient convolutions, speinet_amd/train.py); residual: NULL or [batch][Hout*Wout][N] fp32, added in the epilogue.  Per map the tiles and the arithmetic are those of
 * spei_conv_slab16 (bit-identical results); a launch has batch x the workgroups (several resident rounds instead of half of one at
 * H/4) and a frame needs 1/batch of the launches. */
int spei_conv_slab16_batched(int fmt, const void* a0, int k0, int a_fmt, const void* wfrag, const void* wfrag_lo, const float* bias,
                             void* out, int out_fmt, const float* residual, int batch, int Hin, int Win, int Hout, int Wout, int N,
                             int ksize, int stride, int pad, int act, spei_stream_t stream);

/* The same convolution (stride 1, pad k/2, one dense fp32 input map, single-product arithmetic) with the PREVIOUS ResBlock's gated
 * residual sum folded into its staging (model/block.py:136-140 feeding the next block's first conv, :127-131):
 *     x'[p][c] = x[p][c] + x1[p][c] * (s[c] + g1[y][c] + g2[x][c]),    out = act(conv(x', w) + bias)
 * x' is what spei_resblock_apply would have written; every pixel of it is also stored to x_out (fp32 [H*W][K], must not alias x)
 * by the workgroup that owns it: the residual stream of the next block.  x1: 16-bit (fmt) [H*W][K]; s [K], g1 [H][K], g2 [W][K]
 * from spei_resblock_gates.  K in {32, 64, 128, 256}. */
int spei_conv_slab16_fa(int fmt, const float* x, int K, const void* x1, const float* s, const float* g1, const float* g2,
                        float* x_out, const void* wfrag, const float* bias, void* out, int ldo, int out_fmt, int H, int W, int N,
                        int ksize, int act, spei_stream_t stream);

/* 32 -> 32 channel 5x5 convolution (stride 1, zero padding 2), weight-stationary (round 4, csrc/conv32_ws16.hip): the two convs of a
 * ResBlock at full resolution (model/block.py:26-47,127-131 in model/recons_video_ori.py:26-43 inBlock and :72-75 outBlock) — every wave
 * keeps the layer's 50 weight fragments in registers for the whole launch, one persistent workgroup per CU walks over 16 x 32 pixel
 * tiles of all `batch` maps and double-buffers the tiles' slabs in LDS.  a: [batch][H*W][32] fp32 or `fmt`; wfrag: the layer's weights
 * in fragment order (as spei_conv_slab16 takes them); bias [32] or NULL; out: [batch][H*W][32] `fmt` or fp32, must not be `a`;
 * act: SPEI_ACT_NONE / SPEI_ACT_RELU.  Same operand rounding as spei_conv_slab16, fp32 sums in (tap, channel) order. */
int spei_conv32_ws16(int fmt, const void* a, int a_fmt, const void* wfrag, const float* bias, void* out, int out_fmt, int batch,
                     int H, int W, int act, spei_stream_t stream);

/* Fused Swin MLP branch (model/swinir.py:12-29 Mlp.forward + the `x + mlp(norm2(x))` tail of :279), 16-bit matrix pipe:
 * out = x + fc2(GELU(fc1(LayerNorm256(x)))), LayerNorm affine folded into w1/b1 (pack.py); w*_frag in MFMA fragment
 * order; the normalised tokens and the 512-wide hidden activations live only in LDS.  x, out: [M][256] fp32, may alias. */
int spei_mlp_fused16(int fmt, const float* x, float* out, const void* w1_frag, const float* b1, const void* w2_frag,
                     const float* b2, int64_t M, spei_stream_t stream);

/* ConvTranspose2d(k = 3, stride 2, padding 1, output_padding 1) on the slab kernel (reference model/recons_video_ori.py:58-71,
 * the tails of decoder_second / decoder_first): the four output-parity classes are stride-1 convolutions over the input
 * grid with 1 / 2 / 2 / 4 taps, written with pixel stride 2.  wfrag<py><px>: fragment-ordered 16-bit weights of class
 * (oy % 2, ox % 2) (speinet_amd/pack.py).  a0 [Hin*Win][lda0] fp32 or `fmt`, out [2Hin*2Win][ldo] fp32 or `fmt`. */
int spei_convt2_slab16(int fmt, const void* a0, int lda0, int k0, int a_fmt, const void* wfrag00, const void* wfrag01,
                       const void* wfrag10, const void* wfrag11, const float* bias, void* out, int ldo, int out_fmt,
                       int Hin, int Win, int N, int act, spei_stream_t stream);

/* The same transposed conv in split (bf16x3, f32-grade) arithmetic: fp32 in, fp32 out; whi4 / wlo4: the four class weights (order (0,0),
 * (0,1), (1,0), (1,1)) as bf16 high halves and low halves (= bf16(w - hi)) in fragment order.  Replaces the round-1 igemm path for the
 * ConvTranspose2d that ends decoder_second inside an f16 frame's split stages (model/recons_video_ori.py:59-70, model/speinet.py:99). */
int spei_convt2_slab16x3(const float* a0, int lda0, int k0, const void* const* whi4, const void* const* wlo4, const float* bias,
                         float* out, int ldo, int Hin, int Win, int N, int act, spei_stream_t stream);

/* 3x3 convolution 256 -> 256 channels (stride 1, padding 1) on a batch of fp32 token maps: the `conv(blocks(x)) + x` tail of every
 * residual Swin group and conv_after_body (reference model/swinir.py:467,483-484,742) as a persistent, software-pipelined kernel — one
 * workgroup per CU walks 6 x 16 pixel tiles, every weight fragment feeds three MFMAs, the next tile's halo is staged inside the
 * current tile's 144 k-steps (round 4; conv_slab_kernel took 190 us per launch for the frame's two maps).  x, out, residual (or NULL;
 * may alias out): [batch][H*W][256] fp32, x != out; w_frag: fragment-ordered 16-bit weights [8][9][16][64][8] (pack.py); bias [256] or
 * NULL.  H % 6 == 0, W % 16 == 0, H*W <= 2^21. */
int spei_conv3x3_256_pipe16(int fmt, const float* x, const void* w_frag, const float* bias, const float* residual, float* out,
                            int batch, int H, int W, spei_stream_t stream);

/* Last conv (model/recons_video_ori.py:75-77: 5x5, 32 -> 3 channels, NHWC in, three NCHW fp32 planes out) on the
 * slab kernel: wfrag = fragment-ordered weights zero-padded to 32 output channels, bias32 = bias padded to 32. */
int spei_conv5_out_slab16(int fmt, const void* in, int ldi, int in_fmt, const void* wfrag, const float* bias32, float* out_chw,
                          int H, int W, spei_stream_t stream);

/* Fused attention branch of a Swin block (model/swinir.py:238-278 + :115-149): out = x + proj(W-MSA(q = yhat Wq,
 * [k,v] = LayerNorm(x) Wkv)) with cyclic shift `shift`, 5x5 windows, 8 heads; x,out [H*W][256] fp32 (may alias), yhat
 * [H*W][256] `fmt` (LayerNorm of y without affine); w*_frag in MFMA fragment order with the LayerNorm affine and the q
 * scale folded in (pack.py); relbias [8][25][25].  q, k, v, the attention matrix and its output never reach HBM. */
int spei_attn_fused16(int fmt, const float* x, float* out, const void* yhat, const void* wq_frag, const float* bq,
                      const void* wkv_frag, const float* bkv, const void* wproj_frag, const float* bproj,
                      const float* relbias, int H, int W, int shift, spei_stream_t stream);

/* The same attention branch (model/swinir.py:238-278, :115-149) over a BATCH of equally sized maps that share the block's weights
 * — the two Swin calls of a frame, model/speinet.py:84 — in one launch, four windows per 256-thread workgroup (round 4): every
 * weight fragment fetched from L2 feeds four MFMAs instead of two, one 100-row token slab per workgroup (y-hat, then LayerNorm(x),
 * then the attention output) so that two workgroups still share a CU.  x, out: [batch][H*W][256] fp32 (may alias), yhat
 * [batch][H*W][256] `fmt`; weights, biases, relbias as spei_attn_fused16.  Per map the result differs from spei_attn_fused16 in
 * fp32 rounding only (the residual enters the projection sum first instead of last). */
int spei_attn_win4_16(int fmt, const float* x, float* out, const void* yhat, const void* wq_frag, const float* bq,
                      const void* wkv_frag, const float* bkv, const void* wproj_frag, const float* bproj,
                      const float* relbias, int batch, int H, int W, int shift, spei_stream_t stream);

/* Harness post-processing of one deblurred frame (inference_SPEINet.py:477-482 tensor2numpy, :484-500 calc_PSNR, :502-543 calc_SSIM):
 * out_chw [3][H][W] fp32 (the model's output, unclamped) -> out_hwc [H][W][3] uint8 = round(clamp(255 x, 0, 255)) (half to even), and
 * result[0] = 1 if every value of out_chw was finite else 0, result[1] = PSNR, result[2] = SSIM of out_hwc against gt_hwc ([H][W][3]
 * uint8) on the region cropped by `border` pixels on every side (the reference crops 4).  PSNR from the exact integer squared error
 * (inf for identical frames); SSIM: 11x11 Gaussian window (sigma 1.5), valid region, float64 sums, mean over channels and positions.
 * ws: spei_frame_post_ws_doubles(H, W, border) doubles (-1: the cropped frame is smaller than the window). */
int64_t spei_frame_post_ws_doubles(int H, int W, int border);
int spei_frame_post(const float* out_chw, const unsigned char* gt_hwc, unsigned char* out_hwc, int H, int W, int border,
                    double* ws, double* result, spei_stream_t stream);

/* K3 — ResBlock gates (model/block.py:8-24 SE, 71-96 ZPool+AttentionGate1/2, 108-124 TripletAttention).
 * x1: conv2 output [H][W][C] stored as x1_fmt (SPEI_F32 / SPEI_BF16 / SPEI_F16).  Workspace `ws` floats: spei_gate_ws_floats(H,W,C).
 * Produces s[C], g1[H][C], g2[W][C] such that ResBlock = x + x1*s + (x1*g1 + x1*g2).
 * gate params (packed by speinet_amd/pack.py): se_w1[C/4][C], se_b1[C/4], se_w2[C][C/4], se_b2[C],
 * cw_w[2][7][7], cw_bn[2] = {scale, shift}, hc_w[2][5][5], hc_bn[2]. */
int64_t spei_gate_ws_floats(int H, int W, int C);
int spei_resblock_gates(const void* x1, int x1_fmt, int H, int W, int C, const float* se_w1, const float* se_b1,
                        const float* se_w2, const float* se_b2, const float* cw_w, const float* cw_bn,
                        const float* hc_w, const float* hc_bn, float* s, float* g1, float* g2, float* ws,
                        spei_stream_t stream);
/* out = x + x1*s + (x1*g1 + x1*g2) [+ extra]   (model/block.py:136-140; `extra` fuses speinet.py:84,132) */
int spei_resblock_apply(const float* x, const void* x1, int x1_fmt, const float* s, const float* g1, const float* g2,
                        const float* extra, float* out, int ldo, int H, int W, int C, spei_stream_t stream);

/* spei_resblock_gates / spei_resblock_apply on `batch` dense maps in one launch each (gridDim.y = map): x1 [batch][H*W][C];
 * s [batch][C], g1 [batch][H][C], g2 [batch][W][C]; ws: batch x spei_gate_ws_floats(H, W, C) floats; x, out [batch][H*W][C] fp32.
 * Per map the arithmetic (and every partial-sum order) is that of the single-map entry points. */
int spei_resblock_gates_batched(const void* x1, int x1_fmt, int batch, int H, int W, int C, const float* se_w1, const float* se_b1,
                                const float* se_w2, const float* se_b2, const float* cw_w, const float* cw_bn, const float* hc_w,
                                const float* hc_bn, float* s, float* g1, float* g2, float* ws, spei_stream_t stream);
int spei_resblock_apply_batched(const float* x, const void* x1, int x1_fmt, const float* s, const float* g1, const float* g2, float* out,
                                int batch, int H, int W, int C, spei_stream_t stream);

/* K7 — LayerNorm over C=256, eps 1e-5 (model/swinir.py:244-245,279,528-529,776).  gamma/beta may be NULL
 * (affine folded into the following linear by pack.py); out_fmt: y is fp32, or 16-bit when it only feeds a GEMM. */
int spei_layernorm256(const float* x, void* y, int out_fmt, const float* gamma, const float* beta, int64_t M,
                      spei_stream_t stream);

/* K8 — window attention core: cyclic shift, 5x5 partition, softmax(q k^T + relbias + shift mask) v, reverse
 * (model/swinir.py:115-149, 215-236, 250-275).  q [H*W][256] (scale folded), kv [H*W][512] (K then V, head major),
 * relbias [8][25][25] pre-gathered, out [H*W][256]; heads = 8, head_dim = 32, window 5.  io_fmt: q, kv and out
 * are fp32 or 16-bit in HBM (the arithmetic stays fp32 on the f32 MFMA). */
int spei_window_attention(const void* q, const void* kv, int io_fmt, const float* relbias, void* out, int H, int W,
                          int shift, spei_stream_t stream);
int spei_window_attention_batched(const void* q, const void* kv, int io_fmt, const float* relbias, void* out, int H, int W,
                                  int shift, int batch, spei_stream_t stream);   /* batch maps stored one after the other */

/* K10 — 1 / max(||unfold3x3(f)[p]||_2, 1e-12) per position (F.normalize, model/SearchTransfer.py:30-31). */
int spei_patch_invnorm(const float* f, int ldf, float* inv, int H, int W, int C, spei_stream_t stream);

/* K11 — fused correlation + max/argmax over the reference index j (model/SearchTransfer.py:33-34, 68-69):
 *   S[i] = max_j <P_ref[j], P_lr[i]> * inv_ref[j] * inv_lr[i],  arg[i] = lowest maximising j.
 * R is never materialised.  Workspace floats: spei_corr_ws_floats(Hl*Wl).  C must be 128. */
int64_t spei_corr_ws_floats(int64_t n_lr);
int spei_corr_argmax(const float* lr, int ldl, const float* ref, int ldr, const float* inv_lr,
                     const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg,
                     float* ws, spei_stream_t stream);

/* K11 on the 16-bit pipe.  spei_split16 converts a map once: hi = fmt(x), lo = fmt(x - hi) (lo may be NULL);
 * outputs are dense [M][C] 16-bit.  spei_corr_argmax_bf16 (tile-restaging form, bf16 only): lo pointers NULL -> single
 * bf16 products, else bf16x3. */
int spei_split16(int fmt, const float* x, int ld, void* hi, void* lo, int64_t M, int C, spei_stream_t stream);
int spei_corr_argmax_bf16(const void* lr_hi, const void* lr_lo, const void* ref_hi, const void* ref_lo,
                          const float* inv_lr, const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C, float* S,
                          int32_t* arg, float* ws, spei_stream_t stream);

/* Slab-resident form (C == 128): the query block and the streamed reference blocks live in LDS with their 1-pixel
 * halo, so the 3x3 unfold is LDS addressing, not memory traffic.  lo pointers: NULL, or (SPEI_BF16 only) the bf16x3 split. */
int spei_corr_slab16(int fmt, const void* lr_hi, const void* lr_lo, const void* ref_hi, const void* ref_lo,
                     const float* inv_lr, const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C, float* S,
                     int32_t* arg, float* ws, spei_stream_t stream);

/* Exact arg-max at 16-bit cost (ops.Ctx corr_precision "top2").  spei_corr_slab_top2_16: the slab kernel with single
 * 16-bit products keeping the TWO best candidates of every query: arg / arg2 (arg2 = -1: no second candidate) with their
 * approximate, un-normalised-by-inv_lr scores S / S2.  spei_corr_rescore then re-scores both candidates on the fp32 maps
 * with fp64 accumulation and overwrites S (= dot * inv_ref[j] * inv_lr[i], fp32) and arg (ties -> lowest index): the
 * winner no longer depends on 16-bit rounding (model/SearchTransfer.py:33-34 computes R in fp32). */
int spei_corr_slab_top2_16(int fmt, const void* lr16, const void* ref16, const float* inv_lr, const float* inv_ref,
                           int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg, float* S2, int32_t* arg2,
                           float* ws, spei_stream_t stream);
/* Diagonal-sliding form of the candidate pass (csrc/corr_diag16.hip) for a reference map at least as high as the query
 * map (Hr >= Hl: SearchTransfer's maps of one size, SelfTransfer's rotated landscape map): the score of (y, x) against
 * (y', x') is the sum of three row-against-row terms shared along the diagonal y' - y, so each term is computed once — a
 * third of the flops of model/SearchTransfer.py:33's bmm reach the matrix pipe, every score is still the full 9 C-term
 * fp32 sum.  Same outputs as spei_corr_slab_top2_16 (feed spei_corr_rescore); workspace floats:
 * spei_corr_diag_ws_floats(Hl, Wl, Hr, Wr). */
int64_t spei_corr_diag_ws_floats(int Hl, int Wl, int Hr, int Wr);
int spei_corr_diag_top2_16(int fmt, const void* lr16, const void* ref16, const float* inv_ref, int Hl, int Wl, int Hr, int Wr,
                           int C, float* S, int32_t* arg, float* S2, int32_t* arg2, float* ws, spei_stream_t stream);
int spei_corr_rescore(const float* lr, int ldl, const float* ref, int ldr, const float* inv_lr, const float* inv_ref,
                      int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg, const float* S2, const int32_t* arg2,
                      spei_stream_t stream);

/* K12 — gather the best-matching reference patch and overlap-add (unfold -> bis -> fold / 9,
 * model/SearchTransfer.py:36-46).  scale s in {1,2,4}: patch 3s, stride s, pad s. */
int spei_gather_fold(const float* ref, int ldr, const int32_t* arg, float* out, int ldo, int H3, int W3,
                     int Hr3, int Wr3, int C, int s, spei_stream_t stream);

/* SelfTransfer's reference map x.transpose(2,3).flip(2) (model/SearchTransfer.py:60): [H][W][C] -> [W][H][C]. */
int spei_rot90(const float* in, int ldi, float* out, int H, int W, int C, spei_stream_t stream);

/* K13 — F.interpolate(mode='bicubic', align_corners=False), A=-0.75, scale s in {2,4}
 * (model/speinet.py:96,99,108,111,113; model/SearchTransfer.py:73,75).  act: SPEI_ACT_NONE, or SPEI_ACT_RELU applied to
 * the interpolated value — relu(conv1x1(up(x))) == relu(up(conv1x1(x) + b)) (both maps are linear, the bicubic weights
 * sum to 1), which the throughput mode uses to run those 1x1 convs at a quarter of the pixels. */
int spei_upsample_bicubic(const float* in, int ldi, float* out, int ldo, int H, int W, int C, int s, int act,
                          spei_stream_t stream);

/* K14 — out = a + b over n floats. */
int spei_add(const float* a, const float* b, float* out, int64_t n, spei_stream_t stream);

/* ---- backward kernels: first slice of the training step (the reference obtains these from torch.autograd
 * inside trainer/trainer_swint_hsa_nsf.py:34-40 `loss.backward()`).  fp32, fixed-order reductions (bitwise reproducible).
 * The DATA gradient of a convolution is spei_igemm_f32(mode = SPEI_CONV_TRANSPOSED) on dY with w[t][k][n] = W[t][n][k]. ---- */

/* Weight (and bias) gradient of Conv2d(K -> N, k, stride, pad = k/2) (model/block.py:26-47, recons_video_ori.py:28-71):
 * dw[t][n][k] = sum_m dy[m][n] * x[src(m,t)][k], dbias[n] = sum_m dy[m][n] (dbias may be NULL).  x [Hin*Win][ldx] NHWC,
 * dy [Hout*Wout][ldy]; ws: spei_wgrad_ws_floats(...) floats.  N <= 256; K, N need not be multiples of 32. */
int64_t spei_wgrad_ws_floats(int Hout, int Wout, int N, int K, int ksize);
int spei_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin, int Win,
                        int Hout, int Wout, int N, int K, int ksize, int stride, int pad, spei_stream_t stream);

/* The same summed over `batch` equally sized maps stored one after the other (x: batch * Hin*Win rows, dy: batch * Hout*Wout rows):
 * one launch, one fixed-order reduction over all samples' pixels. */
int spei_conv_wgrad_f32_batched(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin, int Win,
                                int Hout, int Wout, int N, int K, int ksize, int stride, int pad, int batch, spei_stream_t stream);

/* Training: a weight in the reference's layout (Conv2d [N][K][ks][ks] / Linear [N][K], fp32) -> the split pair
 * hi = bf16(w), lo = bf16(w - hi) in the slab kernels' fragment order, one launch.  mode 0: the forward GEMM weight
 * [tap][N][K]; mode 1: the stride-1 data-gradient weight (taps reversed, channel axes swapped: [tap][K][N]).  The
 * weights of a training step change every optimizer step (trainer/trainer_swint.py:39-44). */
int spei_pack_split16(const float* w, int N, int K, int ksize, int mode, void* frag_hi, void* frag_lo, spei_stream_t stream);

/* The same weight / bias gradient with the products split on the 16-bit matrix pipe (a = ah + al in bf16: al*bh + ah*bl + ah*bh,
 * fp32 accumulation: 2^-16 relative per product; speinet_amd/train.py `train_precision = "bf16x3"`): three v_mfma_f32_32x32x16_bf16 per
 * 16 pixels where the fp32 kernel issues eight v_mfma_f32_32x32x2_f32.  Arguments, workspace and summation order as
 * spei_conv_wgrad_f32_batched. */
int spei_conv_wgrad_bf16x3_batched(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin, int Win,
                                   int Hout, int Wout, int N, int K, int ksize, int stride, int pad, int batch, spei_stream_t stream);

/* ReLU backward on the output of a fused conv + ReLU: dz = dy where y > 0, else 0 (n floats, n % 4 == 0). */
int spei_relu_bwd(const float* y, const float* dy, float* dz, int64_t n, spei_stream_t stream);

/* Plane statistics of the ResBlock gates, training form (model/block.py:71-73 ZPool over W and over H, :8-24 SE pooling):
 * prod == 0: rowmax, rowmean [H][C] (over x), colmax, colmean [W][C] (over y), mean [C] of a [H][W][C];
 * prod == 1: the plain sums of a * b over x, over y and over the map into rowmean / colmean / mean (rowmax = colmax = NULL):
 * with a = dOut and b = x1 these are the gradients of g1, g2 and s of out = x + x1 * (s + g1 + g2).
 * ws: spei_plane_ws_floats(H, W, C) floats.  C in {32, 64, 128}. */
int64_t spei_plane_ws_floats(int H, int W, int C);
int spei_plane_stats(const float* a, const float* b, int prod, int H, int W, int C, float* rowmax, float* rowmean, float* colmax,
                     float* colmean, float* mean, float* ws, spei_stream_t stream);

/* The same for `batch` equally sized maps stored one after the other (a, b: batch * H*W rows; outputs [batch][H][C], [batch][W][C],
 * [batch][C]); ws: batch * spei_plane_ws_floats(H, W, C) floats. */
int spei_plane_stats_batched(const float* a, const float* b, int prod, int H, int W, int C, float* rowmax, float* rowmean, float* colmax,
                             float* colmean, float* mean, float* ws, int batch, spei_stream_t stream);

/* Backward of the gated residual sum through x1 (model/block.py:136-140): dx1 = dOut * (s + g1 + g2) + the pooled statistics'
 * gradients routed back (means spread evenly, maxima to the arg-max element).  dx = dOut needs no kernel. */
int spei_resblock_apply_bwd(const float* dout, const float* x1, const float* s, const float* g1, const float* g2, const float* rowmax,
                            const float* colmax, const float* d_rowmax, const float* d_rowmean, const float* d_colmax,
                            const float* d_colmean, const float* d_mean, float* dx1, int H, int W, int C, spei_stream_t stream);
/* The same for `batch` dense maps stored one after the other (and their statistics / gate maps likewise) in one launch. */
int spei_resblock_apply_bwd_batched(const float* dout, const float* x1, const float* s, const float* g1, const float* g2,
                                    const float* rowmax, const float* colmax, const float* d_rowmax, const float* d_rowmean,
                                    const float* d_colmax, const float* d_colmean, const float* d_mean, float* dx1, int batch, int H,
                                    int W, int C, spei_stream_t stream);

/* The gate maps of a ResBlock from its plane statistics, forward and backward, for the training step (model/block.py:8-24 SEBlock,
 * :49-68 BasicConv1 = 2->1 conv + BatchNorm2d(1), :75-96 AttentionGate1/2, :116-119 TripletAttention) — BatchNorm on batch statistics
 * with its running buffers moved (momentum 0.01, unbiased variance) when bn_train, on the running statistics otherwise:
 *   s [N][C] = sigmoid(W2 relu(W1 mean + b1) + b2);  g1 [N][H][C] = BN(conv7x7([rowmax, rowmean]));  g2 [N][W][C] = BN(conv5x5([colmax^T,
 *   colmean^T]))^T.   N = groups x B samples, BatchNorm statistics per group of B consecutive samples, running buffers moved once per
 * group in group order.  prm: HOST array of 10 device pointers se_w1 [C/4][C], se_b1, se_w2 [C][C/4], se_b2, cw_w [2][7][7], cw_g, cw_b,
 * hc_w [2][5][5], hc_g, hc_b; run: HOST array of 4 device pointers cw_rm, cw_rv, hc_rm, hc_rv (scalars).  saved
 * (spei_gate_train_saved_floats): what the backward needs (conv outputs, SE hidden layer, the statistics used); ws
 * (spei_gate_train_ws_floats floats, 8-byte aligned).  The backward returns the gradients of the five statistics and of the ten
 * parameters packed in prm order (spei_gate_train_nparams floats).  float64 sums in a fixed order: bitwise reproducible. */
int64_t spei_gate_train_saved_floats(int N, int groups, int H, int W, int C);
int64_t spei_gate_train_ws_floats(int N, int groups, int H, int W, int C);
int spei_gate_train_nparams(int C);
int spei_gate_maps_fwd(const float* rowmax, const float* rowmean, const float* colmax, const float* colmean, const float* mean,
                       const float* const* prm, float* const* run, int N, int groups, int H, int W, int C, int bn_train,
                       int update_running, float* s, float* g1, float* g2, float* saved, float* ws, spei_stream_t stream);
int spei_gate_maps_bwd(const float* rowmax, const float* rowmean, const float* colmax, const float* colmean, const float* mean,
                       const float* const* prm, float* const* run, int N, int groups, int H, int W, int C, int bn_train,
                       const float* s, const float* saved, const float* ds, const float* dg1, const float* dg2, float* d_rowmax,
                       float* d_rowmean, float* d_colmax, float* d_colmean, float* d_mean, float* dprm, float* ws,
                       spei_stream_t stream);

/* ---- backward of the cross-window-attention SwinIR blocks (model/swinir.py:238-281 under loss.backward(), the training step of
 * trainer/trainer_swint.py:34-44).  fp32; fixed-order reductions. ---- */

/* nn.LayerNorm(256) backward: dx [M][256] from x (the layer's input; mean / rstd are recomputed), gamma (NULL = no affine) and dy.
 * part [spei_ln_bwd_blocks(M)][2][256]: per-block partial sums of dgamma (= dy * xhat) and dbeta (= dy); the caller adds the
 * blocks in index order. */
int64_t spei_ln_bwd_blocks(int64_t M);
int spei_layernorm256_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* part, int64_t M, spei_stream_t stream);

/* nn.GELU (erf form) on a saved pre-activation, and dpre = dy * gelu'(pre).  n floats, n % 4 == 0. */
int spei_gelu_fwd(const float* pre, float* out, int64_t n, spei_stream_t stream);
int spei_gelu_bwd(const float* pre, const float* dy, float* dpre, int64_t n, spei_stream_t stream);

/* WindowAttention core backward (model/swinir.py:115-149): q [H*W][256] pre-scaled, kv [H*W][512], relbias [8][25][25], dout
 * [H*W][256] = gradient of spei_window_attention's output -> dq [H*W][256], dkv [H*W][512] and dbias_part [nwin][8][25][25]
 * (the gradient of relbias is the sum over the windows).  Same window partition / cyclic shift / mask as the forward.  `batch`
 * equally sized maps stored one after the other are processed in one launch (dbias_part [batch][nwin][8][25][25]). */
int spei_window_attention_bwd(const float* q, const float* kv, const float* relbias, const float* dout, float* dq, float* dkv,
                              float* dbias_part, int H, int W, int shift, int batch, spei_stream_t stream);

/* out[m][n] = x[m][n] * rowscale[m] (the DropPath factor of model/swinir.py:278-279 applied to a branch gradient).  N % 4 == 0. */
int spei_scale_rows(const float* x, const float* rowscale, float* out, int64_t M, int N, spei_stream_t stream);

/* ---- backward of SearchTransfer / SelfTransfer and the decoder glue (model/SearchTransfer.py:24-79, model/speinet.py:92-120 under
 * loss.backward(), trainer/trainer_swint_hsa_nsf.py:34-40).  Gather form, fixed summation order. ---- */

/* Gradient of S (the maximal normalised 3x3-patch correlation, spei_corr_argmax) with respect to the query map lr [H*W][C]:
 * dlr from dS [H*W], the forward's S / arg and the two inverse patch norms (spei_patch_invnorm).  C % 4 == 0. */
int spei_corr_s_bwd_lr(const float* lr, const float* ref, const float* inv_lr, const float* inv_ref, const float* S, const int32_t* arg,
                       const float* dS, float* dlr, int H, int W, int Hr, int Wr, int C, spei_stream_t stream);

/* Gradient with respect to a REFERENCE map at scale s in {1,2,4} ([Hr3*s * Wr3*s][C]): the adjoint of spei_gather_fold (dT
 * [H3*s * W3*s][C], or NULL) plus, at s == 1, the gradient of S through the reference patches (dS, or NULL; then ref / lr /
 * inv_lr / inv_ref / S are read).  order [H3*W3]: the queries sorted by arg (stable); start [Hr3*Wr3 + 1]: first entry of each
 * reference position's list. */
int spei_search_bwd_ref(const float* ref, const float* dT, const float* lr, const float* inv_lr, const float* inv_ref, const float* S,
                        const float* dS, const int32_t* order, const int32_t* start, float* dref, int H3, int W3, int Hr3, int Wr3,
                        int C, int s, spei_stream_t stream);

/* Adjoint of spei_upsample_bicubic (act NONE): dy [H*s * W*s][C] -> dx [H*W][C], s in {2,4}. */
int spei_upsample_bicubic_bwd(const float* dy, float* dx, int H, int W, int C, int s, spei_stream_t stream);

/* out[m] = sum_n a[m][n] * b[m][n] (the gradient of a per-row scale such as `* weight_S`, model/speinet.py:93). */
int spei_rowdot(const float* a, const float* b, float* out, int64_t M, int N, spei_stream_t stream);

/* Row a11 — LD sharpness detector features (inference_SPEINet.py:54-189).  spei_det_gray: [N][3][H][W] fp32 0..255 ->
 * gray [N][H][W] in 0..1 (ITU-R 601 weights).  spei_det_features: gray -> out [N][6] = LAP1, MIS3, WAV1, GRA7, STA3, DCT3
 * with window size k (odd; the reference uses 11).  ws: spei_det_ws_floats(N,H,W,k) floats. */
int spei_det_gray(const float* rgb, float* gray, int N, int H, int W, spei_stream_t stream);
int64_t spei_det_ws_floats(int N, int H, int W, int k);
int spei_det_features(const float* gray, float* out, float* ws, int N, int H, int W, int k, spei_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
