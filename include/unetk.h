/* libunetk: C ABI of the MI355X (gfx950) kernels behind the UNet / Swin-UNet hot path.
 *
 * The reference project is pure Python on PyTorch and has no FFI layer of its own; the
 * entry points below are what its nn.Module forwards (and autograd's backwards of them)
 * dispatch to, one family per ATen op the hot path uses.  Each declaration names the
 * reference lines whose arithmetic it replaces.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless named host_*; the caller owns all buffers,
 *    workspaces included (query the *_bytes / *_elems functions).  Nothing here
 *    allocates, frees or synchronises; kernels are enqueued on `stream`.
 *  - Activations are NHWC ("channels last") with an explicit channel stride `ld*`
 *    (elements between consecutive pixels), element type `dtype`.  Parameters,
 *    statistics, gradients of parameters and model inputs/outputs are fp32; model
 *    inputs/outputs are NCHW as in the reference.
 *  - Return value: UNETK_OK or a negative error code; unetk_last_error() returns the
 *    message of the calling thread's last failure.
 */
#ifndef UNETK_H
#define UNETK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* unetk_stream_t; /* == hipStream_t */

enum { UNETK_OK = 0, UNETK_EINVAL = -1, UNETK_ESHAPE = -2, UNETK_EHIP = -3 };
enum { UNETK_F32 = 0, UNETK_BF16 = 1 };
enum { UNETK_ACT_NONE = 0, UNETK_ACT_RELU = 1, UNETK_ACT_GELU = 2 };

const char* unetk_last_error(void);
int unetk_version(void);
/* Stream `to` waits for everything queued so far on stream `from` (both on the current device): hipEventRecord +
 * hipStreamWaitEvent on an event from a ring the library keeps per device.  No counterpart in the reference (it has no
 * second stream); the weight-gradient stream of the backward passes is forked with it at ~2 us per call, where the
 * framework's event objects cost the host ~15 us. */
int unetk_stream_fork(unetk_stream_t from, unetk_stream_t to);

/* ---- convolution as implicit GEMM on MFMA ------------------------------------------
 * nn.Conv2d(k=3, padding=1) / nn.Conv2d(k=1) / nn.Linear:
 *   training.py:524,527 (DoubleConv), swin_unet_model.py:506,512,515,608 (decoder),
 *   swin_unet_model.py:234,236,298,301,383 (Swin linears as 1x1).
 * Weights are re-packed from the reference [Co][Ci][k][k] fp32 layout into the kernel's
 * [Co/bn][Ci/kc][tap][bn][kc] layout; transpose=1 packs the data-gradient convolution
 * (in/out swapped, taps flipped), which is then run with unetk_conv_fwd as well. */
int unetk_conv_pick_bn(int Co);
int64_t unetk_conv_packed_elems(int Co, int Ci, int ks, int bn, int dtype);
int unetk_conv_pack_weight(const float* w, void* packed, int Co, int Ci, int ks, int bn, int dtype,
                           int transpose, unetk_stream_t stream);
/* Many weights in one launch.  jobs (device): 9 int64 per job = fp32 source pointer, destination
 * offset in elements from dst_base (16-byte aligned), Co, Ci, ks, bn, ceil(packed Ci / channels-per-chunk), transpose
 * (+2: the source is a ConvTranspose2d(k=2, s=2) weight [Ci][Co/4][2][2], swin_unet_model.py:501, taken as the
 * matrix [(ky*2+kx)*Co/4 + co][ci] that unetk_convt2x2_fwd multiplies by; ks = 1), packed
 * element count (unetk_conv_packed_elems); Co/Ci are the SOURCE tensor's.  blocks (device): per
 * workgroup two int32 = (job, group inside the job); nblocks of them.  A job has
 * unetk_conv_pack_groups(packed Co, packed Ci, ks, bn, dtype) groups (packed = swapped when transpose=1),
 * numbered from 0. */
int unetk_conv_pack_groups(int Co, int Ci, int ks, int bn, int dtype);
int unetk_conv_pack_weights(const int64_t* jobs, const int* blocks, int nblocks, void* dst_base, int dtype,
                            unetk_stream_t stream);
int unetk_conv_num_tiles(int B, int H, int W, int ks);
/* Rows of batch-statistics partials ([rows][2][Co] floats) that unetk_conv_fwd writes for this problem when it
 * is given a stats buffer and no residual: one per pixel tile, or one per workgroup where the persistent
 * kernel for the full-resolution layers (3x3, bf16, 33..48 input and up to 48 output channels) takes it. */
int unetk_conv_stats_rows(int B, int H, int W, int Ci, int Co, int ks, int bn, int dtype, int ldin, int act);
/* out = act((conv(in, w)) * ep_scale + ep_shift) [* resid_keep[b] + resid].
 * stats (nullable): [unetk_conv_num_tiles][2][Co] per-tile sum / sum of squares of the raw
 * convolution output (fp32 accumulators), the BatchNorm batch statistics of
 * training.py:525,528 before their deterministic reduction by unetk_bn_finalize. */
int unetk_conv_fwd(const void* in, int ldin, const void* wpacked, void* out, int ldout, int B, int H,
                   int W, int Ci, int Co, int ks, int bn, int dtype, float* stats,
                   const float* ep_scale, const float* ep_shift, int act, const void* resid, int ldres,
                   const float* resid_keep, unetk_stream_t stream);
/* Same, with the output channels [split, Co) written to a second tensor out2 (row stride ldout2,
 * channels renumbered from 0); split must be a multiple of bn.  The data gradient of a conv whose
 * input was torch.cat([up, skip], dim=1) (training.py:492-505) comes out as the two gradients. */
int unetk_conv_fwd_split(const void* in, int ldin, const void* wpacked, void* out, int ldout, void* out2,
                         int ldout2, int split, int B, int H, int W, int Ci, int Co, int ks, int bn,
                         int dtype, float* stats, const float* ep_scale, const float* ep_shift, int act,
                         const void* resid, int ldres, const float* resid_keep, unetk_stream_t stream);
/* Same again, with inference-time fusions for 3x3 convs: pooled (nullable) receives the 2x2
 * max-pool of the stored output (training.py:464-473, dense [B][H/2][W/2][Co]); head_out (nullable,
 * NCHW fp32 [B][head_co][H][W]) receives the network's final 1x1 conv with bias applied to the
 * stored output (training.py:482,520; needs Co <= bn).  out may be NULL when only head_out is wanted. */
/* unetk_conv_fwd (3x3) whose input is the channel concatenation [in | in2] of two NHWC tensors, read in place:
 * the decoder's torch.cat((dec, enc), dim=1) before its first conv (training.py:504-517) is never stored.
 * Ci1 channels come from `in` (a multiple of 32 for bf16, 16 for f32), Ci - Ci1 from `in2`; the packed weights
 * are those of the concatenated layer. */
int unetk_conv_fwd_cat(const void* in, int ldin, int Ci1, const void* in2, int ldin2, const void* wpacked, void* out,
                       int ldout, int B, int H, int W, int Ci, int Co, int bn, int dtype, float* stats,
                       const float* ep_scale, const float* ep_shift, int act, unetk_stream_t stream);
/* Data gradient of a 3x3 conv (weights packed transposed, as for unetk_conv_fwd) whose result g is the gradient of
 * the activation relu(y*scale+shift) of the BatchNorm in front (training.py:523-529, the second conv of a block in
 * the backward): besides storing g (dense, Co channels) the launch leaves the two per-channel sums that BatchNorm's
 * backward needs, partials[unetk_conv_num_tiles(B,H,W,3)][2][Co] = sum of dz and sum of dz*(y-mean)*invstd with
 * dz = g where y*scale+shift > 0, formed from the values as stored; unetk_bn_act_bwd_presummed takes them in place of
 * its own reduction pass over g and y.  bfloat16 with a channel block of 64 or 96 (see ..._supported). */
int unetk_conv_bwd_data_bnsums_supported(int Co, int bn, int dtype);
int unetk_conv_bwd_data_bnsums(const void* dy, int lddy, const void* wpackedT, void* out, int B, int H, int W, int Ci,
                               int Co, int bn, int dtype, const void* y, int ldy, const float* scale, const float* shift,
                               const float* mean, const float* invstd, float* partials, unetk_stream_t stream);
/* ... and its weight gradient (workspace as unetk_conv_wgrad_workspace_bytes for the whole Ci). */
int unetk_conv_wgrad_cat(const void* in, int ldin, int Ci1, const void* in2, int ldin2, const void* dy, int lddy,
                         float* dw, int B, int H, int W, int Ci, int Co, int dtype, void* workspace,
                         int64_t workspace_bytes, int accumulate, unetk_stream_t stream);
int unetk_conv_fwd_fused(const void* in, int ldin, const void* wpacked, void* out, int ldout, void* out2,
                         int ldout2, int split, void* pooled, const float* head_w, const float* head_b,
                         float* head_out, int head_co, int B, int H, int W, int Ci, int Co, int ks, int bn,
                         int dtype, float* stats, const float* ep_scale, const float* ep_shift, int act,
                         const void* resid, int ldres, const float* resid_keep, unetk_stream_t stream);
/* dw[Co][Ci][k][k] (+)= sum_pixels dy (x) shifted in   (autograd of the convolutions above) */
int64_t unetk_conv_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co, int ks, int dtype);
int unetk_conv_wgrad(const void* in, int ldin, const void* dy, int lddy, float* dw, int B, int H, int W,
                     int Ci, int Co, int ks, int dtype, void* workspace, int64_t workspace_bytes,
                     int accumulate, unetk_stream_t stream);
/* nn.Linear backward over tokens (swin_unet_model.py:234,236,298,301 through autograd): weight
 * gradient dw[Co][Ci] as the 1x1 case above AND the bias gradient dbias[Co] = sum over tokens of
 * dy, from the same pass over dy (the dy fragments the MFMAs load are multiplied by a matrix of
 * ones as well).  bfloat16 only.  Workspace: unetk_linear_wgrad_workspace_bytes. */
int64_t unetk_linear_wgrad_workspace_bytes(int64_t ntok, int Ci, int Co, int dtype);
int unetk_linear_wgrad(const void* in, int ldin, const void* dy, int lddy, float* dw, float* dbias, int64_t ntok,
                       int Ci, int Co, int dtype, void* workspace, int64_t workspace_bytes, int accumulate,
                       unetk_stream_t stream);

/* nn.Linear over tokens, forward and data gradient (swin_unet_model.py:234, 236, 298-301, 383, 501, 506):
 *   out[t][n] = epilogue(sum_k x[t][k] * w[n][k]),  t < M, with w packed as for unetk_conv_fwd (ks = 1,
 *   channel block bn; transpose = 1 for the data gradient).  bfloat16 only: unetk_linear_supported says
 *   whether a shape can take this path (else use unetk_conv_fwd with ks = 1).
 * Epilogue, every part optional: + bias[n]; act = UNETK_ACT_GELU, with out2 given the pre-activation goes
 * to out and GELU(pre-activation) to out2 (the MLP's hidden layer, :298-299); gelu_grad_of: the result is
 * multiplied by GELU'(gelu_grad_of[t][n]) (data gradient through that GELU); resid (+ resid_keep per
 * sample of tok_per_sample tokens): out = resid + resid_keep[t / tok_per_sample] * value (:371-372). */
int unetk_linear_supported(int64_t M, int K, int N, int bn, int dtype);
int unetk_linear_fwd(const void* x, int ldx, const void* wpacked, void* out, int ldo, void* out2, int ldo2,
                     int64_t M, int K, int N, int bn, int dtype, const float* bias, int act, const void* resid,
                     int ldr, const float* resid_keep, int64_t tok_per_sample, const void* gelu_grad_of, int ldaux,
                     unetk_stream_t stream);

/* The MLP branch of a Swin block in one launch (swin_unet_model.py:372 with :296-303 and nn.LayerNorm):
 *   ln = LayerNorm(x) (gamma, beta, eps; mean / rstd per token kept for the backward),
 *   hpre = ln @ w1^T + b1,  hact = GELU(hpre as stored),  out = x + keep[t / tok_per_sample] * (hact @ w2^T + b2).
 * x, ln, out: dense [M][C]; hpre, hact: dense [M][4C] (hact nullable: then GELU(hpre) is not stored).  w1: the
 * (4C, C) weight packed as for unetk_conv_fwd with ks = 1 and channel block 64; w2: the (C, 4C) weight packed
 * with channel block 96.  bfloat16, C = 96 or 192 (the stages whose hidden tensor is bound by HBM): ask
 * unetk_swin_mlp_supported; other shapes use unetk_layernorm_fwd + unetk_linear_fwd + unetk_gelu_fwd. */
int unetk_swin_mlp_supported(int64_t M, int C, int dtype);
int unetk_swin_mlp_fwd(const void* x, const float* gamma, const float* beta, float eps, const void* w1_packed64,
                       const float* b1, const void* w2_packed96, const float* b2, const float* keep,
                       int64_t tok_per_sample, void* ln, float* mean, float* rstd, void* hpre, void* hact, void* out,
                       int64_t M, int C, int dtype, unetk_stream_t stream);

/* The data-gradient chain of that branch in one launch (autograd of the lines above): dy = gradient of the
 * branch's output with its DropPath factor applied, [M][C];
 *   dhp = (dy @ w2) * GELU'(hpre)                      [M][4C], stored: the weight gradient of fc1 reads it
 *   dx  = LayerNorm'(dhp @ w1; x, mean, rstd, gamma) + add     (add nullable)
 *   dx_scaled = dx * keep[t / tok_per_sample]          (both nullable together)
 *   partials[row][0][c], [row][1][c] = this workgroup's sums over tokens of d ln * xhat and d ln: their column
 *   sums over the unetk_swin_mlp_bwd_rows(M, C) rows are the LayerNorm's weight and bias gradients
 *   (unetk_partial_sums_batched).
 * w2t: the (C, 4C) fc2 weight in its data-gradient pack (transpose = 1) with channel block 64; w1t: the (4C, C)
 * fc1 weight in its data-gradient pack with channel block 96.  Shapes as unetk_swin_mlp_supported. */
int unetk_swin_mlp_bwd_rows(int64_t M, int C);
int unetk_swin_mlp_bwd(const void* dy, const void* hpre, const void* w2t_packed64, const void* w1t_packed96,
                       const void* x, const float* mean, const float* rstd, const float* gamma, const void* add,
                       const float* keep, int64_t tok_per_sample, void* dhp, void* dx, void* dx_scaled,
                       float* partials, int64_t M, int C, int dtype, unetk_stream_t stream);

/* Weight and bias gradient of a token linear (autograd of the products above):
 *   dw[n][k] (+)= sum_t dy[t][n] * x[t][k],   dbias[n] (+)= sum_t dy[t][n]  (dbias nullable).
 * bfloat16 operands, fp32 results; workspace (fp32 partial tiles of the token ranges, summed in a fixed
 * order) sized by unetk_token_wgrad_workspace_bytes. */
int unetk_token_wgrad_supported(int64_t M, int K, int N, int dtype);
int64_t unetk_token_wgrad_workspace_bytes(int64_t M, int K, int N);
int unetk_token_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, int64_t M, int K,
                      int N, int dtype, void* workspace, int64_t workspace_bytes, int accumulate,
                      unetk_stream_t stream);

/* ---- layout ------------------------------------------------------------------------ */
int unetk_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int ld, int dtype,
                       unetk_stream_t stream);
int unetk_nhwc_to_nchw(const void* src, int ld, float* dst, int B, int C, int H, int W, int dtype,
                       unetk_stream_t stream);

/* ---- BatchNorm2d + ReLU + Dropout2d mask + MaxPool2d(2,2) ---------------------------
 * training.py:525-530 and :464-473; swin_unet_model.py:513-517, 609-610.
 * unetk_bn_finalize consumes `partials` (long lists are first reduced in place). */
int unetk_bn_finalize(float* partials, int nparts, int C, int64_t count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      unetk_stream_t stream);
int unetk_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, const float* conv_bias, float* scale,
                       float* shift, int C, unetk_stream_t stream);
/* a = relu(y*scale+shift) * keep[b][c]; pooled (nullable) = 2x2 max of a.
 * scale/shift null = identity, a null = only the pooled output is written (plain MaxPool2d).
 * y has row stride ldy; a has row stride lda (it may be the skip slice of the decoder's concat buffer, which saves the
 * torch.cat copy of training.py:505).  argmax (nullable, [B][H/2][W/2][C] bytes) receives the
 * position 0..3 of the first maximum of each quad, for unetk_pool_bwd. */
int unetk_bn_act_fwd(const void* y, int ldy, void* a, int lda, void* pooled, unsigned char* argmax, int B, int H,
                     int W, int C, const float* scale, const float* shift, const float* keep, int relu,
                     int dtype, unetk_stream_t stream);
int unetk_reduce_num_partials(int64_t npix);
/* dy = BN'(relu'(keep * g)); dgamma/dbeta (nullable) (+)=.
 * scratch owned by the caller: partials [num_partials][2][C] floats, sums [6][C] floats */
int unetk_bn_act_bwd(const void* g, const void* y, void* dy, int B, int H, int W, int C,
                     const float* scale, const float* shift, const float* mean, const float* invstd,
                     const float* keep, int relu, float* dgamma, float* dbeta, int accumulate,
                     float* partials, float* sums, int dtype, unetk_stream_t stream);
/* Same (ReLU, no keep factors) with the reduction already done: partials [nparts][2][C] as unetk_conv_bwd_data_bnsums
 * leaves them.  sums as above. */
int unetk_bn_act_bwd_presummed(const void* g, const void* y, void* dy, int B, int H, int W, int C, const float* scale,
                               const float* shift, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                               int accumulate, const float* partials, int nparts, float* sums, int dtype,
                               unetk_stream_t stream);
/* Same with a rank-one gradient that is never stored: g[b,y,x,c] = gpix[b,y,x] * gchan[c] (rounded to the
 * compute type as the stored tensor would be).  The gradient of the last activation of the UNet is of this
 * form: dout (B,1,H,W) times the weight of the final 1x1 conv (training.py:482, 520). */
int unetk_bn_act_bwd_rank1(const float* gpix, const float* gchan, const void* y, void* dy, int B, int H, int W, int C,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* keep, int relu, float* dgamma, float* dbeta, int accumulate,
                           float* partials, float* sums, int dtype, unetk_stream_t stream);
/* Same with the gradient that comes back through a 2x2 max-pool AND a skip connection (the last activation of an
 * encoder block, training.py:497-503, 524-531 in the backward): g = dskip + route(dpooled by argmax), i.e. what
 * unetk_pool_bwd would store (same rounding), formed inside both passes instead of written once and read twice.
 * dskip has row stride ldskip; argmax is the map unetk_bn_act_fwd left. */
int unetk_bn_act_bwd_pool(const void* dpooled, const unsigned char* argmax, const void* dskip, int ldskip,
                          const void* y, void* dy, int B, int H, int W, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* keep, int relu, float* dgamma,
                          float* dbeta, int accumulate, float* partials, float* sums, int dtype,
                          unetk_stream_t stream);
/* da = dskip + route(dpooled to the first maximum of each 2x2 quad): the maximum is taken from
 * argmax when given (a may then be null), else recomputed from a. */
int unetk_pool_bwd(const void* a, const unsigned char* argmax, const void* dpooled, const void* dskip,
                   int ldskip, void* da, int B, int H, int W, int C, int dtype, unetk_stream_t stream);

/* ---- F.interpolate(bilinear, align_corners=False) to the skip's size + torch.cat -----
 * training.py:504-517.  out has C1+C2 channels per pixel; with skip == NULL only the resized
 * channels [0,C1) are written (the skip half was put there by unetk_bn_act_fwd). */
int unetk_upsample_concat_fwd(const void* low, int h, int w, int C1, const void* skip, int ldskip, int C2,
                              void* out, int B, int H, int W, int dtype, unetk_stream_t stream);
int unetk_upsample_bwd(const void* dout, int ldout, int H, int W, void* dlow, int B, int h, int w, int C1,
                       int dtype, unetk_stream_t stream);
/* The last BatchNorm + ReLU + Dropout2d of a block and the exact 2x resize in front of the next decoder block
 * (training.py:522-531 followed by :504-517) in one pass: out[b, 2h x 2w, 0:C1) = resize(relu(y*scale+shift)*keep),
 * pixel stride ldo elements (C1 for a dense result, C1+C2 for the concat buffer); the activation itself is not
 * stored (nothing else reads it).  keep: (B, C1) fp32 factors or NULL.  bfloat16, C1 a multiple of 32; the same
 * bits as unetk_bn_act_fwd followed by unetk_upsample_concat_fwd. */
int unetk_bn_act_upsample2x_supported(int h, int w, int C1, int dtype);
int unetk_bn_act_upsample2x(const void* y, int B, int h, int w, int C1, const float* scale, const float* shift,
                            const float* keep, void* out, int ldo, int dtype, unetk_stream_t stream);

/* F.interpolate(outputs, size=labels.shape[-2:], mode='bilinear', align_corners=False) on the fp32
 * NCHW prediction before the loss (swin_unet_model.py:812-814): in (planes,h,w) -> out (planes,H,W),
 * planes = B*C.  backward=1: in is the gradient of the (planes,H,W) result and out receives the
 * gradient of the (planes,h,w) source (gather form, deterministic). */
int unetk_resize_planes_f32(const float* in, float* out, int64_t planes, int h, int w, int H, int W, int backward,
                            unetk_stream_t stream);

/* ---- final nn.Conv2d(C, Co, 1) with bias, NHWC in -> NCHW fp32 out --------------------
 * training.py:482,520; swin_unet_model.py:611 */
int unetk_head1x1_fwd(const void* x, int ldx, const float* w, const float* bias, float* out, int B, int H,
                      int W, int C, int Co, int dtype, unetk_stream_t stream);
int unetk_head1x1_bwd(const void* x, int ldx, const float* w, const float* dout, void* dx, float* dw,
                      float* db, int B, int H, int W, int C, int Co, int accumulate, float* partials,
                      int dtype, unetk_stream_t stream);

/* ---- multi-loss reductions ----------------------------------------------------------
 * TIDWeightedLoss / GradientLoss / SpectralLoss / MultiLossFunction,
 * enhanced_training_multi_loss.py:143-416 (== training.py:158-365).  pred/target/w are fp32
 * (B,1,H,W); w may be null.  `res` is a 16-float device record: [0] pixel [1] gradient
 * [2] spectral [3] total, [4..11] normalisation coefficients consumed by the backward kernels.
 * pixel_mae: 0 mse / 1 mae; grad_method: 0 none / 1 sobel / 2 diff. */
int unetk_loss_num_partials(int B, int H, int W);
int unetk_loss_spectral_blocks(int H, int W);
/* partials: [num_partials][11] floats; gx, gy: (B,H,W) Sobel maps of pred-target (sobel only) */
int unetk_loss_pixgrad_fwd(const float* pred, const float* target, const float* w, int B, int H, int W,
                           int pixel_mae, int grad_method, int pixel_use_w, float tid_scale, float* gx,
                           float* gy, float* partials, float* res, unetk_stream_t stream);
/* unetk_loss_pixgrad_fwd for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): the partial sums live in a 45 KB
 * device array inside the library, one copy per GPU, whose address is looked up once per device.
 * This entry point has no counterpart in the reference.  It allocates nothing and gives the same
 * bits as unetk_loss_pixgrad_fwd.  B*H*W may be at most unetk_loss_static_max_pixels() (2^30);
 * larger inputs are refused with UNETK_EINVAL.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (different streams) race; calls on one stream are ordered.
 * The GPU that owns the inputs and the stream must be the CURRENT device: a stream that belongs to
 * another device is refused with UNETK_EINVAL (its kernels would be handed this device's array). */
int64_t unetk_loss_static_max_pixels(void);
int unetk_loss_pixgrad_fwd_static(const float* pred, const float* target, const float* w, int B, int H, int W,
                                  int pixel_mae, int grad_method, int pixel_use_w, float tid_scale, float* gx,
                                  float* gy, float* res, unetk_stream_t stream);
/* Fp, Ft: complex64 fft2 of pred / target (B,H,W); partials, wsum: [B][spectral_blocks]; coef: [B] */
int unetk_loss_spectral_fwd(const void* Fp, const void* Ft, const float* w, int B, int H, int W, float factor,
                            float* partials, float* wsum, float* coef, float* res, unetk_stream_t stream);
int unetk_loss_combine(float* res, float w_pixel, float w_grad, float w_spec, int use_grad, int use_spec,
                       unetk_stream_t stream);
/* dpred (+)= gout[0] * (w_pixel * d pixel/d pred + w_grad * d gradient/d pred) */
int unetk_loss_pixgrad_bwd(const float* pred, const float* target, const float* w, int B, int H, int W,
                           int pixel_mae, int grad_method, const float* gx, const float* gy, const float* res,
                           const float* gout, float w_pixel, float w_grad, int use_grad, float* dpred,
                           int accumulate, unetk_stream_t stream);
/* Z (complex64, B,H,W): d spectral/d pred = Re(fft2(Z)) */
int unetk_loss_spectral_bwd(const void* Fp, const void* Ft, int B, int H, int W, float factor,
                            const float* coef, float w_spec, const float* gout, void* Z,
                            unetk_stream_t stream);

/* ---- tiled large-area inference -------------------------------------------------------
 * LargeAreaPredictor.predict_large_area, large_area_predictor.py:151-216.
 * feats: (nvar, lat, lon) fp32.  Tiles start at (ti*S, tj*S), ti < nti, tj < ntj, lat-major. */
/* valid[ti*ntj+tj] = 0 when some variable is all-NaN in the window (:168-170) */
int unetk_tile_valid(const float* feats, int nvar, int lat, int lon, int P, int S, int nti, int ntj,
                     int* valid, unetk_stream_t stream);
/* pos: n (lat_start, lon_start) pairs -> NHWC batch (n,P,P,ld), NaN -> 0 (:178-179) */
int unetk_tile_gather(const float* feats, int nvar, int lat, int lon, const int* pos, int n, int P, void* out,
                      int ld, int dtype, unetk_stream_t stream);
/* Adds tiles [t0,t1) of the compacted tile list (pred: (t1-t0,P,P) fp32; slot[ti*ntj+tj] = index in
 * that list or -1) into acc/cnt for rows [row0,row1), per pixel in tile order (:202-212). */
int unetk_blend_accumulate(const float* pred, const float* wk, const int* slot, int t0, int t1, int nti,
                           int ntj, int P, int S, float* acc, float* cnt, int lat, int lon, int row0, int row1,
                           unetk_stream_t stream);
/* out = acc / (cnt > 1e-9 ? cnt : 1)  (:215-216) */
int unetk_blend_normalize(const float* acc, const float* cnt, float* out, int64_t n, unetk_stream_t stream);

/* ---- Swin-UNet pieces -------------------------------------------------------------------
 * Tokens are (B,H,W,C) NHWC.  nn.LayerNorm: swin_unet_model.py:289,295,384,562. */
int unetk_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                        int64_t ntok, int C, float eps, int dtype, unetk_stream_t stream);
int unetk_token_reduce_num_partials(int64_t ntok);
/* dx = LN'(dy) + add (add nullable: the residual path's gradient); partials: [num_partials][2][C] */
int unetk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const void* add, void* dx, float* dgamma, float* dbeta, float* partials, int64_t ntok,
                        int C, int dtype, unetk_stream_t stream);
/* The same backward in two halves, so that the small column sums of all LayerNorms of a backward pass
 * (swin_unet_model.py:289,295: two per block) can be taken by ONE launch at its end.
 * unetk_layernorm_bwd_partials writes dx and unetk_layernorm_bwd_rows(ntok, C, dtype) rows of partials
 * [row][2][C] (0 rows: the shape has no one-pass form, use unetk_layernorm_bwd);
 * unetk_partial_sums_batched sums njobs such tables: jobs (device) = five int64 per job: partials, rows, C,
 * dgamma, dbeta (0 = not wanted); max_C = the largest C among them.  Same bits as unetk_layernorm_bwd.
 * dx_scaled (nullable): a second output, dx * keep[token / tok_per_sample] - the DropPath factor (:139-148) of the
 * branch whose backward reads this gradient next (same bits as unetk_scale_sample applied to dx). */
int unetk_layernorm_bwd_rows(int64_t ntok, int C, int dtype);
int unetk_layernorm_bwd_partials(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, const void* add, void* dx, float* partials, int64_t ntok, int C,
                                 int dtype, const float* keep, int64_t tok_per_sample, void* dx_scaled,
                                 unetk_stream_t stream);
int unetk_partial_sums_batched(const int64_t* jobs, int njobs, int max_C, unetk_stream_t stream);
/* bias gradients of the linears: out[c] = sum over tokens */
int unetk_channel_sum(const void* x, int ld, int64_t ntok, int C, float* out, float* partials, int dtype,
                      unetk_stream_t stream);
/* nn.GELU (erf form), swin_unet_model.py:299 */
int unetk_gelu_fwd(const void* x, void* y, int64_t n, int dtype, unetk_stream_t stream);
int unetk_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, unetk_stream_t stream);
/* DropPath scale per sample (:139-148) with optional add: y = x*keep[b] + add */
int unetk_scale_sample(const void* x, const float* keep, const void* add, void* y, int B, int64_t per_sample,
                       int dtype, unetk_stream_t stream);
/* patch embedding gather for Conv2d(k=ps, s=ps) (:559-560, 634): NCHW fp32 -> (B,H/ps,W/ps,ld) */
int unetk_patchify(const float* x, void* tok, int B, int Cin, int H, int W, int ps, int ld, int dtype,
                   unetk_stream_t stream);
int unetk_unpatchify(const void* dtok, float* dx, int B, int Cin, int H, int W, int ps, int ld, int dtype,
                     unetk_stream_t stream);
/* PatchMerging 2x2 gather, concat order (0,0)(1,0)(0,1)(1,1), zero pad for odd sizes (:396-417);
 * backward=1: x is the gradient of the merged (B,ceil(H/2),ceil(W/2),4C) tensor, out that of the input */
int unetk_merge_gather(const void* x, void* out, int B, int H, int W, int C, int backward, int dtype,
                       unetk_stream_t stream);
/* ConvTranspose2d(k=2,s=2) (:501,521) = GEMM to (B,h,w,4*Co) then this shuffle (+bias) into channels
 * [0,Co) of the (B,2h,2w,ldcat) concat buffer; backward=1 gathers the gradient back into tmp */
int unetk_pixel_shuffle(void* tmp, void* cat, int ldcat, const float* bias, int B, int h, int w, int Co,
                        int backward, int dtype, unetk_stream_t stream);
/* ConvTranspose2d(k=2,s=2) + bias forward (:501,521) in one launch: the token GEMM x (B,h,w,K) times the weight packed
 * as for unetk_linear_fwd from the (4*up, K) matrix whose row (ky*2 + kx)*up + co is weight[:, co, ky, kx], with the
 * pixel shuffle in the epilogue's store address: out[b][2y+ky][2x+kx][co] (channels [0,up) of a (B,2h,2w,ldo) buffer,
 * e.g. the first half of the decoder's concat buffer).  bfloat16; shapes as unetk_linear_supported(B*h*w, K, 4*up, bn). */
int unetk_convt2x2_fwd(const void* x, int ldx, const void* wpacked, void* out, int ldo, int B, int h, int w, int K,
                       int up, int bn, int dtype, const float* bias, unetk_stream_t stream);
/* Its data gradient in one launch: dx[b][y][x][:] = sum over (ky, kx, co) of dout[b][2y+ky][2x+kx][co] * W, a GEMM
 * with K = 4 * up whose operand rows are gathered from the gradient image dout (B, 2h, 2w, lddo >= up) - the inverse
 * pixel shuffle is the operand's load address.  wpackedT: the data-gradient pack of the (4 * up, N) matrix
 * (unetk_conv_pack_weights, transpose + ConvTranspose form); up a multiple of 32; dx (B, h, w, ldo >= N). */
int unetk_convt2x2_bwd_data(const void* dout, int lddo, const void* wpackedT, void* dx, int ldo, int B, int h, int w,
                            int up, int N, int bn, int dtype, unetk_stream_t stream);
/* WindowAttention + cyclic shift + window partition/reverse (:161-200, 241-269, 339-360), window 8,
 * head dim 32.  qkv: (B,He,We,3C) from the qkv linear; out: (B,He,We,C) before the proj linear.
 * table: relative_position_bias_table (225, heads) fp32. */
int unetk_window_attn_fwd(const void* qkv, void* out, const float* table, int B, int He, int We, int C, int heads,
                          int shift, int use_mask, float scale, int dtype, unetk_stream_t stream);
int unetk_window_attn_bwd(const void* qkv, const void* dout, void* dqkv, const float* table, float* dtable,
                          float* dtable_partials, int B, int He, int We, int C, int heads, int shift, int use_mask,
                          float scale, int dtype, unetk_stream_t stream);
/* The same attention for any window side and head dimension (swin_unet_model.py:161-200, 241-269 accept any; the two
 * entry points above are the MFMA kernels for window 8 / head dimension 32): window side 2...16, head dimension up to
 * 64, window tokens x head dimension up to 9216 (unetk_window_attn_any_supported).  table: [(2*ws-1)^2][heads];
 * dtable_partials: [B*nW*heads][(2*ws-1)^2] floats, one row per (window, head) in launch order (window-major, head
 * fastest): the caller adds the rows of a head. */
int unetk_window_attn_any_supported(int ws, int head_dim);
int unetk_window_attn_any_fwd(const void* qkv, void* out, const float* table, int B, int He, int We, int C, int heads,
                              int ws, int shift, int use_mask, float scale, int dtype, unetk_stream_t stream);
int unetk_window_attn_any_bwd(const void* qkv, const void* dout, void* dqkv, const float* table, float* dtable_partials,
                              int B, int He, int We, int C, int heads, int ws, int shift, int use_mask, float scale,
                              int dtype, unetk_stream_t stream);
/* dtable_partials: [B * nW * heads][225].  dtable may be null: then the partials are summed later, for every
 * attention layer of the backward pass (:206-219, one table per block) by one launch of
 * unetk_dtable_reduce_batched: jobs (device) = four int64 per job: partials, B * nW, heads, dtable;
 * max_heads = the largest head count among them.  Same bits as the in-place sum. */
int unetk_dtable_reduce_batched(const int64_t* jobs, int njobs, int max_heads, unetk_stream_t stream);

/* ---- optimizer tail: clip_grad_norm_ + AdamW over a list of fp32 tensors ----------------
 * enhanced_training_multi_loss.py:758-763 (AdamW), :827-829 (clip_grad_norm_(1.0), step);
 * swin_unet_model.py:836-842.  Tensors are given as device arrays of pointers + sizes; `chunks`
 * is a device table of (tensor index, chunk index) pairs, one per workgroup, chunk =
 * unetk_optim_chunk_elems() elements.  unetk_grad_norm writes out[0] = global L2 norm and
 * out[1] = min(1, max_norm / (norm + 1e-6)) (partials: nchunks floats of scratch).
 * unetk_adamw_step applies torch.optim.AdamW's update for step number `step` (1-based); with
 * clip_coef (device pointer, nullable) the gradients are scaled by it first (and written back
 * when write_grad != 0). */
int unetk_optim_chunk_elems(void);
int unetk_grad_norm(const float* const* grads, const int64_t* sizes, const int32_t* chunks, int nchunks,
                    float max_norm, float* partials, float* out, unetk_stream_t stream);
int unetk_scale_list(float* const* tensors, const int64_t* sizes, const int32_t* chunks, int nchunks,
                     const float* coef, unetk_stream_t stream);
int unetk_adamw_step(float* const* params, float* const* grads, float* const* m, float* const* v,
                     const int64_t* sizes, const int32_t* chunks, int nchunks, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int64_t step, const float* clip_coef,
                     int write_grad, unetk_stream_t stream);

/* unetk_adamw_step for a step that is captured into a HIP graph and replayed: nothing that changes from step
 * to step may be a launch argument, so the step number and the learning rate live on the device.
 * hyper (5 doubles, device): {steps taken so far, lr, weight_decay, beta1, beta2}; each call adds one to
 * hyper[0] and derives the step's scalars into dyn (3 floats of device scratch) before the update. */
int unetk_adamw_step_device(float* const* params, float* const* grads, float* const* m, float* const* v,
                            const int64_t* sizes, const int32_t* chunks, int nchunks, double* hyper, float* dyn,
                            double eps, const float* clip_coef, int write_grad, double beta1, double beta2,
                            unetk_stream_t stream);

/* ---- training-patch extraction -------------------------------------------------------
 * EnhancedDataPreparation.extract_patches, enhanced_training_multi_loss.py:633-720: candidate
 * windows start at margin + k*S (margin = P/8) while start + P + margin <= size, row-major.
 * unetk_patch_qc writes flags[candidate] = 1 when the window passes the quality rules (:666-701).
 * unetk_patch_gather copies the windows at pos[k] = (lat_start, lon_start) into (n, nvar, P, P),
 * (n, 1, P, P), (n, 1, P, P); sanitize != 0 replaces NaN/inf by 0 and flips[k] in {0, 1 = last
 * axis reversed, 2 = rows reversed} applies the dataset's augmentation (:425-450, :484-489). */
int unetk_patch_qc(const float* feats, int nvar, const float* labels, const float* weights, int lat, int lon,
                   int P, int S, double min_valid, int* flags, unetk_stream_t stream);
int unetk_patch_gather(const float* feats, int nvar, const float* labels, const float* weights, int lat, int lon,
                       const int* pos, const int* flips, int n, int P, int sanitize, float* out_f, float* out_l,
                       float* out_w, unetk_stream_t stream);

/* ---- Gaussian low-pass of a float64 grid (long-wavelength background) -------------------
 * data_preparation.py:392-414 (apply_gaussian_filter: NaN -> 0, scipy.ndimage.gaussian_filter with
 * mode='reflect', NaN restored) is two passes of unetk_gauss_rows_f64 (which filters along the
 * contiguous axis) around two transposes.  wpad: the 2*radius+1 correlation weights followed by
 * zeros up to a multiple of 16 entries.  zero_nan: treat NaN inputs as 0.  nan_src (rows x cols,
 * or null): where it holds NaN the output is NaN.  in and out must not alias. */
int unetk_gauss_rows_max_radius(void);
int unetk_gauss_rows_f64(const double* in, double* out, const double* nan_src, int rows, int cols,
                         const double* wpad, int radius, int zero_nan, unetk_stream_t stream);
int unetk_transpose_f64(const double* in, double* out, int rows, int cols, unetk_stream_t stream);

/* ---- validation metrics as streaming reductions ---------------------------------------
 * enhanced_training_multi_loss.py:883-889 (np.corrcoef over the concatenated validation set),
 * predict.py:188-200 (MSE, MAE, Pearson r).  state: 8 doubles on the device, zeroed by the
 * caller: count, sum p, sum t, sum p^2, sum t^2, sum p*t, sum |p-t|, sum (p-t)^2; each call adds a
 * batch.  partials: unetk_metrics_num_partials(n) * 8 doubles of scratch. */
int unetk_metrics_num_partials(int64_t n);
int unetk_metrics_update(const float* pred, const float* target, int64_t n, double* partials, double* state,
                         unetk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNETK_H */
