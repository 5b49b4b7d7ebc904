/* frl_hip.h -- C ABI of libfrlhip.so: the MI355X (gfx950) VQ-VAE training hot path.
 *
 * The reference (nnnagle/vq-vae) is pure Python/PyTorch and has NO native plugin interface; the boundary these
 * entry points replace is the set of torch ops the reference's hot path executes.  Each
 * declaration cites the reference call site (file:line under /root/reference) whose forward and autograd backward
 * it stands in for.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure; frl_last_error() gives the thread-local text;
 *   - all pointers are BORROWED DEVICE pointers (owned by the caller, e.g. torch tensors); the library allocates
 *     nothing user-visible; scratch comes from a caller-provided workspace sized by *_workspace_bytes();
 *   - explicit hipStream_t (void*) argument, no global state: functions are re-entrant and graph-capturable;
 *   - activations are NHWC rows [P][C] (C contiguous) of `dtype` (FRL_F32 = 0 | FRL_BF16 = 1);
 *     parameters and their gradients are float32 in the reference's own layouts (OIHW / [Cout][Cin][k]);
 *   - act: 0 none, 1 relu, 2 sigmoid.  Backward entry points take the activation OUTPUT y to apply act'(y).
 */
#ifndef FRL_HIP_H
#define FRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* frl_stream_t; /* hipStream_t */

enum { FRL_F32 = 0, FRL_BF16 = 1, FRL_F16 = 2 /* raw chunk-store rows of frl_normalize_tiles only */ };
enum { FRL_ACT_NONE = 0, FRL_ACT_RELU = 1, FRL_ACT_SIGMOID = 2 };

/* ---- library ------------------------------------------------------------------------------------------------- */
int frl_version(void);
const char* frl_last_error(void);
int frl_device_arch(char* buf, int n); /* must report gfx950 */
/* Live per-kernel timing (bench.py): with the switch on, every kernel the library launches is bracketed by a HIP event pair recorded on
 * its launch stream; the report synchronises them and writes "kernel expression \t calls \t total ms \n" lines. */
int frl_kernel_timing_enable(int on);
int frl_kernel_timing_report(char* buf, int n);

/* ---- weight-image cache ------------------------------------------------------------------------------------------------------
 * Every conv-like call first rewrites its float32 weights into a packed MFMA fragment image (a small prologue launch per call).  A
 * caller that knows when the weights change (the trainer: once per optimizer step) keeps the images in an arena of its own:
 *   h = frl_pack_cache_create(arena, bytes)   arena: device memory owned by the caller, at least frl_pack_cache_table_bytes() + images
 *   frl_pack_cache_activate(h)                calls made from now on (any thread) find / register their image in cache h and skip the
 *                                             prologue on a hit; frl_pack_cache_activate(0) restores the default (pack per call)
 *   frl_pack_cache_refresh(h, stream)         rewrites EVERY registered image from the current weights with ONE launch
 *   frl_pack_cache_destroy(h)
 * An image is valid from its registration (or the last refresh) until its weights change: refresh right after the optimizer. */
size_t frl_pack_cache_table_bytes(void);
int frl_pack_cache_create(void* arena, size_t bytes);
int frl_pack_cache_destroy(int handle);
int frl_pack_cache_activate(int handle);      /* returns the previously active handle */
int frl_pack_cache_images(int handle);        /* number of registered images */
int frl_pack_cache_refresh(int handle, frl_stream_t stream);

/* ---- pointwise (1x1) convolution --------------------------------------------------------------------------------
 * nn.Conv2d(.,.,1): frl/models/conv2d_encoder.py:106-114; spatial.py:262-263; representation.py:169;
 * conditioning.py:55-67; decoder template heads.py:128-198.  w [Cout][Cin], bias [Cout] or NULL. */
/* every forward / bwd_data convolution entry point takes a workspace of frl_conv_workspace_bytes(Cin, Cout, taps) bytes
 * (taps = 1 pointwise, 3 temporal, 9 spatial; 6 covers a whole TCN block) for the packed MFMA weight image it builds. */
size_t frl_conv_workspace_bytes(int Cin, int Cout, int taps);
int frl_conv1x1_fwd(const void* x, const float* w, const float* bias, void* y, int64_t P, int Cin, int Cout, int act,
                    int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
int frl_conv1x1_bwd_data(const void* dy, const void* y, int act, const float* w, void* dx, int64_t P, int Cin, int Cout,
                         int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* Backward-pass accumulation in the epilogue (autograd's sum of the gradients of a tensor with two consumers, fused into the
 * kernel that produces one of them): dx = (dy .* act'(y)) W + add.  add [P][Cin] must not alias dx. */
int frl_conv1x1_bwd_data_add(const void* dy, const void* y, int act, const float* w, void* dx, const void* add, int64_t P,
                             int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* Tuning hook (no reference counterpart): upper bound of the workgroups, i.e. float32 partial slabs, of a 1x1 weight-gradient launch;
 * returns the previous bound.  frl_conv1x1_bwd_weight_workspace_bytes follows it: size workspaces after changing it. */
int frl_wgrad_set_max_workgroups(int n);
size_t frl_conv1x1_bwd_weight_workspace_bytes(int64_t P, int Cin, int Cout);
int frl_conv1x1_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, float* dbias, int64_t P,
                           int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* generalised weight gradient: rows are (b, t, hw); x is read at time t + toff (zero outside [0,T));
 * dw destination strides (dso, dsi) address one tap of a [Cout][Cin][ntap] tensor (nn.Conv1d, tcn.py:56-62).
 * flags bit0: scalar LDS fragment reads (debug), bit1: accumulate into dbias. */
int frl_conv_tap_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, int64_t dso, int64_t dsi,
                            float* dbias, int64_t P, int Cin, int Cout, int HW, int T, int toff, int dtype, void* ws,
                            size_t ws_bytes, int flags, frl_stream_t stream);

/* ---- 3x3 convolution, pad 1 ------------------------------------------------------------------------------------
 * nn.Conv2d(.,.,3,padding=1): frl/models/spatial.py:258-261 (mix_backbone), :266-272 (gate_net).
 * x [B][H][W][Cin], w [Cout][Cin][3][3]. */
int frl_conv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                    int act, int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* gate_net's second convolution with the blend in its epilogue (spatial.py:332-335 at min_gate = 0): gate = sigmoid(conv3x3(x) + bias),
 * out = smoothed + gate * residual.  One launch instead of the convolution + frl_gate_blend_fwd. */
int frl_conv3x3_fwd_gate_blend(const void* x, const float* w, const float* bias, const void* smoothed, const void* residual, void* gate,
                               void* out, int B, int H, int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes,
                               frl_stream_t stream);
int frl_conv3x3_bwd_data(const void* dy, const void* y, int act, const float* w, void* dx, int B, int H, int W, int Cin,
                         int Cout, int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* frl_conv3x3_bwd_data with epilogue extras: dx = conv(...) + add (add may be null); with the pair sub_from / out2 (both or neither)
 * also out2 = sub_from - dx.  EdgeAwareSmoothingConv2D's backward uses both: the residual's two gradient streams are summed here and
 * d_smoothed - d_residual leaves in the same launch (spatial.py:331-339: residual = x - smoothed feeds the gate net and the blend). */
int frl_conv3x3_bwd_data_fused(const void* dy, const void* y, int act, const float* w, void* dx, const void* add,
                               const void* sub_from, void* out2, int B, int H, int W, int Cin, int Cout, int dtype, void* ws,
                               size_t ws_bytes, frl_stream_t stream);
/* dx = conv^T(dy .* act'(y)) .* out_act'(out_y): out_y [B][H][W][Cin] is the output of the ReLU / sigmoid (out_act) that the NEXT backward
 * step differentiates through at the same pixels; that step's calls then take dx with act = FRL_ACT_NONE (no mask pass over their input:
 * 16-35 us per 3x3 backward-data call and 6-15 us per weight-gradient call at BASELINE configs[1]). */
int frl_conv3x3_bwd_data_outmask(const void* dy, const void* y, int act, const float* w, void* dx, const void* out_y, int out_act, int B,
                                 int H, int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
size_t frl_conv3x3_bwd_weight_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int frl_conv3x3_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, float* dbias, int B, int H,
                           int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, int flags, frl_stream_t stream);
/* Test / A-B hook (no reference counterpart): 1 routes every 3x3 weight gradient through the generic kernel instead of the bf16
 * band kernel; returns the previous setting. */
int frl_conv3x3_wgrad_force_generic(int on);
/* Test / A-B hook: 0 = conv3x3 forward / bwd-data on 16-row tiles only (default 1: bf16 images of >= 32 rows take 32 x 16 tiles); returns the previous setting. */
int frl_conv3x3_tile32(int on);

/* ---- GroupNorm over NHWC rows ----------------------------------------------------------------------------------
 * nn.GroupNorm(G, C), eps 1e-5, per-sample statistics: frl/models/conv2d_encoder.py:117 (ReLU fused when relu=1,
 * :119-121).  mean/rstd [B][G] f32 are saved for the backward. */
int frl_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B,
                      int HW, int C, int G, float eps, int relu, int dtype, frl_stream_t stream);
size_t frl_groupnorm_bwd_workspace_bytes(int B, int C, int G);
int frl_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                      const float* rstd, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int G, int relu,
                      int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);

/* ---- loss head (csrc/elementwise.hip) ----------------------------------------------------------------------------
 * out[0] = sum_i coef[i] * *terms[i] (device scalars, host coefficients, n <= 8), ok_out[0] (optional) = 1 if out[0] is finite else 0: the weighted
 * sum of the loss terms (scripts/train_vqvae.py:236-248) and the isfinite guard (step.py:1057-1074) in one launch; its backward. */
int frl_scalar_combine(const float* const* terms_host, const float* coef_host, int n, float* out, float* ok_out, frl_stream_t stream);
int frl_scalar_fanout(const float* g, const float* coef_host, int n, float* out, frl_stream_t stream);
/* the same with optional device multipliers (mult_host: n device pointers, entries or the array may be NULL): term i is weighted by
 * coef[i] * *mult[i] -- a loss weight on a per-step schedule (lambda_vq(step), scripts/train_vqvae.py:236-248,324) stays a device word
 * that a captured graph reads at replay time */
int frl_scalar_combine_dev(const float* const* terms_host, const float* coef_host, const float* const* mult_host, int n, float* out,
                           float* ok_out, frl_stream_t stream);
/* frl_scalar_combine_dev plus a second, un-scheduled combination of the same terms: aux_out[0] = sum_i aux_coef[i] * *terms[i] (a reported
 * sub-total such as vq_loss = L_codebook + beta L_commit of scripts/train_vqvae.py:236-248) in the same launch. */
int frl_scalar_combine_aux(const float* const* terms_host, const float* coef_host, const float* const* mult_host, const float* aux_coef_host,
                           int n, float* out, float* ok_out, float* aux_out, frl_stream_t stream);
int frl_scalar_fanout_dev(const float* g, const float* coef_host, const float* const* mult_host, int n, float* out, frl_stream_t stream);

/* ---- fused two-layer type encoder (csrc/enc_fused.hip) -----------------------------------------------------------
 * conv1x1 C0->C1 (no bias) -> GroupNorm(G1) -> ReLU -> conv1x1 C1->C2 (no bias) -> GroupNorm(G2): Conv2DEncoder with two layers
 * (frl/models/conv2d_encoder.py:100-159) as ONE launch per direction, one workgroup per sample; the 128-channel intermediates never
 * reach HBM in the forward.  frl_encoder2_supported: 1 for (64, 128, 64, 8, 8, HW % 16 == 0, FRL_BF16), else compose
 * frl_conv1x1_* / frl_groupnorm_*.  stats [B][32] f32 = mean1[8] rstd1[8] mean2[8] rstd2[8].
 * Backward (parameters only: the encoder's input is data): dw1 [C1][C0], dw2 [C2][C1] and the GroupNorm parameter gradients; the chain
 * is recomputed per pass and the weight gradients are contracted inside the kernel, so no intermediate tensor reaches HBM. */
int frl_encoder2_supported(int C0, int C1, int C2, int G1, int G2, int HW, int dtype);
size_t frl_encoder2_workspace_bytes(int B);
int frl_encoder2_fwd(const void* x, const float* w1, const float* g1, const float* b1, const float* w2, const float* g2, const float* b2,
                     void* z, float* stats, int B, int HW, float eps, void* ws, size_t ws_bytes, frl_stream_t stream);
int frl_encoder2_bwd(const void* x, const void* dz, const float* w1, const float* g1, const float* b1, const float* w2, const float* g2,
                     const float* b2, const float* stats, float* dw1, float* dg1, float* db1, float* dw2, float* dg2, float* db2,
                     int B, int HW, void* ws, size_t ws_bytes, frl_stream_t stream);

/* ---- EdgeAwareSmoothingConv2D fixed stencils -------------------------------------------------------------------
 * Sobel/4 depthwise gradients (frl/models/spatial.py:240-249,295-296): g [B][H][W][2C] = cat[dx, dy]. */
int frl_sobel_fwd(const void* x, void* g, int B, int H, int W, int C, int dtype, frl_stream_t stream);
int frl_sobel_bwd(const void* dg, void* dx, int B, int H, int W, int C, int dtype, frl_stream_t stream);
/* dx = sobel^T(dg) + dx_add (dx_add [B][H][W][C] or null): x also feeds the filter bank, whose dx arrives through dx_add */
int frl_sobel_bwd_add(const void* dg, void* dx, const void* dx_add, int B, int H, int W, int C, int dtype, frl_stream_t stream);
/* directional bank + rank-R mixing (spatial.py:224-237,300-331): a_logit [P][8R] (k*R+r), b_logit [P][C*R] (c*R+r);
 * outputs smoothed, residual = x - smoothed, and the softmaxed maps a_soft / b_soft saved for the backward. */
int frl_edge_smooth_stencil_fwd(const void* x, const void* a_logit, const void* b_logit, void* smoothed, void* residual,
                                void* a_soft, void* b_soft, int B, int H, int W, int C, int R, int coarse_dilation,
                                int dtype, frl_stream_t stream);
int frl_edge_smooth_stencil_bwd(const void* d_smoothed, const void* x, const void* a_soft, const void* b_soft, void* dx,
                                void* da_logit, void* db_logit, const void* dx_add /* optional: added to dx in the store */, int B, int H,
                                int W, int C, int R, int coarse_dilation, int dtype, frl_stream_t stream);
/* Fused mixing heads + softmaxes + directional bank for the hot configuration (bf16, C = 64, 64 hidden features, rank 4; replaces
 * mix_head_A / mix_head_B (nn.Conv2d 1x1, spatial.py:262-263), the two softmaxes (:300-307) and the bank + mix (:314-331) in one launch;
 * the [P,32] / [P,256] logits and their soft-maxed copies never reach memory).  feat = relu(mix_backbone(cat[dx,dy])) [B][H][W][64];
 * wa [32][64], ba [32], wb [256][64], bb [256] float32 with output channels k*R+r resp. c*R+r. */
int frl_smooth_heads_supported(int C, int hidden, int rank, int dtype);
size_t frl_smooth_heads_workspace_bytes(int64_t npix);
int frl_smooth_heads_fwd(const void* x, const void* feat, const float* wa, const float* ba, const float* wb, const float* bb,
                         void* smoothed, void* residual, int B, int H, int W, int coarse_dilation, void* ws, size_t ws_bytes,
                         frl_stream_t stream);
/* backward: d_smoothed already carries the residual path (d smoothed - d residual); dx_add (optional) is added to dx.  The heads are
 * recomputed from feat; outputs dx, dfeat [P][64] (bf16) and the four head parameter gradients (float32).  scratch: exchange tensors of
 * the two launches (u = ds * softmax B [P][256], softmax A [P][32], bf16), frl_smooth_heads_bwd_scratch_bytes(npix) bytes. */
size_t frl_smooth_heads_bwd_scratch_bytes(int64_t npix);
int frl_smooth_heads_force_gather(int on);     /* test hook: d x of the backward by the gather kernel instead of the LDS-tiled one; returns the previous setting */
int frl_smooth_heads_bwd(const void* d_smoothed, const void* x, const void* feat, const float* wa, const float* ba, const float* wb,
                         const float* bb, const void* dx_add, void* dx, void* dfeat, float* dwa, float* dba, float* dwb, float* dbb,
                         void* scratch, size_t scratch_bytes, int B, int H, int W, int coarse_dilation, void* ws, size_t ws_bytes,
                         frl_stream_t stream);
/* The same with dfeat_relu != 0: dfeat is returned multiplied by [feat > 0] (feat is the output of the ReLU convolution mix_backbone,
 * spatial.py:258-261), so that convolution's backward-data / backward-weight calls take it with act = FRL_ACT_NONE. */
int frl_smooth_heads_bwd_masked(const void* d_smoothed, const void* x, const void* feat, const float* wa, const float* ba, const float* wb,
                                const float* bb, const void* dx_add, void* dx, void* dfeat, float* dwa, float* dba, float* dwb, float* dbb,
                                void* scratch, size_t scratch_bytes, int B, int H, int W, int dil, int dfeat_relu, void* ws,
                                size_t ws_bytes, frl_stream_t stream);
/* ---- FiLM conditioning of the phase path, fused (bf16, 64 conditioning channels, hidden 32, 12 target channels) ----------------------
 * Replaces FiLMLayer.forward (frl/models/conditioning.py:82-102: two conv1x1 -> ReLU -> conv1x1 nets) and the modulation
 * gamma * h + beta broadcast over T (frl/models/representation.py:369-372) with one launch per direction.  z_type [B][HW][64] is a
 * stop-gradient input (representation.py:350-351); h, z, dz, dh [B][T][HW][12]; gamma, beta [B][HW][12]; w1* [32][64], w2* [12][32]. */
int frl_film_fused_supported(int cond_dim, int hidden, int target_dim, int dtype);
size_t frl_film_fused_workspace_bytes(void);
int frl_film_fused_fwd(const void* z_type, const void* h, const float* w1g, const float* b1g, const float* w2g, const float* b2g,
                       const float* w1b, const float* b1b, const float* w2b, const float* b2b, void* z, void* gamma, void* beta, int B,
                       int T, int HW, void* ws, size_t ws_bytes, frl_stream_t stream);
int frl_film_fused_bwd(const void* z_type, const void* h, const void* dz, const float* w1g, const float* b1g, const float* w2g,
                       const float* b2g, const float* w1b, const float* b1b, const float* w2b, const float* b2b, void* dh, float* dw1g,
                       float* db1g, float* dw2g, float* db2g, float* dw1b, float* db1b, float* dw2b, float* db2b, int B, int T, int HW,
                       void* ws, size_t ws_bytes, frl_stream_t stream);
/* out = smoothed + max(gate_raw, min_gate) * residual (spatial.py:333-335); n = element count */
int frl_gate_blend_fwd(const void* smoothed, const void* residual, const void* gate_raw, float min_gate, void* out,
                       void* gate_out, int64_t n, int dtype, frl_stream_t stream);
int frl_gate_blend_bwd(const void* dout, const void* dgate_ext, const void* residual, const void* gate_raw, float min_gate,
                       void* d_residual, void* d_gate_raw, int64_t n, int dtype, frl_stream_t stream);
/* The same with sigmoid_mask != 0: d_gate_raw is returned multiplied by gate_raw (1 - gate_raw), the derivative of the sigmoid that produced
 * gate_raw (gate_net, spatial.py:266-272): the convolution's backward calls take it with act = FRL_ACT_NONE. */
int frl_gate_blend_bwd_masked(const void* dout, const void* dgate_ext, const void* residual, const void* gate_raw, float min_gate,
                              void* d_residual, void* d_gate_raw, int64_t n, int dtype, int sigmoid_mask, frl_stream_t stream);

/* ---- fused TCN GatedResidualBlock ------------------------------------------------------------------------------
 * frl/models/tcn.py:78-111 on x [B][T][HW][Cin] (npix = B*HW); conv_w [Cout][Cin][3], gate_w [Cout][Cout],
 * proj_w [Cout][Cin] or NULL (identity residual, tcn.py:71-76). */
int frl_tcn_block_fwd(const void* x, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b,
                      const float* gate_w, const float* gate_b, const float* proj_w, const float* proj_b, void* y,
                      int64_t npix, int HW, int T, int Cin, int Cout, int dilation, int G, float eps, int dtype, void* ws,
                      size_t ws_bytes, frl_stream_t stream);
size_t frl_tcn_block_bwd_workspace_bytes(int64_t npix, int Cout);
int frl_tcn_block_bwd(const void* x, const void* dy, const float* conv_w, const float* conv_b, const float* gn_w,
                      const float* gn_b, const float* gate_w, const float* gate_b, const float* proj_w,
                      const float* proj_b, void* dconv, void* dgpre, void* normed, void* dres, float* dgamma, float* dbeta,
                      int64_t npix, int HW, int T, int Cin, int Cout, int dilation, int G, float eps, int dtype, void* ws,
                      size_t ws_bytes, frl_stream_t stream);
int frl_tcn_block_bwd_data(const void* dconv, const void* dres, const float* conv_w, const float* proj_w, void* dx,
                           int64_t npix, int HW, int T, int Cin, int Cout, int dilation, int dtype, void* ws, size_t ws_bytes,
                           frl_stream_t stream);

/* fully fused backward (bf16, Cin = Cout = 64, T <= 5, identity residual): one launch reads x, dy and writes dx and every
 * parameter gradient of the block; weight gradients are contracted inside the kernel (no HBM side outputs). */
int frl_tcn_block_bwd_fused_supported(int T, int Cin, int Cout, int G, int has_proj, int dtype);
size_t frl_tcn_block_bwd_fused_workspace_bytes(int64_t npix);
int frl_tcn_block_bwd_fused(const void* x, const void* dy, const float* conv_w, const float* conv_b, const float* gn_w,
                            const float* gn_b, const float* gate_w, const float* gate_b, void* dx, float* d_conv_w,
                            float* d_conv_b, float* d_gn_w, float* d_gn_b, float* d_gate_w, float* d_gate_b, int64_t npix,
                            int HW, int T, int dilation, int G, float eps, void* ws, size_t ws_bytes, frl_stream_t stream);

/* ---- deferred weight-gradient reductions (csrc/defer.hip) -------------------------------------------------------------------------
 * Every weight-gradient kernel leaves per-workgroup float32 slabs that a small fixed-order reduction turns into the gradient tensors.
 * Between frl_defer_begin() and frl_defer_flush(stream) those reductions (of frl_tcn_hot_bwd, frl_conv3x3_bwd_weight,
 * frl_decoder_mse_bwd, frl_encoder2_bwd, frl_film_fused_bwd, frl_smooth_heads_bwd, frl_conv1x1_bwd_weight and frl_vq_bwd without per-code
 * sums -- whose epilogue reads `counts`, the codebook and `gscale` when it runs: those three stay alive and unchanged until the flush;
 * calls that cut their output channels into several slices are not deferred) are parked and the flush runs them all in ONE launch: same summation order, bit-identical gradients, ~12 launches fewer per
 * train step.  The caller owns two promises: the workspaces handed to the deferred calls stay untouched until the flush (hand every
 * call its own), and nothing reads the gradient tensors before it.  frl_defer_destinations lists the gradient pointers of the parked
 * jobs so that the caller can check they still are the tensors its optimizer reads.  Process-wide state (any thread's calls are
 * parked); at most 24 jobs, further ones run undeferred.  frl_defer_flush returns the number of jobs run. */
int frl_defer_begin(void);
int frl_defer_pending(void);
int frl_defer_destinations(void** out, int max);
int frl_defer_flush(frl_stream_t stream);
int frl_defer_abort(void);

/* hot-configuration block kernels (bf16, Cin = Cout = 64, T = 5, G = 8, identity residual, dilation 1/2/4: the three phase-path
 * blocks of configs/vae_v0.yaml): (T, dilation) are compile-time, the pixel's time series stays in registers, the temporal
 * conv is evaluated once per tile.  frl_tcn_hot_bwd is one launch for dx + every parameter gradient (tcn.py:78-111). */
int frl_tcn_hot_supported(int T, int Cin, int Cout, int G, int dilation, int has_proj, int dtype);
size_t frl_tcn_hot_fwd_workspace_bytes(void);
size_t frl_tcn_hot_bwd_workspace_bytes(int64_t npix);
/* drop_mask: NULL, or the training-mode Dropout1d mask of the block (tcn.py:53) as [B][HW][64] bf16 holding 0 or 1/(1-p) per
 * (pixel series, channel): the temporal conv sees x .* mask, the residual path the untouched x (tcn.py:89-90). */
int frl_tcn_hot_fwd(const void* x, const void* drop_mask, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b,
                    const float* gate_w, const float* gate_b, void* y, int64_t npix, int HW, int dilation, float eps, void* ws,
                    size_t ws_bytes, frl_stream_t stream);
/* 1: frl_tcn_hot_bwd takes dx = NULL for this shape (block input without gradient: no conv^T GEMM, no dx store; drop_mask must be NULL) */
int frl_tcn_hot_bwd_nodx_supported(int64_t npix, int HW);
int frl_tcn_hot_bwd(const void* x, const void* drop_mask, const void* dy, const float* conv_w, const float* conv_b, const float* gn_w,
                    const float* gn_b, const float* gate_w, const float* gate_b, void* dx, float* d_conv_w, float* d_conv_b,
                    float* d_gn_w, float* d_gn_b, float* d_gate_w, float* d_gate_b, int64_t npix, int HW, int dilation,
                    float eps, void* ws, size_t ws_bytes, frl_stream_t stream);
/* The dense phase chain forward in ONE launch (hot configuration, three blocks with dilation 1, 2, 4: TCNEncoder.forward, tcn.py:242-290)
 * followed by the 1x1 phase head (representation.py:169,362-366): x [B][5][HW][64] -> y1, y2, y3 (the blocks' outputs, kept for the
 * backward) and h [B][5][HW][Ch] = head_w y3 + head_b, Ch in {4, 8, 12, 16}.  conv_w .. gate_b: arrays of three pointers (block 0, 1, 2). */
size_t frl_tcn_chain_fwd_workspace_bytes(void);
int frl_tcn_chain_static_tiles(int on);   /* A/B hook: 1 = fixed tile sequence per wave, 0 (default) = tiles handed out by an LDS counter; returns the previous setting */
int frl_tcn_chain_fwd(const void* x, const float* const* conv_w, const float* const* conv_b, const float* const* gn_w,
                      const float* const* gn_b, const float* const* gate_w, const float* const* gate_b, const float* head_w,
                      const float* head_b, void* y1, void* y2, void* y3, void* h, int64_t npix, int HW, int Ch, float eps, void* ws,
                      size_t ws_bytes, frl_stream_t stream);
/* The last block of the phase encoder together with the backward-data of the 1x1 phase head behind it (representation.py:169): `dh`
 * [B][5][HW][Ch] bf16 is the head's output gradient, head_w [Ch][64] float32, Ch in {4, 8, 12, 16}; dy = dh head_w is formed inside the
 * kernel on the matrix cores (float32, never written).  Same outputs as frl_tcn_hot_bwd; needs the two-subgroup kernel: no mask,
 * HW % 64 == 0, dilation 4, dx != NULL.  Workspace: frl_tcn_hot_bwd_head_workspace_bytes(npix). */
int frl_tcn_hot_bwd_head_supported(int64_t npix, int HW, int Ch);
size_t frl_tcn_hot_bwd_head_workspace_bytes(int64_t npix);
int frl_tcn_hot_bwd_head(const void* x, const void* dh, const float* head_w, int Ch, const float* conv_w, const float* conv_b,
                         const float* gn_w, const float* gn_b, const float* gate_w, const float* gate_b, void* dx, float* d_conv_w,
                         float* d_conv_b, float* d_gn_w, float* d_gn_b, float* d_gate_w, float* d_gate_b, int64_t npix, int HW,
                         int dilation, float eps, void* ws, size_t ws_bytes, frl_stream_t stream);
/* Two kernels stand behind frl_tcn_hot_bwd: tcn_hot_bwd3 (no mask, HW a multiple of 64: x of a 64-pixel tile is staged once in LDS by
 * LDS-DMA, next tile prefetched) and the 8-wave kernel that also takes a mask and ragged pixel counts.  Test hook: on != 0 routes
 * every call through the latter so that the two can be compared on the same inputs. */
int frl_tcn_hot_force_generic_tiles(int on);   /* returns the previous setting */
/* Which kernel serves the unmasked, HW % 64 == 0 backward: 4 (default) = tcn_hot_bwd4_kernel (two independent 4-wave subgroups per
 * workgroup over 32-pixel tiles), 3 = tcn_hot_bwd3_kernel (8 waves in lockstep over 64-pixel tiles).  Same results up to the rounding
 * of dres (bf16); returns the previous setting.  For A/B measurements and the parity tests of both kernels. */
int frl_tcn_hot_bwd_variant(int v);
/* Tuning hook of tcn_hot_bwd4: subgroup 0's share (in 32nds, 1..31) of a workgroup's tiles for variant 0 (with dx), 1 (without dx), 2 (head);
 * returns the previous value (share outside 1..31: query only).  Any value gives the same gradients up to float32 summation order. */
int frl_tcn_hot_bwd4_share(int variant, int share);

/* ---- optimizer step (frl/training/representation/step.py:1081-1087: clip_grad_norm_(1.0) then AdamW.step()) ---------------
 * Two launches for the whole parameter set.  desc: HOST table of ntensors records {float* p; const float* g; float* m; float* v;
 * int64_t n; float weight_decay; int lag} (48 bytes; lag = updates the tensor skipped) -- it travels in the kernel-argument
 * segment, so moved gradient buffers cost no copy or sync; chunks: DEVICE table of nchunks int32 pairs {tensor index,
 * 4096-element window} sorted by tensor, chunk_tensor: its tensor column on the HOST.  step is the 1-based update count (bias
 * corrections in float64 as torch.optim.AdamW); max_norm <= 0 disables clipping; norm_out (device float, may be NULL) receives
 * the pre-clip global gradient norm.  ok (device float, may be NULL): the reference's isfinite guard evaluated ON THE DEVICE --
 * ok[0] <= 0 leaves parameters and moments untouched; counters (device int[2], may be NULL) = {updates applied, updates skipped},
 * and when given, counters[0] + 1 replaces `step` as the update number (exact under skipped batches, no host sync).
 * lr_dev (device float, may be NULL): when given, lr_dev[0] replaces `lr` -- the learning rate of a train step captured in a
 * hipGraph is written to that word before every replay (per-batch scheduler.step() of loops.py:110 without re-capturing). */
size_t frl_adamw_workspace_bytes(void);
int frl_adamw_clip_step(const void* desc_host, int ntensors, const void* chunks, const int* chunk_tensor, int nchunks, float max_norm,
                        float lr, double beta1, double beta2, float eps, int step, float* norm_out, const float* ok, int* counters,
                        const float* lr_dev, void* ws, size_t ws_bytes, frl_stream_t stream);
/* dst[i] = scale * src[i] over a HOST table of {const float* src; float* dst; int64_t n} records (24 bytes; src NULL -> zeros):
 * flattens the scattered gradients of a bucket for the data-parallel all-reduce in one launch. */
int frl_multi_tensor_scale_copy(const void* desc_host, int ntensors, const void* chunks, const int* chunk_tensor, int nchunks, float scale,
                                frl_stream_t stream);

/* ---- FiLM modulation, time mean, add ---------------------------------------------------------------------------
 * z = gamma * h + beta broadcast over T (frl/models/representation.py:369-372); h [B][T][HW][C], gamma [B][HW][C]. */
int frl_film_modulate_fwd(const void* h, const void* gamma, const void* beta, void* out, int64_t B, int T, int64_t HW,
                          int C, int dtype, frl_stream_t stream);
int frl_film_modulate_bwd(const void* dout, const void* h, const void* gamma, void* dh, void* dgamma, void* dbeta,
                          int64_t B, int T, int64_t HW, int C, int dtype, frl_stream_t stream);
int frl_mean_time_fwd(const void* tile, void* out, int64_t B, int T, int64_t HWC, int dtype, frl_stream_t stream);
int frl_add(const void* a, const void* b, float scale_b, void* out, int64_t n, int dtype, frl_stream_t stream); /* out = a + scale_b*b */

/* ---- tile ingest: chunk-store rows -> normalised training rows -------------------------------
 * Replaces the per-channel numpy normalisation + masking of FeatureBuilder on the host
 * (frl/data/loaders/builders/feature_builder.py:402-462 _apply_normalization, :487-548 _normalize_array, :709-737
 * _apply_mask_to_data; presets of frl/data/normalization/normalization.py:116-254) by one pass on the device over the
 * (time,y,x,feature) rows as stored (utils/data_stack.py:271-309).  Per feature f:
 *     r = (x - sub) / div;  if (flags & 1) r = r * mul + add;  if (flags & 2) r = max(r, lo);  if (flags & 4) r = min(r, hi)
 * (float32, that operation order, no fused multiply-add).  A row is valid iff valid[row] != 0 (when given) and all its F raw
 * values are finite; invalid rows are written as zeros and mask_out[row] = 0.
 * raw [rows][F] FRL_F32 | FRL_F16, table [F] device records, out [rows][F] FRL_F32 | FRL_BF16, F in {8,16,...,512}. */
typedef struct FrlNormRec {
  float sub, div, mul, add, lo, hi;
  int flags; /* 1 rescale, 2 clamp below, 4 clamp above */
  int pad;
} FrlNormRec;
int frl_normalize_tiles(const void* raw, int raw_dtype, const uint8_t* valid, const void* table, void* out, int out_dtype,
                        uint8_t* mask_out, int64_t rows, int F, frl_stream_t stream);
/* Same arithmetic with the tile cut done on the device: `chunk` is one whole stored chunk [T][CY][CX][F] (uploaded with ONE
 * contiguous copy; batches are chunk-locked, utils/samplers.py:42-108), desc [ntiles] = int32 {y0, x0, h, w} per tile (device).
 * Replaces the window read + zero padding of partial patches on the host (forest_dataset_v2.py:328-369): rows/columns beyond
 * h/w are written as zeros with mask 0.  out [ntiles][T][tile][tile][F], mask_out [ntiles][T][tile][tile]. */
int frl_normalize_chunk_tiles(const void* chunk, int raw_dtype, int T, int CY, int CX, int F, const int32_t* desc, int ntiles,
                              int tile, const void* table, void* out, int out_dtype, uint8_t* mask_out, frl_stream_t stream);

/* Host-side helper of the tile loader: multi-threaded memcpy of one stored chunk into the pinned upload buffer (no GPU work). */
int frl_host_parallel_copy(void* dst, const void* src, size_t nbytes, int nthreads);

/* ---- masked L2 reconstruction loss -----------------------------------------------------------------------------
 * frl/losses/reconstruction.py:95-139 (loss_type "l2", reduction "mean"); out = {mean, n_valid_elements}. */
size_t frl_mse_workspace_bytes(void);
int frl_mse_fwd(const void* pred, const void* target, const uint8_t* mask, int64_t P, int C, float* out, int dtype,
                void* ws, size_t ws_bytes, frl_stream_t stream);
int frl_mse_bwd(const void* pred, const void* target, const uint8_t* mask, const float* gscale, const float* stats,
                int64_t P, int C, void* dpred, int dtype, frl_stream_t stream);

/* fused decoder + loss (bf16, hidden 128, 64 features, latent <= 64 ch): xhat = W2 relu(W1 z + b1) + b2, L = mean_valid (xhat - x)^2.
 * Replaces the conv1x1 -> ReLU -> conv1x1 -> reconstruction_loss chain (heads.py:128-198 template, reconstruction.py:95-139) and its
 * backward with two launches; the hidden tensor and xhat stay on chip (xhat is written only when a buffer is passed). */
int frl_decoder_mse_fused_supported(int Cz, int hidden, int F, int dtype);
size_t frl_decoder_mse_workspace_bytes(int64_t P, int Cz);
int frl_decoder_mse_fwd(const void* z, const float* w1, const float* b1, const float* w2, const float* b2, const void* target,
                        const uint8_t* mask, void* xhat, float* out, int64_t P, int Cz, void* ws, size_t ws_bytes,
                        frl_stream_t stream);
int frl_decoder_mse_bwd(const void* z, const float* w1, const float* b1, const float* w2, const float* b2, const void* target,
                        const uint8_t* mask, const float* gscale, const float* stats, void* dz, float* dw1, float* db1,
                        float* dw2, float* db2, int64_t P, int Cz, void* ws, size_t ws_bytes, frl_stream_t stream);
int frl_decoder_mse_bwd_subgroups(int on);   /* A/B hook: 1 (default) = two independent 4-wave subgroups per workgroup, 0 = lockstep workgroups; returns the previous setting */

/* ---- vector quantizer ------------------------------------------------------------------------------------------
 * Not in the reference tree; constants frl/config/frl_model_v0.yaml:29-35,
 * scripts/train_vqvae.py:410-436.  idx bit-exact vs float64 argmin (first index on ties). */
size_t frl_vq_workspace_bytes(int64_t N, int K, int d);
int frl_vq_assign_fwd(const void* z, const float* E, int64_t N, int K, int d, int32_t* idx_out, void* zq_out,
                      float* stats_out, int32_t* counts_out, int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* Prepared codebook (||e||^2 + the packed MFMA fragment image of -2 e): it depends on the codebook content and dtype only, so a
 * caller that knows when the codebook changes (once per optimizer step; never during inference) builds it once and the assignment
 * itself is ONE kernel launch with nothing to zero beforehand: arg-min on the matrix cores, exact float64 re-evaluation of near-ties
 * while the codebook is still in LDS, z_q, squared error, code histogram (integer atomics) and the statistics (folded by the workgroup
 * that arrives last) -- when the codebook fits one LDS chunk (K * d_pad * sizeof(dtype) <= 64 KB), else as frl_vq_assign_fwd.
 * The image also holds the kernel's arrival counter and histogram accumulator (zeroed by frl_vq_prepare and left zeroed by every
 * call), so at most ONE assignment may be in flight per prepared image. */
size_t frl_vq_prepared_bytes(int K, int d);
/* A/B hook for the resident-codebook assignment of bf16 rows with d = 64 whose row count is a whole number of batches: nt = 1 / 2 / 4
 * selects the streaming kernel (one 16-wave workgroup per CU, 16 * nt * 16 rows per batch, no workgroup barrier in the batch loop),
 * 0 the kernel of round 2, -1 the default (environment variable FRL_VQ_STREAM, else the built-in choice).  Returns the previous
 * setting; outputs are the same bit for bit (the squared-error sum up to float32 summation order). */
int frl_vq_stream_tiles(int nt);
int frl_vq_prepare(const float* E, int64_t N, int K, int d, int dtype, void* prep, size_t prep_bytes, frl_stream_t stream);
int frl_vq_assign_fwd_prepared(const void* z, const float* E, void* prep /* NULL: prepare inside the call */, int64_t N, int K, int d,
                               int32_t* idx_out, void* zq_out, float* stats_out, int32_t* counts_out, int dtype, void* ws,
                               size_t ws_bytes, frl_stream_t stream);
int frl_vq_bwd(const void* g_out, const void* z, const void* zq /* optional */, const float* E, const int32_t* idx, const int32_t* counts,
               const float* gscale, float beta, int64_t N, int K, int d, void* g_z_out, float* g_E_out, float* sums_out,
               int dtype, void* ws, size_t ws_bytes, frl_stream_t stream);
/* ok: NULL, or a device float -- ok[0] <= 0 (or NaN) skips the update on the device (the train step's isfinite guard, evaluated
 * without a host synchronisation; frl/training/representation/step.py:1057-1074 "skip the batch"). */
int frl_vq_ema_update(const float* sums, const int32_t* counts, int K, int d, float decay, float eps, float* ema_count,
                      float* ema_sum, float* E, const float* ok, frl_stream_t stream);
/* Dead-code revival of the legacy trainer's CodebookManager (scripts/train_vqvae.py:92,196-198 constructs and attaches it; the
 * module is not in the reference tree -- build definition): codes with window_counts[k] < min_count are re-seeded with the
 * encoder row z[splitmix64(seed + k) mod N], their AdamW moment rows m / v (optional) are cleared, *revived += number of codes. */
int frl_vq_revive_dead_codes(float* E, const int64_t* window_counts, int64_t min_count, const void* z, int64_t N, int K, int d,
                             uint64_t seed, float* m, float* v, int32_t* revived, int dtype, frl_stream_t stream);

/* ---- sparse-location gather + InfoNCE over mined pairs (csrc/contrastive.hip) -----------------------------
 * frl_gather_locations_fwd: extract_at_locations of frl/utils/spatial.py:132-173 -- out[n][c] = feat[c*sC + row_n*sH + col_n*sW] for
 * coords [N][2] int64 (row, col; negative counts from the end), any element strides (NHWC rows and the reference's [C,H,W] view alike).
 * frl_segment_sum_rows: out[key][:] (+)= sum over each run of equal keys_sorted of vals[order[i]][:], rows added in list order -- the
 * bit-reproducible scatter-add behind both backward passes (order NULL = identity).
 * frl_infonce_fwd / frl_infonce_pair_grads: contrastive_loss of frl/losses/contrastive.py:29-212 over T pairs SORTED by anchor
 * (pairs [T][2] int64 rows of emb [.][D] f32, weights [T] or NULL, is_pos [T] bytes, seg [nseg+1] segment bounds; similarity 0 = l2
 * (-|a-b|^2/D), 1 = cosine, 2 = dot): per-anchor loss_a = -log(sum_pos + 1e-8) + log(sum_all + 1e-8) relative to the anchor's largest
 * logit, loss[0] = mean; coef[p] = d loss_a / d logit_p; the gradient rows ga / gb [T][D] carry gscale[0] / (nseg * temperature). */
int frl_gather_locations_fwd(const void* feat, int64_t sC, int64_t sH, int64_t sW, int C, int H, int W, const int64_t* coords, int64_t N,
                             void* out, int dtype, frl_stream_t stream);
int frl_segment_sum_rows(const float* vals, const int64_t* order, const int64_t* keys_sorted, int64_t M, int D, float* out, int64_t out_stride,
                         int accumulate, frl_stream_t stream);
int frl_infonce_fwd(const float* emb, int D, const int64_t* pairs, const float* weights, const unsigned char* is_pos, int64_t T,
                    const int64_t* seg, int64_t nseg, float temperature, int similarity, float* sims, float* logits, float* loss_a, float* coef,
                    float* loss, frl_stream_t stream);
int frl_infonce_pair_grads(const float* emb, int D, const int64_t* pairs, const float* sims, const float* coef, const float* gscale, int64_t T,
                           int64_t nseg, float temperature, int similarity, float* ga, float* gb, frl_stream_t stream);

/* ---- mutual k-nearest-neighbour pair mining -----------------------------------------------------
 * frl/losses/pairs.py:531-610 pairs_mutual_knn_chunked: per anchor the k nearest anchors by L2 distance in feature space, excluding
 * itself and same-patch anchors closer than pos_min_spatial pixels; knn_idx [N][k] in ascending distance (ties: lower index),
 * -1 where fewer than k candidates remain; mutual[i][r] = 1 iff i is also among the neighbours of knn_idx[i][r].
 * feat [N][D] float32 with D in {16, 32, 48, 64, 96, 128, 256} (pad narrower features with zero columns), patch_id [N] int32,
 * coords [N][2] float32 (row, col).  N is bounded by frl_mutual_knn_max_points(D) (four distance rows share the LDS). */
size_t frl_mutual_knn_max_points(int D);
int frl_mutual_knn(const float* feat, int N, int D, const int32_t* patch_id, const float* coords, float pos_min_spatial, int k,
                   int32_t* knn_idx, uint8_t* mutual, frl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FRL_HIP_H */
