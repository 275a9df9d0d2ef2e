/* littlegan_hip — C ABI of the MI355X (gfx950) kernels for the LittleGAN training-step hot path.
 *
 * Drop-in boundary: the reference has no FFI; its hot path sits behind Python objects
 * (model.py Generator/Discriminator/Adjuster, eager_trainer.py EagerTrainer) whose arithmetic TensorFlow
 * executes.  Each entry point below replaces the TF op(s) named in its comment (file:line relative to
 * /root/reference).  Conventions:
 *   - every pointer is a DEVICE pointer (fp32 unless stated), tensors are NHWC, row-major, dense;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work, never allocate or synchronise
 *     (graph-capturable); scratch comes from caller-provided workspaces sized by *_workspace_bytes();
 *   - return 0 on success, <0 on error (LG_ERR_*), message via lg_last_error(); no exceptions, no global
 *     state besides the thread-local error string;
 *   - `dtype` selects the MFMA operand type of the contraction: LG_DT_F32 = exact f32
 *     (v_mfma_f32_32x32x2_f32), LG_DT_BF16 = bf16 operands / f32 accumulate (v_mfma_f32_32x32x16_bf16).
 *     Storage in HBM: weights, their gradients, statistics records, images and image gradients are fp32 in both modes.
 *     Activation-sized tensors are fp32 with LG_DT_F32; with LG_DT_BF16 the entry points that take `*16` pointers
 *     (x16 / dy16 / z16 / out16 ...: bf16 tensors of the same shape) read and write them as bf16 ONLY — raw conv outputs,
 *     activated maps and inter-layer gradients then never exist in fp32 (the "bf16 activation path").
 *   - the thread-local strings of lg_last_error / lg_last_kernel are the only global state;
 * Stride-2 layers are described by (cb, cs, Hs, Ws): cb = channels of the BIG (2Hs x 2Ws) tensor,
 * cs = channels of the SMALL (Hs x Ws) tensor.  tf Conv2D kernels (HWIO, in=cb, out=cs) and
 * Conv2DTranspose kernels (HWOI, out=cb, in=cs) then share ONE memory layout [5][5][cb][cs].
 */
#ifndef LITTLEGAN_HIP_H
#define LITTLEGAN_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_ABI_VERSION 1
#define LG_OK 0
#define LG_ERR_ARG (-1)
#define LG_ERR_LAUNCH (-2)
#define LG_ERR_UNSUPPORTED (-3)
#define LG_DT_F32 0
#define LG_DT_BF16 1

int lg_abi_version(void);
const char* lg_last_error(void);
/* Name of the kernel template the calling thread's last conv / weight-gradient entry point launched ("" if none yet): a
 * static string such as "conv_up3_kernel<64,32>".  Measurement aid (bench.py's per-kernel roofline); no reference counterpart. */
const char* lg_last_kernel(void);
/* Resets that name to "": the name is sticky, so a timing harness clears it before a call whose launch path may name no kernel. */
int lg_clear_kernel(void);

/* ---- weight packing (once per layer per step; weights change every step) ------------------------
 * master kernel w[5][5][cb][cs] -> MFMA B-operand images in `dtype` (down pack + up pack). */
size_t lg_conv_pack_bytes(int cb, int cs, int dtype);
int lg_conv_pack(const float* w, void* pack, int cb, int cs, int dtype, void* stream);

/* ---- tf.compat.v1.layers.Conv2D(f,5,2,"same")  model.py:15 (Encoder) ----------------------------- */
/* y[B,Hs,Ws,cs] = conv(x[B,2Hs,2Ws,cb]) + bias ; cb == 3 uses the 3-channel patch kernel.
 * Results are deterministic for a given (B, shape).  dtype f32 on 8 x 8 output maps: while the launch has at most 256 blocks (B <= 84 at
 * cs = 384) the contraction is split in two halves that meet in y, so y differs from the same images inside a larger
 * launch by the order of one fp32 addition per element (1.3e-6 rms over the 6400-term contraction); every other shape's result does not
 * depend on B. */
int lg_conv2d_s2_fwd(const float* x, const void* pack, const float* bias, float* y, int B, int Hs, int Ws, int cb,
                     int cs, int dtype, void* stream);
/* same + fused InstanceNormalization moment partials of y (instance.py:114-115): when the chosen kernel supports it,
 * *nparts > 0 and spart holds [B][*nparts][3] doubles {count, mean, M2}; finish with lg_instnorm_stats_finalize.
 * *nparts == 0: not produced, use lg_instnorm_leaky_stats. */
/* spart size that is always enough for a layer: mode 0 = conv (Hm x Wm = OUTPUT map), 1 = transposed conv (INPUT map) */
size_t lg_conv_stats_workspace_bytes(int mode, int B, int Hm, int Wm, int N);
/* 1 if the *_fwd_stats call of this shape (up: 0 = lg_conv2d_s2_fwd_stats, 1 = lg_convT_s2_fwd_stats) will return fused
 * moment partials; Hs x Ws = the small map of the layer */
int lg_conv_fwd_stats_fused(int up, int dtype, int B, int Hs, int Ws, int cb, int cs);
/* y16 (may be null; bf16 dtype only): write the result z as bf16 there INSTEAD of fp32 to y (y may then be null) - the
 * bf16 activation path keeps the raw conv outputs in HBM as bf16; the moments come from the fp32 accumulators. */
int lg_conv2d_s2_fwd_stats(const float* x, const void* x16, const void* pack, const float* bias, float* y, void* y16, int B,
                           int Hs, int Ws, int cb, int cs, int dtype, void* spart, size_t spart_bytes, int* nparts,
                           void* stream);
/* lg_conv2d_s2_fwd_stats fed with the RAW bf16 conv output z16 [B,2Hs,2Ws,cb] of the level below and its statistics records
 * (zstats [B][8]): InstanceNormalization + LeakyReLU(alpha) (/root/reference/instance.py:105-128, model.py:22-24) are applied
 * while the operand is staged — bit-identical to the stand-alone apply pass, whose tensor is never written.  For forward passes
 * whose normalised maps have no other reader (/root/reference/eager_trainer.py:158-160: D on the Adjuster's output).  Result
 * bf16 in y16, moment partials always fused.  LG_ERR_UNSUPPORTED unless ..._zn_supported. */
int lg_conv2d_s2_fwd_stats_zn_supported(int B, int Hs, int Ws, int cb, int cs, int dtype);
int lg_conv2d_s2_fwd_stats_zn(const void* z16, const float* zstats, float alpha, const void* pack, const float* bias, void* y16,
                              int B, int Hs, int Ws, int cb, int cs, int dtype, void* spart, size_t spart_bytes, int* nparts,
                              void* stream);
/* *_m16: the activation operands may additionally be given as bf16 mirrors (x16 / dy16, same layout, may be null);
 * the bf16 MFMA kernels then read those instead of re-reading and re-rounding the fp32 tensors (bit-identical result) */
/* dx16 (may be null): write the data gradient as bf16 there INSTEAD of fp32 to dx (dx may then be null) */
int lg_conv2d_s2_dgrad_m16(const float* dy, const void* dy16, const void* pack, float* dx, void* dx16, int B, int Hs, int Ws,
                           int cb, int cs, int dtype, void* stream);
int lg_conv2d_s2_wgrad_m16(const float* x, const void* x16, const float* dy, const void* dy16, float* dw, void* workspace,
                           size_t ws_bytes, int B, int Hs, int Ws, int cb, int cs, int accumulate, int dtype, void* stream);
/* dx[B,2Hs,2Ws,cb] = conv2d_backprop_input(dy[B,Hs,Ws,cs]) */
int lg_conv2d_s2_dgrad(const float* dy, const void* pack, float* dx, int B, int Hs, int Ws, int cb, int cs, int dtype,
                       void* stream);
/* dw[5][5][cb][cs] (+)= conv2d_backprop_filter(x, dy) */
int lg_conv2d_s2_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t ws_bytes, int B, int Hs,
                       int Ws, int cb, int cs, int accumulate, int dtype, void* stream);

/* ---- tf.compat.v1.layers.Conv2DTranspose(f,5,(2,2),"same")  model.py:39-40 (Decoder) ------------- */
/* y[B,2Hs,2Ws,cb] = convT(x[B,Hs,Ws,cs]) + bias   (4-phase sub-pixel implicit GEMM) */
int lg_convT_s2_fwd(const float* x, const void* pack, const float* bias, float* y, int B, int Hs, int Ws, int cb,
                    int cs, int dtype, void* stream);
int lg_convT_s2_fwd_stats(const float* x, const void* x16, const void* pack, const float* bias, float* y, void* y16, int B,
                          int Hs, int Ws, int cb, int cs, int dtype, void* spart, size_t spart_bytes, int* nparts,
                          void* stream);
int lg_convT_s2_dgrad_m16(const float* dy, const void* dy16, const void* pack, float* dx, void* dx16, int B, int Hs, int Ws,
                          int cb, int cs, int dtype, void* stream);
int lg_convT_s2_wgrad_m16(const float* x, const void* x16, const float* dy, const void* dy16, float* dw, void* workspace,
                          size_t ws_bytes, int B, int Hs, int Ws, int cb, int cs, int accumulate, int dtype, void* stream);
int lg_convT_s2_dgrad(const float* dy, const void* pack, float* dx, int B, int Hs, int Ws, int cb, int cs, int dtype,
                      void* stream);
int lg_convT_s2_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t ws_bytes, int B, int Hs,
                      int Ws, int cb, int cs, int accumulate, int dtype, void* stream);
size_t lg_wgrad_workspace_bytes(int B, int Hs, int Ws, int cb, int cs, int dtype);

/* ---- Conv2DTranspose(image_channel,5,(1,1),"same",activation="tanh")  model.py:86-87 ------------- */
/* y[B,H,W,cb] = tanh(convT_s1(x[B,H,W,cs]) + bias) ; kernel [5][5][cb][cs], cb = image channels (3) */
int lg_convT_s1_tanh_fwd(const float* x, const void* pack, const float* bias, float* y, int B, int H, int W, int cb,
                         int cs, int dtype, void* stream);
/* given dpre = dL/d(pre-tanh) [B,H,W,cb]: dx[B,H,W,cs] (may be null), dw (+)=, db (+)= (dw/db may be null) */
int lg_convT_s1_tanh_bwd(const float* x, const float* dpre, const void* pack, float* dx, float* dw, float* db,
                         void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs, int accumulate,
                         int dtype, void* stream);
size_t lg_convT_s1_bwd_workspace_bytes(int B, int H, int W, int cb, int cs, int dtype);
/* bf16 path of the 3-channel layers (this one and the data gradient of Encoder.conv1): 1 if the kernels of this shape
 * read the bf16 mirror of the wide operand alone, so that its fp32 copy need not exist (H, W: the wide operand's map) */
int lg_n3_m16_supported(int H, int W, int cb, int cs, int dtype);
/* as above with x16 = bf16 mirror of x (x may be null where lg_n3_m16_supported); dx16 (may be null): the data
 * gradient is written as bf16 there instead of fp32 to dx (give at most one of dx, dx16) */
int lg_convT_s1_tanh_fwd_m16(const float* x, const void* x16, const void* pack, const float* bias, float* y, int B, int H,
                             int W, int cb, int cs, int dtype, void* stream);
/* the same layer fed with the RAW bf16 conv output z16 of the level below and its statistics records (stats [B][8]):
 * InstanceNormalization + LeakyReLU(alpha) are applied while the input is staged, the normalised tensor is never written
 * (/root/reference/model.py:46-50 feeding model.py:86-87).  LG_ERR_UNSUPPORTED unless ..._z16_supported. */
int lg_convT_s1_tanh_fwd_z16_supported(int H, int W, int cb, int cs, int dtype);
int lg_convT_s1_tanh_fwd_z16(const void* z16, const float* stats, float alpha, const void* pack, const float* bias, float* y,
                             int B, int H, int W, int cb, int cs, int dtype, void* stream);
int lg_convT_s1_tanh_bwd_m16(const float* x, const void* x16, const float* dpre, const void* pack, float* dx, void* dx16,
                             float* dw, float* db, void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs,
                             int accumulate, int dtype, void* stream);

/* bias gradient of any conv layer: db[C] (+)= column sums of dy[M][C]  (C % 4 == 0) */
size_t lg_bias_grad_workspace_bytes(long long M, int C);
int lg_bias_grad(const float* dy, float* db, void* workspace, size_t ws_bytes, long long M, int C, int accumulate,
                 void* stream);

int lg_bias_grad_m16(const float* dy, const void* dy16, float* db, void* workspace, size_t ws_bytes, long long M, int C,
                     int accumulate, void* stream);
/* 1 if the LDS halo-tile kernel covers the shape (mode 0 = conv-form "down", 1 = convT-form "up"); the fp32 copy of a
 * tensor that has a bf16 mirror is then never read by the conv that consumes it */
int lg_conv_halo_supported(int mode, int dtype, int B, int Hm, int Wm, int Cs, int N);

/* ---- InstanceNormalization(axis=None) + LeakyReLU + skip add  instance.py:105-128, model.py:24,46-50 */
size_t lg_instnorm_workspace_bytes(int B, long long L);
/* stats[B][8] = {mu_hi, sigma, a, beta, mu_lo, 0,0,0} of (pre_leaky ? leaky(x) : x); a = gamma/(sigma+1e-3);
 * mu = mu_hi + mu_lo (float-float: the mean is subtracted from every element, its rounding error is coherent) */
int lg_instnorm_stats_stride(void);
int lg_instnorm_leaky_stats(const float* x, float* stats, const float* gamma, const float* beta, void* workspace,
                            size_t ws_bytes, int B, long long L, int pre_leaky, float alpha, void* stream);
/* same; x16_out (may be null) additionally receives bf16(x) (bf16 activation path, layers without fused moments) */
int lg_instnorm_leaky_stats_z16(const float* x, float* stats, const float* gamma, const float* beta, void* workspace,
                                size_t ws_bytes, int B, long long L, int pre_leaky, float alpha, void* x16_out, void* stream);
int lg_instnorm_stats_finalize(const void* partials, int nparts, float* stats, const float* gamma, const float* beta,
                               int B, void* stream);
/* lg_instnorm_stats_finalize + lg_instnorm_leaky_apply_z16 (pre_leaky = 0) in ONE launch: partials = the [B][nparts][3] moment
 * records of a *_fwd_stats conv; stats [B][8] receives the finished records, bit-identical to the two-call form
 * (instance.py:114-127 + model.py:24,50) */
int lg_instnorm_leaky_apply_z16_p(const void* z16, const void* partials, int nparts, const float* gamma, const float* beta,
                                  float* stats, const void* skip, int skip_is_bf16, float* y, void* y16, int B, long long L,
                                  int post_leaky, float alpha, void* stream);
/* y = [post_leaky](a*([pre_leaky](x) - mu) + beta) [+ skip] ; y16 (may be null): bf16 mirror of y, the MFMA operand
 * image the bf16 conv / wgrad kernels consume instead of re-reading and re-rounding the fp32 tensor; y may be
 * null when only the mirror is wanted (at least one of y, y16) */
int lg_instnorm_leaky_apply(const float* x, const float* stats, const float* skip, float* y, void* y16, int B,
                            long long L, int pre_leaky, int post_leaky, float alpha, void* stream);
/* g = dL/dy (before skip), fp32 or (g_is_bf16) bf16 -> dx fp32 and/or its bf16 mirror dx16 (at least one non-null);
 * dgamma/dbeta (device scalars, may be null) */
int lg_instnorm_leaky_bwd(const float* x, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                          float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, int B, long long L,
                          int pre_leaky, int post_leaky, float alpha, int accumulate, void* stream);
/* same, plus db[C] (+)= column sums of dx viewed as [B*L/C][C]: the bias gradient of the conv layer that produced x
 * (C = its channels, innermost), taken in the pass that writes dx (no separate lg_bias_grad pass over dx) */
size_t lg_instnorm_bwd_db_workspace_bytes(int B, long long L, int C);
int lg_instnorm_leaky_bwd_db(const float* x, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                             float* dgamma, float* dbeta, float* db, int C, void* workspace, size_t ws_bytes, int B,
                             long long L, int pre_leaky, int post_leaky, float alpha, int accumulate, void* stream);
/* bf16 activation path: the same two ops with the conv output z given as bf16 (z16); skip (may be null) is fp32, or bf16
 * when skip_is_bf16.  L % 8 == 0.  Workspace of the backward: lg_instnorm_bwd_db_workspace_bytes. */
int lg_instnorm_leaky_apply_z16(const void* z16, const float* stats, const void* skip, int skip_is_bf16, float* y, void* y16,
                                int B, long long L, int pre_leaky, int post_leaky, float alpha, void* stream);
int lg_instnorm_leaky_bwd_z16(const void* z16, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                              float* dgamma, float* dbeta, float* db, int C, void* workspace, size_t ws_bytes, int B,
                              long long L, int pre_leaky, int post_leaky, float alpha, int accumulate, void* stream);
/* same; partials / nparts_in (may be null / 0): the sums of the first pass as [B][nparts_in][2] doubles, already written by
 * the conv epilogue that produced g (lg_*_dgrad_nf below) - that pass is then skipped */
int lg_instnorm_leaky_bwd_z16_p(const void* z16, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                                float* dgamma, float* dbeta, float* db, int C, const void* partials, int nparts_in,
                                void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky, int post_leaky, float alpha,
                                int accumulate, void* stream);
/* data gradients of the bf16 activation path that ALSO write the first-pass sums of the InstanceNormalization backward of
 * the layer the gradient belongs to (z16 / stats: that layer's bf16 conv output and statistics records; instance.py:105-128,
 * model.py:22-24,46-50): *nparts > 0 -> part holds [B][*nparts][2] doubles for lg_instnorm_leaky_bwd_z16_p; 0 -> not
 * produced for this shape (the gradient itself is always written).  part_bytes >= lg_conv_stats_workspace_bytes(...). */
int lg_conv2d_s2_dgrad_nf(const void* dy16, const void* pack, void* dx16, int B, int Hs, int Ws, int cb, int cs,
                          const void* z16, const float* stats, float alpha, void* part, size_t part_bytes, int* nparts,
                          void* stream);
int lg_convT_s2_dgrad_nf(const void* dy16, const void* pack, void* dx16, int B, int Hs, int Ws, int cb, int cs,
                         const void* z16, const float* stats, float alpha, void* part, size_t part_bytes, int* nparts,
                         void* stream);

/* The same data gradient fed with the level's RAW pair (z16, g16) instead of dz16: the InstanceNorm + LeakyReLU backward of the
 * level (instance.py:105-128 differentiated) is applied while the operand is staged, from the per-sample records `coef`
 * (lg_instnorm_bwd_coef), so the norm-backward apply pass and the dz tensor do not exist.  For tapes that ask the level for no
 * weight gradient (eager_trainer.py:158-163: the Adjuster differentiates its dense + norm only).  Always with the fused sums of the
 * level below (zl16, stats_l, alpha_l -> part / nparts, as lg_convT_s2_dgrad_nf).  Bit-identical to lg_instnorm_leaky_bwd_z16_p +
 * lg_convT_s2_dgrad_nf.  Hs x Ws = the small (output) map; z16, g16: [B][2Hs][2Ws][cb]; dx16: [B][Hs][Ws][cs]. */
int lg_convT_s2_dgrad_bn_supported(int B, int Hs, int Ws, int cb, int cs);
int lg_convT_s2_dgrad_bn(const void* z16, const void* g16, const float* coef, float alpha, const void* pack, void* dx16,
                         int B, int Hs, int Ws, int cb, int cs, const void* zl16, const float* stats_l, float alpha_l,
                         void* part, size_t part_bytes, int* nparts, void* stream);
/* coef[B][8] = {mu_hi, mu_lo, a, beta, m1_hi, m2'_hi, m1_lo, m2'_lo} from the statistics records and the producer-fused sums
 * {sum g', sum g' c} ([B][nparts][2] doubles): the norm backward without its apply pass (L = elements per sample) */
int lg_instnorm_bwd_coef(const float* stats, const void* partials, int nparts, float* coef, int B, long long L, void* stream);

int lg_convT_s1_tanh_bwd_nf(const float* x, const void* x16, const float* dpre, const void* pack, void* dx16, float* dw,
                            float* db, void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs, int accumulate,
                            int dtype, const void* z16, const float* stats, float alpha, void* part, size_t part_bytes,
                            int* nparts, void* stream);

/* ---- tf.compat.v1.layers.Dense  model.py:62-63 (heads, sigmoid), :83, :120 ----------------------- */
int lg_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, void* stream);
int lg_dense_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int K, int N, int accumulate,
                   void* stream);
/* dx[B][K] = dy[B][N] @ w[K][N]^T (not needed by the step's tapes: the dense inputs are noise / conditions) */
int lg_dense_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, void* stream);
/* out[B][ka+kc] = [a | c]: keras.layers.concatenate([input_noise, input_cond], axis=-1) in front of the Generator's
   dense layer, model.py:97-98 */
int lg_concat_cols(const float* a, int ka, const float* c, int kc, float* out, int B, void* stream);
/* t[2B][c] = tf.concat([first, second], 0), u = (t + 1) * 0.5: the Adjuster's target / input conditions,
   eager_trainer.py:153-154 (first = real_cond_2, second = real_cond_1) */
int lg_adj_conditions(const float* first, const float* second, float* t, float* u, int B, int c, void* stream);
/* p[B][1+c] = sigmoid(x[B][K] @ [wpr | wc] + [bpr | bc]) : column 0 = output_pr, 1.. = output_cond */
size_t lg_heads_fwd_workspace_bytes(int B, int K, int c);
int lg_heads_fwd(const float* x, const float* wpr, const float* bpr, const float* wc, const float* bc, float* p,
                 void* workspace, size_t ws_bytes, int B, int K, int c, void* stream);
int lg_heads_dgrad(const float* dz, const float* wpr, const float* wc, float* dx, int B, int K, int c, void* stream);
int lg_heads_wgrad(const float* x, const float* dz, float* dwpr, float* dbpr, float* dwc, float* dbc, int B, int K,
                   int c, int accumulate, void* stream);

/* ---- losses  eager_trainer.py:85-102 -------------------------------------------------------------- */
/* loss (+)= w_pr*mean BCE(t_pr, p[:,0]) + w_c*mean BCE(t_c, p[:,1:]) ; dz[B][1+c] = dloss/dlogits */
int lg_bce_heads_loss_fwd_bwd(const float* p, const float* t_c, float t_pr, float w_pr, float w_c, float* loss,
                              float* dz, int B, int c, int accumulate, void* stream);
size_t lg_l1_workspace_bytes(void);
/* loss (+)= lambda*mean|t-img| ; dpre = (g_in - lambda*sign(t-img)/n)*(1-img^2) (g_in, dpre may be null) */
int lg_l1_tanh_loss_fwd_bwd(const float* t, const float* img, const float* g_in, float* dpre, float* loss,
                            void* workspace, size_t ws_bytes, long long n, float lambda, int accumulate, void* stream);

/* ---- tf.clip_by_value + tf.compat.v1.train.AdamOptimizer  eager_trainer.py:28-30,146-148,164-168 -- */
/* state = {beta1_power, beta2_power} on the device; g is scaled by gscale (1/world_size) then clipped */
int lg_clip_adam_update(float* w, const float* g, float* m, float* v, long long n, const float* state, float lr,
                        float b1, float b2, float eps, float clip, float gscale, void* stream);
int lg_adam_advance(float* state, float b1, float b2, void* stream);
int lg_axpby(float* y, const float* x, float a, float b, long long n, void* stream);

/* ---- step inputs drawn on the device  eager_trainer.py:125-131 ------------------------ */
/* counter-based Philox4x32-10: block i = philox(counter = offset + i, key = seed); 4 x 32 bits per block */
int lg_philox4x32(unsigned* out, int nblocks, unsigned long long seed, unsigned long long offset, void* stream);
/* out[i] = mean + std * N(0,1) (tf.random.normal :125); element i = normal (i & 3) of block offset + i/4 */
int lg_randn(float* out, long long n, float mean, float std, unsigned long long seed, unsigned long long offset,
             void* stream);
size_t lg_augment_workspace_bytes(int B);
/* random_flip_left_right / random_brightness / random_contrast / random_hue / + noise (:127-131) with the draws
 * given by the caller: flip[B] bytes (may be null), brightness delta db, contrast factor cf (about the per-image,
 * per-channel mean), hue delta dh (turns), noise_scale * N(0,1) per element (Philox block offset + pixel index) */
int lg_augment(const float* img, float* out, int B, int H, int W, const unsigned char* flip, float db, float cf, float dh,
               float noise_scale, unsigned long long seed, unsigned long long offset, void* workspace, size_t ws_bytes,
               void* stream);
size_t lg_augment_drawn_workspace_bytes(int B);
/* the same with the random draws of :127-130 made on the device (no host round trip on the step's input side):
 * word w of the Philox window at draw_offset gives u_w = (bits >> 8) / 2^24;  db = (2 u_0 - 1) db_max,
 * cf = c_lo + u_1 (c_hi - c_lo), dh = (2 u_2 - 1) dh_max, image n is flipped iff u_{3+n} < 1/2 */
int lg_augment_drawn(const float* img, float* out, int B, int H, int W, float db_max, float c_lo, float c_hi, float dh_max,
                     float noise_scale, unsigned long long seed, unsigned long long draw_offset,
                     unsigned long long noise_offset, void* workspace, size_t ws_bytes, void* stream);

/* ---- FID activation statistics  fid.py:185-188 ----------------------------------------------- */
/* mu[D] = mean over the N samples, sigma[D][D] = np.cov(act, rowvar=False) (divisor N - 1) of act[N][D] (fp32), both fp64
 * on the device; Gram matrix of the centred activations on the fp64 matrix instruction.  N >= 2. */
size_t lg_fid_stats_workspace_bytes(long long N, int D);
int lg_fid_stats(const float* act, long long N, int D, double* mu, double* sigma, void* workspace, size_t ws_bytes,
                 void* stream);

/* ---- run-time services (no reference counterpart: the reference has no distributed code and no clock to report) ---- */
/* CU budget of the persistent kernels.  Under data parallelism RCCL's ring kernels occupy CUs on a side stream while the
 * backward convs run (littlegan_amd/dist.py); every persistent launcher sizes its grid to lg_grid_cus() = CUs - reserved
 * blocks-per-CU multiples, so that all of its blocks are resident from the start.  Process-wide, default 0. */
int lg_device_cus(void);
int lg_set_reserved_cus(int n);
int lg_grid_cus(void);
/* one-GPU rehearsal of an all-reduce's CU footprint (bench.py --dp-contention): `workgroups` blocks of `threads` (256 | 512)
 * stream dst = 0.5 dst + src over n floats (n % 4 == 0), `passes` times — what a ring step does to the CUs it occupies */
int lg_contention_probe(float* dst, const float* src, long long n, int workgroups, int threads, int passes, void* stream);
/* in-kernel clock census: while buf3 (3 x 64-bit, device memory, zeroed by the caller) is registered, every block of the kernels that
 * support it (conv_down3: the step's dominant kernel) adds {d s_memtime, d s_memrealtime (100 MHz), 1} of its own life to it;
 * nullptr switches it off */
int lg_set_clock_census(unsigned long long* buf3);
/* one wave on the COMPUTE stream that spins spin_us microseconds behind the previous kernel and ADDS {d s_memtime, d s_memrealtime, 1}
 * to out3: the clock the chip holds at that point of the step */
int lg_clock_sample(unsigned long long* out3, int spin_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif
