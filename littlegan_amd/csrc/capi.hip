// Typed C-ABI entry points (include/littlegan_hip.h) on top of the shared kernels.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include "lg_common.h"
#include "../../include/littlegan_hip.h"

static thread_local char g_err[512] = "";

extern "C" void lg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lg_last_error(void) { return g_err; }
static thread_local const char* g_kernel = "";
extern "C" void lg_note_kernel(const char* name) { g_kernel = name ? name : ""; }
extern "C" const char* lg_last_kernel(void) { return g_kernel; }
extern "C" int lg_clear_kernel(void) { g_kernel = ""; return LG_OK; }
extern "C" int lg_abi_version(void) { return LG_ABI_VERSION; }

extern "C" int lg_conv_igemm(int mode, int dtype, const float* src, const void* wpack, const float* bias, float* out,
                             int B, int Hm, int Wm, int Cs, int N, int act, int pstride, int ppad, void* stream);
extern "C" int lg_conv_igemm_ex(int mode, int dtype, const float* src, const void* src16, const void* wpack,
                                const float* bias, float* out, void* out16, int B, int Hm, int Wm, int Cs, int N, int act,
                                int pstride, int ppad, void* spart, size_t spart_bytes, int* nparts_out, void* stream);
extern "C" int lg_conv_wgrad_m16(const float* big, const void* big16, const float* small, const void* small16, float* dw,
                                 void* workspace, size_t ws_bytes, int B, int Hm, int Wm, int cb, int cs, int pstride,
                                 int ppad, int accumulate, int dtype, void* stream);
extern "C" size_t lg_conv_pack_up_offset(int cb, int cs, int dtype);
extern "C" size_t lg_conv_pack_raw_offset(int cb, int cs, int dtype);
extern "C" int lg_n3_s1t_fwd_try(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                 void* stream);
extern "C" int lg_n3_up_try(const float* src, const float* w, float* out, int B, int H, int W, int C, void* stream);
extern "C" int lg_n3_p16_supported(int H, int W, int C);
extern "C" int lg_n3_s1t_fwd_p16_try(const void* x16, const float* w, const float* bias, float* y, int B, int H, int W,
                                     int C, void* stream);
extern "C" int lg_n3_up_p16_try(const void* src16, const float* w, float* out, int B, int H, int W, int C, void* stream);
extern "C" int lg_n3_conv1_fwd_p16_try(const float* img, const float* w, const float* bias, float* z, void* z16, int B, int H,
                                       int W, int N, void* spart, size_t spart_bytes, int* nparts, void* stream);
extern "C" int lg_n3_s1_dgrad_p16_try(const float* dpre, const float* w, float* dx, void* dx16, int B, int H, int W, int N,
                                      void* stream);
extern "C" int lg_n3_rows_supported(int H, int W, int C);
extern "C" int lg_n3_s1t_fwd_rows_try(const void* x16, const float* stats, float alpha, const float* w, const float* bias, float* y,
                                      int B, int H, int W, int C, void* stream);
static bool n3_enabled() {
  static int v = -1;
  if (v < 0) v = lg_env_flag("LG_NO_N3") ? 0 : 1;  // A/B switch
  return v == 1;
}
static inline const float* raw_pack(const void* pack, int cb, int cs, int dtype) {
  return (const float*)((const char*)pack + lg_conv_pack_raw_offset(cb, cs, dtype));
}
extern "C" int lg_conv_wgrad(const float* big, const float* small, float* dw, void* workspace, size_t ws_bytes, int B,
                             int Hm, int Wm, int cb, int cs, int pstride, int ppad, int accumulate, int dtype,
                             void* stream);

enum { MODE_DOWN = 0, MODE_UP = 1, MODE_S1T = 2, MODE_PATCH = 3 };

static inline const void* up_pack(const void* pack, int cb, int cs, int dtype) {
  return (const char*)pack + lg_conv_pack_up_offset(cb, cs, dtype);
}

extern "C" int lg_conv_halo_supported(int mode, int dtype, int B, int Hm, int Wm, int Cs, int N);
extern "C" int lg_n3_conv1_p16_supported(int H, int W, int N);

// 1 if lg_conv2d_s2_fwd_stats (up == 0) / lg_convT_s2_fwd_stats (up == 1) on this shape returns fused moment partials
// (*nparts > 0) when given a workspace of lg_conv_stats_workspace_bytes: the kernel's tiling puts one sample per block.
// The bf16 activation path asks BEFORE the call, because only then may the conv write its result as bf16 alone.
extern "C" int lg_conv_fwd_stats_fused(int up, int dtype, int B, int Hs, int Ws, int cb, int cs) {
  if (B <= 0 || Hs <= 0 || Ws <= 0) return 0;
  if (cb == 3) return (!up && dtype == LG_DT_BF16 && n3_enabled() && lg_n3_conv1_p16_supported(Hs, Ws, cs)) ? 1 : 0;
  if (lg_env_flag("LG_NO_HALO")) return 0;
  if ((long long)Hs * Ws < 128 && !(Hs == 8 && Ws == 8)) return 0;  // > 2 samples per 128-row tile: no per-sample record
  return up ? lg_conv_halo_supported(1, dtype, B, Hs, Ws, cs, cb) : lg_conv_halo_supported(0, dtype, B, Hs, Ws, cb, cs);
}

// "down": big [B,2Hs,2Ws,cb] -> small [B,Hs,Ws,cs]
static int run_down(const float* big, const void* pack, const float* bias, float* small, int B, int Hs, int Ws, int cb,
                    int cs, int dtype, void* stream) {
  if (cb == 3) return lg_conv_igemm(MODE_PATCH, dtype, big, pack, bias, small, B, Hs, Ws, 3, cs, 0, 2, 1, stream);
  return lg_conv_igemm(MODE_DOWN, dtype, big, pack, bias, small, B, Hs, Ws, cb, cs, 0, 0, 0, stream);
}
// "up": small [B,Hs,Ws,cs] -> big [B,2Hs,2Ws,cb]
static int run_up(const float* small, const void* pack, const float* bias, float* big, int B, int Hs, int Ws, int cb,
                  int cs, int dtype, void* stream) {
  if (cb == 3 && !bias && n3_enabled()) {  // image-side data gradient: VALU kernel, exact f32 in both dtypes
    const int rc = lg_n3_up_try(small, raw_pack(pack, cb, cs, dtype), big, B, Hs, Ws, cs, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  return lg_conv_igemm(MODE_UP, dtype, small, up_pack(pack, cb, cs, dtype), bias, big, B, Hs, Ws, cs, cb, 0, 0, 0,
                       stream);
}

extern "C" int lg_conv2d_s2_fwd(const float* x, const void* pack, const float* bias, float* y, int B, int Hs, int Ws,
                                int cb, int cs, int dtype, void* stream) {
  return run_down(x, pack, bias, y, B, Hs, Ws, cb, cs, dtype, stream);
}
// forward + fused InstanceNorm moment partials (finish with lg_instnorm_stats_finalize when *nparts > 0);
// x16 (optional): bf16 mirror of x for the bf16 MFMA path
// y16 (optional, bf16 path): the result z is written as bf16 THERE instead of fp32 to y (y may then be null); the moment
// partials are taken from the fp32 accumulators either way
extern "C" int lg_conv2d_s2_fwd_stats(const float* x, const void* x16, const void* pack, const float* bias, float* y,
                                      void* y16, int B, int Hs, int Ws, int cb, int cs, int dtype, void* spart,
                                      size_t spart_bytes, int* nparts, void* stream) {
  if (nparts) *nparts = 0;
  LG_CHECK_ARG(!y16 || dtype == LG_DT_BF16, "lg_conv2d_s2_fwd_stats: a bf16 result needs dtype bf16");
  if (cb == 3) {
    if (dtype == LG_DT_BF16 && n3_enabled()) {  // bf16 path: patch kernel with coalesced row stores + fused moments
      const int rc = lg_n3_conv1_fwd_p16_try(x, raw_pack(pack, cb, cs, dtype), bias, y16 ? nullptr : y, y16, B, Hs, Ws, cs,
                                             spart, spart_bytes, nparts, stream);
      if (rc != LG_ERR_UNSUPPORTED) return rc;
    }
    return lg_conv_igemm_ex(MODE_PATCH, dtype, x, nullptr, pack, bias, y16 ? nullptr : y, y16, B, Hs, Ws, 3, cs, 0, 2, 1, nullptr,
                            0, nullptr, stream);
  }
  return lg_conv_igemm_ex(MODE_DOWN, dtype, x, x16, pack, bias, y16 ? nullptr : y, y16, B, Hs, Ws, cb, cs, 0, 0, 0, spart,
                          spart_bytes, nparts, stream);
}
// The same forward pass fed with the RAW bf16 conv output z16 [B,2Hs,2Ws,cb] of the level below and its statistics records
// (zstats [B][8]): x = bf16(LeakyReLU_alpha(InstanceNorm(z))) is formed while the kernel stages its operand (bit-identical to
// lg_instnorm_leaky_apply_z16's output), the normalised map is never written.  For passes whose normalised maps have no other
// reader (no weight gradient of this layer, no skip use): the discriminator run on the Adjuster's output.  The result is bf16
// (y16) with fused moment partials, as lg_conv2d_s2_fwd_stats on the bf16 path.  LG_ERR_UNSUPPORTED unless ..._zn_supported.
extern "C" int lg_conv_down3_zn_supported(int B, int Hm, int Wm, int Cs, int N);
extern "C" int lg_conv_down3_zn_try(const void* z16, const float* zstats, float alpha, const void* wpack, const float* bias, void* out16,
                                    int B, int Hm, int Wm, int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream);
extern "C" int lg_conv2d_s2_fwd_stats_zn_supported(int B, int Hs, int Ws, int cb, int cs, int dtype) {
  return (dtype == LG_DT_BF16 && cb != 3 && !lg_env_flag("LG_NO_HALO") && lg_conv_down3_zn_supported(B, Hs, Ws, cb, cs)) ? 1 : 0;
}
extern "C" int lg_conv2d_s2_fwd_stats_zn(const void* z16, const float* zstats, float alpha, const void* pack, const float* bias,
                                         void* y16, int B, int Hs, int Ws, int cb, int cs, int dtype, void* spart,
                                         size_t spart_bytes, int* nparts, void* stream) {
  if (nparts) *nparts = 0;
  LG_CHECK_ARG(z16 && zstats && pack && bias && y16 && spart && nparts, "lg_conv2d_s2_fwd_stats_zn: null pointer");
  if (!lg_conv2d_s2_fwd_stats_zn_supported(B, Hs, Ws, cb, cs, dtype)) return LG_ERR_UNSUPPORTED;
  return lg_conv_down3_zn_try(z16, zstats, alpha, pack, bias, y16, B, Hs, Ws, cb, cs, spart, spart_bytes, nparts, stream);
}
extern "C" int lg_convT_s2_fwd_stats(const float* x, const void* x16, const void* pack, const float* bias, float* y,
                                     void* y16, int B, int Hs, int Ws, int cb, int cs, int dtype, void* spart,
                                     size_t spart_bytes, int* nparts, void* stream) {
  if (nparts) *nparts = 0;
  LG_CHECK_ARG(!y16 || (dtype == LG_DT_BF16 && cb != 3), "lg_convT_s2_fwd_stats: a bf16 result needs dtype bf16, cb != 3");
  if (cb == 3) return run_up(x, pack, bias, y, B, Hs, Ws, cb, cs, dtype, stream);
  return lg_conv_igemm_ex(MODE_UP, dtype, x, x16, up_pack(pack, cb, cs, dtype), bias, y16 ? nullptr : y, y16, B, Hs, Ws, cs, cb,
                          0, 0, 0, spart, spart_bytes, nparts, stream);
}
// data / weight gradients with optional bf16 mirrors of their activation operands
extern "C" int lg_conv2d_s2_dgrad_m16(const float* dy, const void* dy16, const void* pack, float* dx, void* dx16, int B,
                                      int Hs, int Ws, int cb, int cs, int dtype, void* stream) {
  if (cb == 3) {
    LG_CHECK_ARG(dx && !dx16, "lg_conv2d_s2_dgrad_m16: the 3-channel image gradient is fp32 only");
    if (dtype == LG_DT_BF16 && dy16 && n3_enabled()) {  // bf16 path: tap-product GEMM straight from the bf16 mirror
      const int rc = lg_n3_up_p16_try(dy16, raw_pack(pack, cb, cs, dtype), dx, B, Hs, Ws, cs, stream);
      if (rc != LG_ERR_UNSUPPORTED) return rc;
    }
    LG_CHECK_ARG(dy, "lg_conv2d_s2_dgrad_m16: this shape needs the fp32 gradient (see lg_n3_m16_supported)");
    return run_up(dy, pack, nullptr, dx, B, Hs, Ws, cb, cs, dtype, stream);
  }
  return lg_conv_igemm_ex(MODE_UP, dtype, dy, dy16, up_pack(pack, cb, cs, dtype), nullptr, dx, dx16, B, Hs, Ws, cs, cb, 0, 0,
                          0, nullptr, 0, nullptr, stream);
}
extern "C" int lg_convT_s2_dgrad_m16(const float* dy, const void* dy16, const void* pack, float* dx, void* dx16, int B,
                                     int Hs, int Ws, int cb, int cs, int dtype, void* stream) {
  if (cb == 3) {
    LG_CHECK_ARG(dx && !dx16, "lg_convT_s2_dgrad_m16: 3-channel layers are fp32 only");
    return run_down(dy, pack, nullptr, dx, B, Hs, Ws, cb, cs, dtype, stream);
  }
  return lg_conv_igemm_ex(MODE_DOWN, dtype, dy, dy16, pack, nullptr, dx, dx16, B, Hs, Ws, cb, cs, 0, 0, 0, nullptr, 0, nullptr,
                          stream);
}
// ---- data gradients that also emit the norm-backward sums of the gradient they write (bf16 activation path) -----------
// The gradient g = dL/dh of the layer BELOW feeds that layer's InstanceNormalization backward, whose first pass needs
// per-sample sums over (z, g).  Given that layer's bf16 conv output z16 and statistics records, the kernels that cover
// the shape add the sums to their epilogue: *nparts > 0 records per sample in part ([B][*nparts][2] doubles), to be
// handed to lg_instnorm_leaky_bwd_z16_p.  *nparts == 0: not produced (shape not covered) - run the ordinary backward.
extern "C" int lg_conv_up3_nf_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                                  int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                  size_t nf_bytes, void* stream);
extern "C" int lg_conv_up4_nf_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                                  int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                  size_t nf_bytes, void* stream);
extern "C" int lg_conv_down3_nf_try(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm,
                                    int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                    size_t nf_bytes, void* stream);
extern "C" int lg_n3_s1_dgrad_p16_nf_try(const float* dpre, const float* w, float* dx, void* dx16, int B, int H, int W, int N,
                                         const LgNormFuse* nf, size_t nf_bytes, int* nparts_out, void* stream);

extern "C" int lg_conv2d_s2_dgrad_nf(const void* dy16, const void* pack, void* dx16, int B, int Hs, int Ws, int cb, int cs,
                                     const void* z16, const float* stats, float alpha, void* part, size_t part_bytes,
                                     int* nparts, void* stream) {
  LG_CHECK_ARG(dy16 && pack && dx16 && nparts, "lg_conv2d_s2_dgrad_nf: null pointer");
  *nparts = 0;
  if (cb != 3 && z16 && stats && part && n3_enabled()) {
    LgNormFuse nf{(const __bf16*)z16, stats, (double*)part, alpha, 0};
    int rc = lg_conv_up3_nf_try(dy16, up_pack(pack, cb, cs, LG_DT_BF16), nullptr, dx16, B, Hs, Ws, cs, cb, nullptr, 0, nparts,
                                &nf, part_bytes, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
    *nparts = 0;
    rc = lg_conv_up4_nf_try(dy16, up_pack(pack, cb, cs, LG_DT_BF16), nullptr, dx16, B, Hs, Ws, cs, cb, nullptr, 0, nparts, &nf,
                            part_bytes, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
    *nparts = 0;
  }
  return lg_conv2d_s2_dgrad_m16(nullptr, dy16, pack, nullptr, dx16, B, Hs, Ws, cb, cs, LG_DT_BF16, stream);
}
extern "C" int lg_convT_s2_dgrad_nf(const void* dy16, const void* pack, void* dx16, int B, int Hs, int Ws, int cb, int cs,
                                    const void* z16, const float* stats, float alpha, void* part, size_t part_bytes,
                                    int* nparts, void* stream) {
  LG_CHECK_ARG(dy16 && pack && dx16 && nparts, "lg_convT_s2_dgrad_nf: null pointer");
  *nparts = 0;
  if (cb != 3 && z16 && stats && part) {
    LgNormFuse nf{(const __bf16*)z16, stats, (double*)part, alpha, 0};
    const int rc = lg_conv_down3_nf_try(dy16, pack, nullptr, dx16, B, Hs, Ws, cb, cs, nullptr, 0, nparts, &nf, part_bytes, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
    *nparts = 0;
  }
  return lg_convT_s2_dgrad_m16(nullptr, dy16, pack, nullptr, dx16, B, Hs, Ws, cb, cs, LG_DT_BF16, stream);
}

extern "C" int lg_conv_down3_bn_supported(int B, int Hm, int Wm, int Cs, int N);
extern "C" int lg_conv_down3_bn_try(const void* z16, const void* g16, const float* bcoef, float alpha, const void* wpack, void* out16,
                                    int B, int Hm, int Wm, int Cs, int N, int* nparts_out, const LgNormFuse* nf, size_t nf_bytes,
                                    void* stream);
extern "C" int lg_convT_s2_dgrad_bn_supported(int B, int Hs, int Ws, int cb, int cs) {
  return cb != 3 ? lg_conv_down3_bn_supported(B, Hs, Ws, cb, cs) : 0;
}
extern "C" int lg_convT_s2_dgrad_bn(const void* z16, const void* g16, const float* coef, float alpha, const void* pack, void* dx16,
                                    int B, int Hs, int Ws, int cb, int cs, const void* zl16, const float* stats_l, float alpha_l,
                                    void* part, size_t part_bytes, int* nparts, void* stream) {
  LG_CHECK_ARG(z16 && g16 && coef && pack && dx16 && zl16 && stats_l && part && nparts, "lg_convT_s2_dgrad_bn: null pointer");
  *nparts = 0;
  LgNormFuse nf{(const __bf16*)zl16, stats_l, (double*)part, alpha_l, 0};
  const int rc = lg_conv_down3_bn_try(z16, g16, coef, alpha, pack, dx16, B, Hs, Ws, cb, cs, nparts, &nf, part_bytes, stream);
  if (rc == LG_ERR_UNSUPPORTED) lg_set_error("lg_convT_s2_dgrad_bn: shape B=%d %dx%d cb=%d cs=%d has no backward-normalising kernel "
                                             "(see lg_convT_s2_dgrad_bn_supported)", B, Hs, Ws, cb, cs);
  return rc;
}

extern "C" int lg_conv2d_s2_wgrad_m16(const float* x, const void* x16, const float* dy, const void* dy16, float* dw,
                                      void* workspace, size_t ws_bytes, int B, int Hs, int Ws, int cb, int cs,
                                      int accumulate, int dtype, void* stream) {
  return lg_conv_wgrad_m16(x, x16, dy, dy16, dw, workspace, ws_bytes, B, Hs, Ws, cb, cs, 2, 1, accumulate, dtype, stream);
}
extern "C" int lg_convT_s2_wgrad_m16(const float* x, const void* x16, const float* dy, const void* dy16, float* dw,
                                     void* workspace, size_t ws_bytes, int B, int Hs, int Ws, int cb, int cs,
                                     int accumulate, int dtype, void* stream) {
  return lg_conv_wgrad_m16(dy, dy16, x, x16, dw, workspace, ws_bytes, B, Hs, Ws, cb, cs, 2, 1, accumulate, dtype, stream);
}
extern "C" int lg_conv2d_s2_dgrad(const float* dy, const void* pack, float* dx, int B, int Hs, int Ws, int cb, int cs,
                                  int dtype, void* stream) {
  return run_up(dy, pack, nullptr, dx, B, Hs, Ws, cb, cs, dtype, stream);
}
extern "C" int lg_conv2d_s2_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t ws_bytes, int B,
                                  int Hs, int Ws, int cb, int cs, int accumulate, int dtype, void* stream) {
  return lg_conv_wgrad(x, dy, dw, workspace, ws_bytes, B, Hs, Ws, cb, cs, 2, 1, accumulate, dtype, stream);
}
extern "C" int lg_convT_s2_fwd(const float* x, const void* pack, const float* bias, float* y, int B, int Hs, int Ws,
                               int cb, int cs, int dtype, void* stream) {
  return run_up(x, pack, bias, y, B, Hs, Ws, cb, cs, dtype, stream);
}
extern "C" int lg_convT_s2_dgrad(const float* dy, const void* pack, float* dx, int B, int Hs, int Ws, int cb, int cs,
                                 int dtype, void* stream) {
  return run_down(dy, pack, nullptr, dx, B, Hs, Ws, cb, cs, dtype, stream);
}
extern "C" int lg_convT_s2_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t ws_bytes, int B,
                                 int Hs, int Ws, int cb, int cs, int accumulate, int dtype, void* stream) {
  return lg_conv_wgrad(dy, x, dw, workspace, ws_bytes, B, Hs, Ws, cb, cs, 2, 1, accumulate, dtype, stream);
}

// 1 if the 3-channel layers of this shape run from the bf16 mirror alone in the bf16 path (no fp32 operand needed)
extern "C" int lg_n3_m16_supported(int H, int W, int cb, int cs, int dtype) {
  return (dtype == LG_DT_BF16 && cb == 3 && n3_enabled() && lg_n3_p16_supported(H, W, cs)) ? 1 : 0;
}

extern "C" int lg_convT_s1_tanh_fwd_m16(const float* x, const void* x16, const void* pack, const float* bias, float* y,
                                        int B, int H, int W, int cb, int cs, int dtype, void* stream) {
  if (x16 && lg_n3_m16_supported(H, W, cb, cs, dtype)) {
    int rc = lg_n3_s1t_fwd_rows_try(x16, nullptr, 0.f, raw_pack(pack, cb, cs, dtype), bias, y, B, H, W, cs, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
    rc = lg_n3_s1t_fwd_p16_try(x16, raw_pack(pack, cb, cs, dtype), bias, y, B, H, W, cs, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  LG_CHECK_ARG(x, "lg_convT_s1_tanh_fwd_m16: this shape needs the fp32 input (see lg_n3_m16_supported)");
  return lg_convT_s1_tanh_fwd(x, pack, bias, y, B, H, W, cb, cs, dtype, stream);
}

// 1 if lg_convT_s1_tanh_fwd_z16 runs this shape (InstanceNorm + LeakyReLU of the input applied while it is staged)
extern "C" int lg_convT_s1_tanh_fwd_z16_supported(int H, int W, int cb, int cs, int dtype) {
  return (dtype == LG_DT_BF16 && cb == 3 && n3_enabled() && !lg_env_flag("LG_NO_ROWS") && lg_n3_rows_supported(H, W, cs)) ? 1 : 0;
}

// y = tanh(convT_s1(h) + bias) with h = bf16(LeakyReLU_alpha(a*((z - mu) - mu_lo) + beta)) formed on the fly from the raw
// bf16 conv output z16 [B,H,W,cs] and its statistics records stats [B][8] (lg_instnorm_leaky_stats layout): the
// normalised activation is never written.  LG_ERR_UNSUPPORTED outside lg_convT_s1_tanh_fwd_z16_supported.
extern "C" int lg_convT_s1_tanh_fwd_z16(const void* z16, const float* stats, float alpha, const void* pack, const float* bias,
                                        float* y, int B, int H, int W, int cb, int cs, int dtype, void* stream) {
  LG_CHECK_ARG(z16 && stats && pack && bias && y, "lg_convT_s1_tanh_fwd_z16: null pointer");
  if (!lg_convT_s1_tanh_fwd_z16_supported(H, W, cb, cs, dtype)) return LG_ERR_UNSUPPORTED;
  return lg_n3_s1t_fwd_rows_try(z16, stats, alpha, raw_pack(pack, cb, cs, dtype), bias, y, B, H, W, cs, stream);
}

extern "C" int lg_convT_s1_tanh_fwd(const float* x, const void* pack, const float* bias, float* y, int B, int H, int W,
                                    int cb, int cs, int dtype, void* stream) {
  if (cb == 3 && n3_enabled()) {
    const int rc = lg_n3_s1t_fwd_try(x, raw_pack(pack, cb, cs, dtype), bias, y, B, H, W, cs, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  return lg_conv_igemm(MODE_S1T, dtype, x, up_pack(pack, cb, cs, dtype), bias, y, B, H, W, cs, cb, 1, 0, 0, stream);
}

namespace {
// column sums of [M][3] (M % 4 == 0): 12 floats = 4 pixels per thread-iteration
__global__ __launch_bounds__(256) void colsum3_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                      long long n12) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  float s[3] = {0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n12; i += stride) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 12);
    const f32x4 b = *reinterpret_cast<const f32x4*>(x + i * 12 + 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(x + i * 12 + 8);
    s[0] += (a[0] + a[3]) + (b[2] + c[1]);
    s[1] += (a[1] + b[0]) + (b[3] + c[2]);
    s[2] += (a[2] + b[1]) + (c[0] + c[3]);
  }
  __shared__ float sred[48];
  lg_block_sum<3>(s, sred);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = s[0];
    partial[blockIdx.x * 3 + 1] = s[1];
    partial[blockIdx.x * 3 + 2] = s[2];
  }
}
// one wave per column: lane l adds partials l, l + 64, ... in fp64, then a fixed-order wave sum (three threads walking 512 dependent
// loads each took 40 us)
__global__ __launch_bounds__(192) void colsum3_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ db, int accumulate) {
  const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int i = lane; i < nb; i += 64) s += (double)partial[i * 3 + j];
  s = lg_wave_sum_d(s);
  if (lane == 0) db[j] = (accumulate ? db[j] : 0.f) + (float)s;
}
}  // namespace

extern "C" size_t lg_convT_s1_bwd_workspace_bytes(int B, int H, int W, int cb, int cs, int dtype) {
  size_t a = lg_wgrad_workspace_bytes(B, H, W, cb, cs, dtype);
  size_t b = 512 * 3 * sizeof(float);
  return a > b ? a : b;
}

extern "C" int lg_convT_s1_tanh_bwd_m16(const float* x, const void* x16, const float* dpre, const void* pack, float* dx,
                                        void* dx16, float* dw, float* db, void* workspace, size_t ws_bytes, int B, int H,
                                        int W, int cb, int cs, int accumulate, int dtype, void* stream);

extern "C" int lg_convT_s1_tanh_bwd(const float* x, const float* dpre, const void* pack, float* dx, float* dw, float* db,
                                    void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs,
                                    int accumulate, int dtype, void* stream) {
  return lg_convT_s1_tanh_bwd_m16(x, nullptr, dpre, pack, dx, nullptr, dw, db, workspace, ws_bytes, B, H, W, cb, cs,
                                  accumulate, dtype, stream);
}

// x16 (optional, bf16 path): bf16 mirror of x read by the weight-gradient kernel (x may then be null where
// lg_n3_m16_supported); dx16 (optional): the data gradient is written as bf16 there instead of fp32 to dx
static int s1_tanh_bwd_impl(const float* x, const void* x16, const float* dpre, const void* pack, float* dx, void* dx16,
                            float* dw, float* db, void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs,
                            int accumulate, int dtype, const LgNormFuse* nf, size_t nf_bytes, int* nparts, void* stream);
extern "C" int lg_convT_s1_tanh_bwd_m16(const float* x, const void* x16, const float* dpre, const void* pack, float* dx,
                                        void* dx16, float* dw, float* db, void* workspace, size_t ws_bytes, int B, int H,
                                        int W, int cb, int cs, int accumulate, int dtype, void* stream) {
  return s1_tanh_bwd_impl(x, x16, dpre, pack, dx, dx16, dw, db, workspace, ws_bytes, B, H, W, cb, cs, accumulate, dtype, nullptr, 0,
                          nullptr, stream);
}
// as lg_convT_s1_tanh_bwd_m16 (bf16 data gradient dx16) + the norm-backward sums of dx16 (see lg_conv2d_s2_dgrad_nf)
extern "C" int lg_convT_s1_tanh_bwd_nf(const float* x, const void* x16, const float* dpre, const void* pack, void* dx16, float* dw,
                                       float* db, void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs,
                                       int accumulate, int dtype, const void* z16, const float* stats, float alpha, void* part,
                                       size_t part_bytes, int* nparts, void* stream) {
  LG_CHECK_ARG(nparts, "lg_convT_s1_tanh_bwd_nf: null pointer");
  *nparts = 0;
  LgNormFuse nf{(const __bf16*)z16, stats, (double*)part, alpha, 0};
  return s1_tanh_bwd_impl(x, x16, dpre, pack, nullptr, dx16, dw, db, workspace, ws_bytes, B, H, W, cb, cs, accumulate, dtype,
                          (z16 && stats && part && dx16) ? &nf : nullptr, part_bytes, nparts, stream);
}
static int s1_tanh_bwd_impl(const float* x, const void* x16, const float* dpre, const void* pack, float* dx, void* dx16,
                            float* dw, float* db, void* workspace, size_t ws_bytes, int B, int H, int W, int cb, int cs,
                            int accumulate, int dtype, const LgNormFuse* nf, size_t nf_bytes, int* nparts, void* stream) {
  LG_CHECK_ARG(cb == 3, "lg_convT_s1_tanh_bwd: only image_channel == 3 is supported (got %d)", cb);
  LG_CHECK_ARG(dpre && pack && workspace, "lg_convT_s1_tanh_bwd: null pointer");
  LG_CHECK_ARG(ws_bytes >= lg_convT_s1_bwd_workspace_bytes(B, H, W, cb, cs, dtype),
               "lg_convT_s1_tanh_bwd: workspace too small");
  LG_CHECK_ARG(!(dx && dx16), "lg_convT_s1_tanh_bwd: give dx or dx16, not both");
  int rc;
  if (dx || dx16) {  // dx[i,ci] = sum_k,co dpre[i+k-2,co] W[k,co,ci]  -> patch conv, stride 1, pad 2
    rc = LG_ERR_UNSUPPORTED;
    if (dtype == LG_DT_BF16 && n3_enabled())
      rc = lg_n3_s1_dgrad_p16_nf_try(dpre, raw_pack(pack, cb, cs, dtype), dx, dx16, B, H, W, cs, nf, nf_bytes, nparts, stream);
    if (rc == LG_ERR_UNSUPPORTED) rc = lg_conv_igemm_ex(MODE_PATCH, dtype, dpre, nullptr, pack, nullptr, dx, dx16, B, H, W, 3, cs, 0, 1, 2, nullptr, 0,
                          nullptr, stream);
    if (rc) return rc;
  }
  if (dw) {
    LG_CHECK_ARG(x || x16, "lg_convT_s1_tanh_bwd: x is null but dw requested");
    rc = lg_conv_wgrad_m16(dpre, nullptr, x, dtype == LG_DT_BF16 ? x16 : nullptr, dw, workspace, ws_bytes, B, H, W, 3, cs, 1,
                           2, accumulate, dtype, stream);
    if (rc) return rc;
  }
  if (db) {
    const long long M = (long long)B * H * W;
    LG_CHECK_ARG(M % 4 == 0, "lg_convT_s1_tanh_bwd: B*H*W must be a multiple of 4");
    const long long n12 = M / 4;
    int nb = (int)((n12 + 255) / 256);
    if (nb > 512) nb = 512;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum3_kernel, dim3(nb), dim3(256), 0, st, dpre, (float*)workspace, n12);
    LG_CHECK_LAUNCH("lg_convT_s1_tanh_bwd(bias)");
    hipLaunchKernelGGL(colsum3_final_kernel, dim3(1), dim3(192), 0, st, (const float*)workspace, nb, db, accumulate);
    LG_CHECK_LAUNCH("lg_convT_s1_tanh_bwd(bias final)");
  }
  return LG_OK;
}
