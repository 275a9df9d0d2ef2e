// Input side of the training step, on the device (/root/reference/eager_trainer.py:125-131):
//   noise     = tf.random.normal([B, noise_dim])
//   new_image = random_flip_left_right -> random_brightness(0.02) -> random_contrast(0.75, 1.003)
//               -> random_hue(0.03) -> + 0.1 * tf.random.normal(shape, 0, 0.2)
// TensorFlow's random streams cannot be reproduced (and parity treats these tensors as step INPUTS), so
// the draws come from a counter-based generator — Philox4x32-10, the stateless generator of Random123 / cuRAND / torch —
// keyed by (seed, offset): any element of any step can be regenerated independently, on any rank, in any order.
// The deterministic part of the transform (given flip mask, brightness delta, contrast factor, hue delta) follows the
// TF-1.15 image ops: brightness adds the delta; contrast scales about the per-(image, channel) mean over H x W; hue is
// the fused AdjustHue op — a rotation of the hue angle that keeps each pixel's min and max channel values, defined for
// any value range (the images here live in [-1, 1]).
#include "lg_common.h"

namespace {

struct u4 { unsigned x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, unsigned k0, unsigned k1) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const unsigned hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = u4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += W0; k1 += W1;
  }
  return c;
}

// 32 random bits -> (0, 1]  (never 0: safe for the log of Box-Muller), 24-bit resolution like cuRAND's uniform
__device__ __forceinline__ float u01(unsigned b) { return ((float)(b >> 8) + 1.0f) * (1.0f / 16777216.0f); }

// 4 standard normals from one Philox block (two Box-Muller pairs)
__device__ __forceinline__ void normal4(unsigned long long seed, unsigned long long ctr, float (&z)[4]) {
  const u4 r = philox4x32_10(u4{(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u}, (unsigned)seed, (unsigned)(seed >> 32));
  const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
  float s0, c0, s1, c1;
  sincosf(6.28318530717958647692f * u01(r.y), &s0, &c0);
  sincosf(6.28318530717958647692f * u01(r.w), &s1, &c1);
  z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

// out[i] = mean + std * N(0,1); element i uses normal (i & 3) of Philox block (offset + i / 4)
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long long n, float mean, float stdv,
                                                    unsigned long long seed, unsigned long long offset) {
  const long long nblk = (n + 3) / 4, stride = (long long)gridDim.x * blockDim.x;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
    float z[4];
    normal4(seed, offset + (unsigned long long)b, z);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (b * 4 + k < n) out[b * 4 + k] = mean + stdv * z[k];
  }
}

// raw generator output (known-answer tests): out[4 i .. 4 i + 3] = philox(counter = offset + i, key = seed)
__global__ void philox_kernel(unsigned* __restrict__ out, int nblk, unsigned long long seed, unsigned long long offset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  const unsigned long long c = offset + (unsigned long long)i;
  const u4 r = philox4x32_10(u4{(unsigned)c, (unsigned)(c >> 32), 0u, 0u}, (unsigned)seed, (unsigned)(seed >> 32));
  out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// means[b][c] = mean over H x W of img[b][..][c]   (3 channels; one block per image, fp64 merge)
__global__ __launch_bounds__(256) void chan_mean3_kernel(const float* __restrict__ img, float* __restrict__ means,
                                                         int HW) {
  const float* p = img + (long long)blockIdx.x * HW * 3;
  float s[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < HW; i += 256) { s[0] += p[i * 3]; s[1] += p[i * 3 + 1]; s[2] += p[i * 3 + 2]; }
  __shared__ double sred[48];
  double d[3] = {(double)s[0], (double)s[1], (double)s[2]};
  lg_block_sum_d<3>(d, sred);
  if (threadIdx.x == 0)
    for (int c = 0; c < 3; ++c) means[blockIdx.x * 3 + c] = (float)(d[c] / (double)HW);
}

// hue rotation by dh (fraction of a turn) that keeps the pixel's min and max channel values
__device__ __forceinline__ void hue_rotate(float& r, float& g, float& b, float dh) {
  const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b)), range = vmax - vmin;
  if (!(range > 0.f)) return;  // grey: hue undefined, unchanged
  float h;  // hue in sixths of a turn, [0, 6)
  if (r == vmax) h = (g - b) / range;
  else if (g == vmax) h = 2.f + (b - r) / range;
  else h = 4.f + (r - g) / range;
  h += 6.f * dh;
  h -= 6.f * floorf(h * (1.f / 6.f));
  if (h >= 6.f) h = 0.f;
  const int sect = (int)h;
  const float f = h - (float)sect;
  const float up = vmin + range * f, dn = vmax - range * f;  // rising / falling edge inside the sector
  switch (sect) {
    case 0: r = vmax; g = up; b = vmin; break;
    case 1: r = dn; g = vmax; b = vmin; break;
    case 2: r = vmin; g = vmax; b = up; break;
    case 3: r = vmin; g = dn; b = vmax; break;
    case 4: r = up; g = vmin; b = vmax; break;
    default: r = vmax; g = vmin; b = dn; break;
  }
}

// one thread per pixel (3 channels)
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                      const float* __restrict__ means, const unsigned char* __restrict__ flip,
                                                      int B, int H, int W, float db, float cf, float dh, float nscale,
                                                      unsigned long long seed, unsigned long long offset,
                                                      const float* __restrict__ dparams) {
  if (dparams) { db = dparams[0]; cf = dparams[1]; dh = dparams[2]; }  // draws made on the device (draws_kernel)
  const long long npix = (long long)B * H * W, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int x = (int)(i % W);
    const long long row = i / W;
    const int n = (int)(row / H);
    const int sx = (flip && flip[n]) ? W - 1 - x : x;
    const float* p = img + (row * W + sx) * 3;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float m = means[n * 3 + k] + db;  // the mean is taken after the brightness shift (flips do not move it)
      c[k] = ((p[k] + db) - m) * cf + m;
    }
    if (dh != 0.f) hue_rotate(c[0], c[1], c[2], dh);
    if (nscale != 0.f) {
      float z[4];
      normal4(seed, offset + (unsigned long long)i, z);
      c[0] += nscale * z[0]; c[1] += nscale * z[1]; c[2] += nscale * z[2];
    }
    float* o = out + i * 3;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
  }
}

// The scalar draws of the TF ops (eager_trainer.py:127-130: one brightness delta, one contrast factor, one hue delta per
// batch; one coin per image for the flip) from the Philox window at `offset`: word w of the window is 24-bit uniform
// u_w = (bits >> 8) / 2^24;  u_0 -> brightness, u_1 -> contrast, u_2 -> hue, u_{3+n} -> flip of image n.
// params[0..2] = {db, cf, dh}; flip[n] = u_{3+n} < 0.5.  No host round trip: the step has no sync on its input side.
__global__ __launch_bounds__(256) void draws_kernel(float* __restrict__ params, unsigned char* __restrict__ flip, int B,
                                                    float db_max, float c_lo, float c_hi, float dh_max,
                                                    unsigned long long seed, unsigned long long offset) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;  // word index of the window
  if (w >= B + 3) return;
  const unsigned long long c = offset + (unsigned long long)(w >> 2);
  const u4 r = philox4x32_10(u4{(unsigned)c, (unsigned)(c >> 32), 0u, 0u}, (unsigned)seed, (unsigned)(seed >> 32));
  const unsigned bits = (w & 3) == 0 ? r.x : (w & 3) == 1 ? r.y : (w & 3) == 2 ? r.z : r.w;
  const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);
  if (w == 0) params[0] = (2.0f * u - 1.0f) * db_max;
  else if (w == 1) params[1] = c_lo + u * (c_hi - c_lo);
  else if (w == 2) params[2] = (2.0f * u - 1.0f) * dh_max;
  else flip[w - 3] = u < 0.5f ? 1 : 0;
}

inline int grid_for(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

}  // namespace

extern "C" int lg_philox4x32(unsigned* out, int nblocks, unsigned long long seed, unsigned long long offset, void* stream) {
  LG_CHECK_ARG(out && nblocks > 0, "lg_philox4x32: bad arguments");
  hipLaunchKernelGGL(philox_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, nblocks, seed, offset);
  LG_CHECK_LAUNCH("lg_philox4x32");
  return LG_OK;
}

extern "C" int lg_randn(float* out, long long n, float mean, float stdv, unsigned long long seed,
                        unsigned long long offset, void* stream) {
  LG_CHECK_ARG(out && n > 0, "lg_randn: bad arguments");
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, mean, stdv, seed,
                     offset);
  LG_CHECK_LAUNCH("lg_randn");
  return LG_OK;
}

extern "C" size_t lg_augment_workspace_bytes(int B) { return (size_t)B * 3 * sizeof(float); }

// out = [+ noise_scale * N(0,1)] hue(dh)( contrast(cf)( flip?(img) + db ) );  img, out [B,H,W,3] (out != img);
// flip [B] bytes (device) or null; the noise of pixel i is Philox block (offset + i) under `seed`
extern "C" int lg_augment(const float* img, float* out, int B, int H, int W, const unsigned char* flip, float db, float cf,
                          float dh, float noise_scale, unsigned long long seed, unsigned long long offset,
                          void* workspace, size_t ws_bytes, void* stream) {
  LG_CHECK_ARG(img && out && img != out && workspace, "lg_augment: null pointer (or in-place call)");
  LG_CHECK_ARG(B > 0 && H > 0 && W > 0, "lg_augment: bad shape B=%d H=%d W=%d", B, H, W);
  LG_CHECK_ARG(ws_bytes >= lg_augment_workspace_bytes(B), "lg_augment: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* means = (float*)workspace;
  hipLaunchKernelGGL(chan_mean3_kernel, dim3(B), dim3(256), 0, st, img, means, H * W);
  LG_CHECK_LAUNCH("lg_augment(mean)");
  hipLaunchKernelGGL(augment_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, st, img, out, (const float*)means,
                     flip, B, H, W, db, cf, dh, noise_scale, seed, offset, (const float*)nullptr);
  LG_CHECK_LAUNCH("lg_augment");
  return LG_OK;
}

extern "C" size_t lg_augment_drawn_workspace_bytes(int B) {
  return ((size_t)B * 3 * sizeof(float) + 15) / 16 * 16 + 16 + ((size_t)B + 15) / 16 * 16;
}

// lg_augment with the random draws of eager_trainer.py:127-130 made ON THE DEVICE from the Philox window at draw_offset
// (see draws_kernel): flip per image with probability 1/2, brightness delta U(-db_max, db_max), contrast factor
// U(c_lo, c_hi), hue delta U(-dh_max, dh_max); the pixel noise uses the window at noise_offset as in lg_augment.
// The whole input side of the step is then enqueued without a host synchronisation.
extern "C" int lg_augment_drawn(const float* img, float* out, int B, int H, int W, float db_max, float c_lo, float c_hi,
                                float dh_max, float noise_scale, unsigned long long seed, unsigned long long draw_offset,
                                unsigned long long noise_offset, void* workspace, size_t ws_bytes, void* stream) {
  LG_CHECK_ARG(img && out && img != out && workspace, "lg_augment_drawn: null pointer (or in-place call)");
  LG_CHECK_ARG(B > 0 && H > 0 && W > 0, "lg_augment_drawn: bad shape B=%d H=%d W=%d", B, H, W);
  LG_CHECK_ARG(ws_bytes >= lg_augment_drawn_workspace_bytes(B), "lg_augment_drawn: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* means = (float*)workspace;
  float* params = (float*)((char*)workspace + ((size_t)B * 3 * sizeof(float) + 15) / 16 * 16);
  unsigned char* flip = (unsigned char*)(params + 4);
  hipLaunchKernelGGL(draws_kernel, dim3((B + 3 + 255) / 256), dim3(256), 0, st, params, flip, B, db_max, c_lo, c_hi, dh_max,
                     seed, draw_offset);
  LG_CHECK_LAUNCH("lg_augment_drawn(draws)");
  hipLaunchKernelGGL(chan_mean3_kernel, dim3(B), dim3(256), 0, st, img, means, H * W);
  LG_CHECK_LAUNCH("lg_augment_drawn(mean)");
  hipLaunchKernelGGL(augment_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, st, img, out, (const float*)means,
                     (const unsigned char*)flip, B, H, W, 0.f, 1.f, 1.f, noise_scale, seed, noise_offset, (const float*)params);
  LG_CHECK_LAUNCH("lg_augment_drawn");
  return LG_OK;
}
