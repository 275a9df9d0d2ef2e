// Dense layers of the LittleGAN step (/root/reference/model.py:62-63 D heads with sigmoid,
// :83 Generator.dense, :120 Adjuster.dense).  All are "skinny" GEMMs (one dimension <= 133 or
// <= 41) and therefore HBM/L2-bound on the 24576-wide operand: plain f32 VALU kernels with 16-B
// coalesced accesses along the wide dimension and the small operand broadcast from LDS.
//   dense_fwd    : y[B,N] = x[B,K] @ w[K,N] + bias          (K small, N wide)
//   dense_wgrad  : dw[K,N] = x^T @ dy, db[N] = colsum(dy)
// (the D heads live in heads.hip)
#include "lg_common.h"

namespace {

// ---------------------------------------------------------------- dense fwd (small K)
template <int TB>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B,
                                                        int K, int N) {
  extern __shared__ float xs[];  // [TB][K]
  const int b0 = blockIdx.y * TB;
  for (int i = threadIdx.x; i < TB * K; i += 256) {
    const int b = i / K, k = i - b * K;
    xs[i] = (b0 + b < B) ? x[(long long)(b0 + b) * K + k] : 0.f;
  }
  __syncthreads();
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (n >= N) return;
  f32x4 acc[TB];
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < TB; ++b) acc[b] = bv;
  for (int k = 0; k < K; ++k) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long long)k * N + n);
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[b] += xs[b * K + k] * wv;
  }
#pragma unroll
  for (int b = 0; b < TB; ++b)
    if (b0 + b < B) *reinterpret_cast<f32x4*>(y + (long long)(b0 + b) * N + n) = acc[b];
}

// ---------------------------------------------------------------- dense wgrad (small K)
// dw[k][n] = sum_b x[b][k] dy[b][n], db[n] = sum_b dy[b][n].  Block = 256 columns (a lane owns 4) x KT rows of dw; its
// 4 waves split the BATCH (the per-thread chain of dependent 8-load rounds over b is what bounds this kernel, not
// bandwidth) and merge in wave order through LDS (deterministic).
template <int KT>
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ db, int B, int K,
                                                          int N, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                       // [B][KT] the K-slice of x for the whole batch
  f32x4* red = reinterpret_cast<f32x4*>(sm + ((B * KT + 3) / 4) * 4);  // [KT + 1][64] merge buffer
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k0 = blockIdx.y * KT;
  const int n = (blockIdx.x * 64 + lane) * 4;
  for (int i = threadIdx.x; i < B * KT; i += 256) {
    const int b = i / KT, k = i - b * KT;
    xs[i] = (k0 + k < K) ? x[(long long)b * K + k0 + k] : 0.f;
  }
  __syncthreads();
  f32x4 acc[KT];
  f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int bq = (B + 3) / 4, bs = wid * bq, be = min(B, bs + bq);
  if (n < N) {
    int b = bs;
    for (; b + 8 <= be; b += 8) {  // 8 independent 16-B loads in flight per thread
      f32x4 dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dv[u] = *reinterpret_cast<const f32x4*>(dy + (long long)(b + u) * N + n);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        accb += dv[u];
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] += xs[(b + u) * KT + k] * dv[u];
      }
    }
    for (; b < be; ++b) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + (long long)b * N + n);
      accb += dv;
#pragma unroll
      for (int k = 0; k < KT; ++k) acc[k] += xs[b * KT + k] * dv;
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int k = 0; k < KT; ++k) red[k * 64 + lane] = (w == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : red[k * 64 + lane]) + acc[k];
      red[KT * 64 + lane] = (w == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : red[KT * 64 + lane]) + accb;
    }
    __syncthreads();
  }
  if (n >= N) return;
  for (int k = wid; k < KT; k += 4) {
    if (k0 + k < K) {
      f32x4* o = reinterpret_cast<f32x4*>(dw + (long long)(k0 + k) * N + n);
      *o = accumulate ? (*o + red[k * 64 + lane]) : red[k * 64 + lane];
    }
  }
  if (blockIdx.y == 0 && db && wid == 0) {
    f32x4* o = reinterpret_cast<f32x4*>(db + n);
    *o = accumulate ? (*o + red[KT * 64 + lane]) : red[KT * 64 + lane];
  }
}

// ---------------------------------------------------------------- dense dgrad (small K)
// dx[b][k] = sum_n dy[b][n] w[k][n]: one wave per (b, k) dot product of length N (16-B loads), 4 k's per block.
// No tape of the training step needs it (the dense inputs are noise / conditions); part of the boundary for completeness.
__global__ __launch_bounds__(256) void dense_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, int K, int N) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.y, k = blockIdx.x * 4 + wid;
  if (k >= K) return;
  const float* dr = dy + (long long)b * N;
  const float* wr = w + (long long)k * N;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int n = lane * 4; n < N; n += 256) acc += *reinterpret_cast<const f32x4*>(dr + n) * *reinterpret_cast<const f32x4*>(wr + n);
  const float s = lg_wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
  if (lane == 0) dx[(long long)b * K + k] = s;
}

// ---------------------------------------------------------------- dense inputs
// out[b] = [a[b] | c[b]]: the Generator's dense input `concat([noise, cond], -1)` (model.py:97-98)
__global__ __launch_bounds__(256) void concat_cols_kernel(const float* __restrict__ a, int ka, const float* __restrict__ c, int kc,
                                                          float* __restrict__ out, long long total) {
  const int k = ka + kc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / k;
    const int j = (int)(i - b * k);
    out[i] = j < ka ? a[b * ka + j] : c[b * kc + (j - ka)];
  }
}

// t = [first ; second] (rows), u = (t + 1) * 0.5: the Adjuster's target and input conditions (eager_trainer.py:153-154)
__global__ __launch_bounds__(256) void adj_conditions_kernel(const float* __restrict__ first, const float* __restrict__ second,
                                                             float* __restrict__ t, float* __restrict__ u, long long half) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < 2 * half; i += (long long)gridDim.x * 256) {
    const float v = i < half ? first[i] : second[i - half];
    t[i] = v;
    u[i] = (v + 1.0f) * 0.5f;
  }
}

}  // namespace

extern "C" int lg_concat_cols(const float* a, int ka, const float* c, int kc, float* out, int B, void* stream) {
  LG_CHECK_ARG(a && c && out && B > 0 && ka > 0 && kc > 0, "lg_concat_cols: bad arguments B=%d ka=%d kc=%d", B, ka, kc);
  const long long total = (long long)B * (ka + kc);
  const long long nb = (total + 255) / 256;
  hipLaunchKernelGGL(concat_cols_kernel, dim3((int)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream, a, ka, c, kc, out, total);
  LG_CHECK_LAUNCH("lg_concat_cols");
  return LG_OK;
}

extern "C" int lg_adj_conditions(const float* first, const float* second, float* t, float* u, int B, int c, void* stream) {
  LG_CHECK_ARG(first && second && t && u && B > 0 && c > 0, "lg_adj_conditions: bad arguments B=%d c=%d", B, c);
  const long long half = (long long)B * c;
  const long long nb = (2 * half + 255) / 256;
  hipLaunchKernelGGL(adj_conditions_kernel, dim3((int)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream, first, second, t, u, half);
  LG_CHECK_LAUNCH("lg_adj_conditions");
  return LG_OK;
}

extern "C" int lg_dense_fwd_mfma_try(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                                     void* stream);
extern "C" int lg_dense_wgrad_mfma_try(const float* x, const float* dy, float* dw, float* db, int B, int K, int N, int accumulate,
                                       void* stream);

extern "C" int lg_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                            void* stream) {
  LG_CHECK_ARG(x && w && y, "lg_dense_fwd: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && K <= 1024 && N > 0 && N % 4 == 0, "lg_dense_fwd: bad shape B=%d K=%d N=%d", B, K, N);
  {
    const int rc = lg_dense_fwd_mfma_try(x, w, bias, y, B, K, N, stream);  // aligned shapes: fp32 matrix instruction
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  constexpr int TB = 8;
  dim3 grid(lg_cdiv(N, 1024), lg_cdiv(B, TB));
  hipLaunchKernelGGL(dense_fwd_kernel<TB>, grid, dim3(256), TB * K * sizeof(float), (hipStream_t)stream, x, w, bias, y,
                     B, K, N);
  LG_CHECK_LAUNCH("lg_dense_fwd");
  return LG_OK;
}

extern "C" int lg_dense_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int K, int N,
                              int accumulate, void* stream) {
  LG_CHECK_ARG(x && dy && dw, "lg_dense_wgrad: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && N > 0 && N % 4 == 0, "lg_dense_wgrad: bad shape B=%d K=%d N=%d", B, K, N);
  {
    const int rc = lg_dense_wgrad_mfma_try(x, dy, dw, db, B, K, N, accumulate, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  constexpr int KT = 16;
  const size_t lds = ((size_t)((B * KT + 3) / 4) * 4 + (size_t)(KT + 1) * 64 * 4) * sizeof(float);
  LG_CHECK_ARG(lds <= 160 * 1024, "lg_dense_wgrad: batch %d too large for the LDS staging of x", B);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wgrad_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(lg_cdiv(N, 256), lg_cdiv(K, KT));
  hipLaunchKernelGGL(dense_wgrad_kernel<KT>, grid, dim3(256), lds, (hipStream_t)stream, x, dy, dw, db, B, K, N, accumulate);
  LG_CHECK_LAUNCH("lg_dense_wgrad");
  return LG_OK;
}

extern "C" int lg_dense_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, void* stream) {
  LG_CHECK_ARG(dy && w && dx, "lg_dense_dgrad: null pointer");
  LG_CHECK_ARG(B > 0 && B <= 65535 && K > 0 && N > 0 && N % 4 == 0, "lg_dense_dgrad: bad shape B=%d K=%d N=%d", B, K, N);
  hipLaunchKernelGGL(dense_dgrad_kernel, dim3(lg_cdiv(K, 4), B), dim3(256), 0, (hipStream_t)stream, dy, w, dx, K, N);
  LG_CHECK_LAUNCH("lg_dense_dgrad");
  return LG_OK;
}
