// Dense layers of the LittleGAN step (/root/reference/model.py:62-63 D heads with sigmoid,
// :83 Generator.dense, :120 Adjuster.dense).  All are "skinny" GEMMs (one dimension <= 133 or
// <= 41) and therefore HBM/L2-bound on the 24576-wide operand: plain f32 VALU kernels with 16-B
// coalesced accesses along the wide dimension and the small operand broadcast from LDS.
//   dense_fwd    : y[B,N] = x[B,K] @ w[K,N] + bias          (K small, N wide)
//   dense_wgrad  : dw[K,N] = x^T @ dy, db[N] = colsum(dy)
//   heads_fwd    : p[B,1+c] = sigmoid(x[B,K] @ [wpr | wc] + [bpr | bc])   (K wide)
//   heads_dgrad  : dx[B,K] = dz[B,1+c] @ [wpr | wc]^T
//   heads_wgrad  : dwpr[K], dwc[K,c] = x^T @ dz ; dbpr, dbc = colsum(dz)
#include "lg_common.h"

namespace {

constexpr int HC_MAX = 40;  // max cond_dim (CelebA has 40 attributes)

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
template <int KT>
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ db, int B, int K,
                                                          int N, int accumulate) {
  extern __shared__ float xs[];  // [BT][KT] batch tile
  constexpr int BT = 64;
  const int k0 = blockIdx.y * KT;
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  f32x4 acc[KT];
  f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int bb = 0; bb < B; bb += BT) {
    __syncthreads();
    for (int i = threadIdx.x; i < BT * KT; i += 256) {
      const int b = i / KT, k = i - b * KT;
      xs[i] = (bb + b < B && k0 + k < K) ? x[(long long)(bb + b) * K + k0 + k] : 0.f;
    }
    __syncthreads();
    if (n < N) {
      const int be = min(BT, B - bb);
      for (int b = 0; b < be; ++b) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + (long long)(bb + b) * N + n);
        accb += dv;
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] += xs[b * KT + k] * dv;
      }
    }
  }
  if (n >= N) return;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k0 + k < K) {
      f32x4* o = reinterpret_cast<f32x4*>(dw + (long long)(k0 + k) * N + n);
      *o = accumulate ? (*o + acc[k]) : acc[k];
    }
  }
  if (blockIdx.y == 0 && db) {
    f32x4* o = reinterpret_cast<f32x4*>(db + n);
    *o = accumulate ? (*o + accb) : accb;
  }
}

// ---------------------------------------------------------------- D heads
// p[b][0] = sigmoid(x[b].wpr + bpr) ; p[b][1+j] = sigmoid(x[b].wc[:,j] + bc[j]);  2 samples per block
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wpr,
                                                        const float* __restrict__ bpr, const float* __restrict__ wc,
                                                        const float* __restrict__ bc, float* __restrict__ p, int B,
                                                        int K, int c) {
  const int b0 = blockIdx.x * 2;
  const bool has1 = b0 + 1 < B;
  const float* x0 = x + (long long)b0 * K;
  const float* x1 = x + (long long)(has1 ? b0 + 1 : b0) * K;
  float a0[HC_MAX + 1], a1[HC_MAX + 1];
#pragma unroll
  for (int j = 0; j <= HC_MAX; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v0 = x0[k], v1 = x1[k];
    const float wp = wpr[k];
    a0[0] += v0 * wp; a1[0] += v1 * wp;
    const float* wr = wc + (long long)k * c;
#pragma unroll
    for (int j = 0; j < HC_MAX; ++j) {
      if (j < c) { const float wv = wr[j]; a0[1 + j] += v0 * wv; a1[1 + j] += v1 * wv; }
    }
  }
  __shared__ float sred[4][2][HC_MAX + 1];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j <= HC_MAX; ++j) {
    if (j <= c) {
      const float s0 = lg_wave_sum(a0[j]), s1 = lg_wave_sum(a1[j]);
      if (lane == 0) { sred[wid][0][j] = s0; sred[wid][1][j] = s1; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (c + 1); i += 256) {
    const int s = i / (c + 1), j = i - s * (c + 1);
    if (s == 1 && !has1) continue;
    const float z = sred[0][s][j] + sred[1][s][j] + sred[2][s][j] + sred[3][s][j] + (j == 0 ? bpr[0] : bc[j - 1]);
    p[(long long)(b0 + s) * (c + 1) + j] = 1.f / (1.f + __expf(-z));
  }
}

// dx[b][k] = dz[b][0]*wpr[k] + sum_j dz[b][1+j]*wc[k][j] ; 8 samples per block row
__global__ __launch_bounds__(256) void heads_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wpr,
                                                          const float* __restrict__ wc, float* __restrict__ dx, int B,
                                                          int K, int c) {
  constexpr int TB = 8;
  __shared__ float sdz[TB][HC_MAX + 1];
  const int b0 = blockIdx.y * TB;
  for (int i = threadIdx.x; i < TB * (c + 1); i += 256) {
    const int b = i / (c + 1), j = i - b * (c + 1);
    sdz[b][j] = (b0 + b < B) ? dz[(long long)(b0 + b) * (c + 1) + j] : 0.f;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float acc[TB];
  const float wp = wpr[k];
#pragma unroll
  for (int b = 0; b < TB; ++b) acc[b] = sdz[b][0] * wp;
  const float* wr = wc + (long long)k * c;
  for (int j = 0; j < c; ++j) {
    const float wv = wr[j];
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[b] += sdz[b][1 + j] * wv;
  }
#pragma unroll
  for (int b = 0; b < TB; ++b)
    if (b0 + b < B) dx[(long long)(b0 + b) * K + k] = acc[b];
}

// dwpr[k], dwc[k][j] (+)= sum_b x[b][k]*dz[b][.] ; block 0 also writes dbpr, dbc
__global__ __launch_bounds__(256) void heads_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                          float* __restrict__ dwpr, float* __restrict__ dbpr,
                                                          float* __restrict__ dwc, float* __restrict__ dbc, int B, int K,
                                                          int c, int accumulate) {
  constexpr int BT = 32;
  __shared__ float sdz[BT][HC_MAX + 1];
  const int k = blockIdx.x * 256 + threadIdx.x;
  float acc[HC_MAX + 1];
#pragma unroll
  for (int j = 0; j <= HC_MAX; ++j) acc[j] = 0.f;
  float bsum = 0.f;  // thread j < c+1 of block 0 accumulates db[j]
  for (int bb = 0; bb < B; bb += BT) {
    __syncthreads();
    for (int i = threadIdx.x; i < BT * (c + 1); i += 256) {
      const int b = i / (c + 1), j = i - b * (c + 1);
      sdz[b][j] = (bb + b < B) ? dz[(long long)(bb + b) * (c + 1) + j] : 0.f;
    }
    __syncthreads();
    const int be = min(BT, B - bb);
    if (blockIdx.x == 0 && threadIdx.x <= c)
      for (int b = 0; b < be; ++b) bsum += sdz[b][threadIdx.x];
    if (k < K) {
      for (int b = 0; b < be; ++b) {
        const float xv = x[(long long)(bb + b) * K + k];
#pragma unroll
        for (int j = 0; j <= HC_MAX; ++j)
          if (j <= c) acc[j] += xv * sdz[b][j];
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x <= c) {
    float* o = threadIdx.x == 0 ? dbpr : dbc + (threadIdx.x - 1);
    *o = (accumulate ? *o : 0.f) + bsum;
  }
  if (k >= K) return;
  dwpr[k] = (accumulate ? dwpr[k] : 0.f) + acc[0];
  float* wr = dwc + (long long)k * c;
#pragma unroll
  for (int j = 0; j < HC_MAX; ++j)
    if (j < c) wr[j] = (accumulate ? wr[j] : 0.f) + acc[1 + j];
}

}  // namespace

extern "C" int lg_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                            void* stream) {
  LG_CHECK_ARG(x && w && y, "lg_dense_fwd: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && K <= 1024 && N > 0 && N % 4 == 0, "lg_dense_fwd: bad shape B=%d K=%d N=%d", B, K, N);
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
  constexpr int KT = 16;
  dim3 grid(lg_cdiv(N, 1024), lg_cdiv(K, KT));
  hipLaunchKernelGGL(dense_wgrad_kernel<KT>, grid, dim3(256), 64 * KT * sizeof(float), (hipStream_t)stream, x, dy, dw,
                     db, B, K, N, accumulate);
  LG_CHECK_LAUNCH("lg_dense_wgrad");
  return LG_OK;
}

extern "C" int lg_heads_fwd(const float* x, const float* wpr, const float* bpr, const float* wc, const float* bc,
                            float* p, int B, int K, int c, void* stream) {
  LG_CHECK_ARG(x && wpr && bpr && wc && bc && p, "lg_heads_fwd: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && c >= 1 && c <= HC_MAX, "lg_heads_fwd: bad shape B=%d K=%d c=%d", B, K, c);
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(lg_cdiv(B, 2)), dim3(256), 0, (hipStream_t)stream, x, wpr, bpr, wc, bc, p, B,
                     K, c);
  LG_CHECK_LAUNCH("lg_heads_fwd");
  return LG_OK;
}

extern "C" int lg_heads_dgrad(const float* dz, const float* wpr, const float* wc, float* dx, int B, int K, int c,
                              void* stream) {
  LG_CHECK_ARG(dz && wpr && wc && dx, "lg_heads_dgrad: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && c >= 1 && c <= HC_MAX, "lg_heads_dgrad: bad shape B=%d K=%d c=%d", B, K, c);
  dim3 grid(lg_cdiv(K, 256), lg_cdiv(B, 8));
  hipLaunchKernelGGL(heads_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, dz, wpr, wc, dx, B, K, c);
  LG_CHECK_LAUNCH("lg_heads_dgrad");
  return LG_OK;
}

extern "C" int lg_heads_wgrad(const float* x, const float* dz, float* dwpr, float* dbpr, float* dwc, float* dbc, int B,
                              int K, int c, int accumulate, void* stream) {
  LG_CHECK_ARG(x && dz && dwpr && dbpr && dwc && dbc, "lg_heads_wgrad: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && c >= 1 && c <= HC_MAX, "lg_heads_wgrad: bad shape B=%d K=%d c=%d", B, K, c);
  hipLaunchKernelGGL(heads_wgrad_kernel, dim3(lg_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, x, dz, dwpr, dbpr,
                     dwc, dbc, B, K, c, accumulate);
  LG_CHECK_LAUNCH("lg_heads_wgrad");
  return LG_OK;
}
