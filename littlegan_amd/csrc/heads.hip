// Discriminator heads (/root/reference/model.py:62-63,70-72): Dense(1, sigmoid) and Dense(cond_dim, sigmoid) on the
// flattened last encoder map, fused into one [B, 1+c] output (column 0 = output_pr, 1.. = output_cond).
// K = init_dim^2 * conv_filter[0] (24576 at 128^2) is wide, N = 1+c <= 41 is tiny: the kernels put the N outputs on
// the LANES of a wave (weight rows [k][0..c) are contiguous -> one coalesced 164-B read per k) and read the
// activation values as wave-uniform 16-B loads (4 consecutive k, all lanes same address).
//   heads_fwd   : block = 16 samples x 256 k (split-K over blocks), x tile in LDS; partials + heads_final (bias, sigmoid)
//   heads_dgrad : dx[b][k] = dz[b][:] . [wpr[k] | wc[k][:]]          (thread per k)
//   heads_wgrad : lanes = k, dz rows wave-uniform; block owns 64 k's, its 4 waves split the batch
#include "lg_common.h"

namespace {

constexpr int HC_MAX = 40;  // max cond_dim (CelebA has 40 attributes); 1 + c <= 64 lanes

__device__ __forceinline__ float head_w(const float* __restrict__ wpr, const float* __restrict__ wc, int k, int j, int c) {
  return j == 0 ? wpr[k] : wc[(long long)k * c + (j - 1)];
}

constexpr int FS = 16;    // samples per block (fwd)
constexpr int FKC = 256;  // k per block (fwd): the K sum is split over gridDim.x blocks, partials merged in fixed order

// Block = 16 samples x 256 k.  The activation tile goes through LDS (16 KB, 4 float4 loads in flight per thread) and is
// read back as 4-k ds_read_b128 broadcasts; each wave takes 64 of the k's and streams its weight rows straight from
// global (row [k][0..c) contiguous -> one coalesced read per k, no redundancy inside the block).  part[kc][b][j] is
// reduced over kc by heads_final_kernel.
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wpr,
                                                        const float* __restrict__ wc, float* __restrict__ part, int B,
                                                        int K, int c) {
  __shared__ __attribute__((aligned(16))) float xs[FS][FKC];
  __shared__ float sred[4][FS][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b0 = blockIdx.y * FS, k0 = blockIdx.x * FKC;
  const bool act = lane <= c;
#pragma unroll
  for (int i = 0; i < FS * FKC / 4 / 256; ++i) {
    const int q = threadIdx.x + 256 * i, b = q / (FKC / 4), k4 = (q % (FKC / 4)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (b0 + b < B && k0 + k4 < K) v = *reinterpret_cast<const f32x4*>(x + (long long)(b0 + b) * K + k0 + k4);
    *reinterpret_cast<f32x4*>(&xs[b][k4]) = v;
  }
  __syncthreads();
  float acc[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) acc[s] = 0.f;
  const int kw = wid * 64;
#pragma unroll 2
  for (int kk = 0; kk < 64; kk += 4) {
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + kw + kk + e;
      w[e] = (act && k < K) ? head_w(wpr, wc, k, lane, c) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < FS; ++s) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[s][kw + kk]);
      acc[s] += (xv[0] * w[0] + xv[1] * w[1]) + (xv[2] * w[2] + xv[3] * w[3]);
    }
  }
#pragma unroll
  for (int s = 0; s < FS; ++s) sred[wid][s][lane] = acc[s];
  __syncthreads();
  for (int i = threadIdx.x; i < FS * 64; i += 256) {
    const int s = i >> 6, j = i & 63;
    if (j <= c && b0 + s < B)
      part[((long long)blockIdx.x * B + b0 + s) * (c + 1) + j] = (sred[0][s][j] + sred[1][s][j]) + (sred[2][s][j] + sred[3][s][j]);
  }
}

__global__ __launch_bounds__(256) void heads_final_kernel(const float* __restrict__ part, const float* __restrict__ bpr,
                                                          const float* __restrict__ bc, float* __restrict__ p, int B,
                                                          int c, int nkc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = B * (c + 1);
  if (i >= n) return;
  const int j = i % (c + 1);
  float z = 0.f;
  for (int q = 0; q < nkc; ++q) z += part[(long long)q * n + i];
  z += j == 0 ? bpr[0] : bc[j - 1];
  p[i] = 1.f / (1.f + expf(-z));
}

// dx[b][k] = dz[b][0] wpr[k] + sum_j dz[b][1+j] wc[k][j]: thread per k, DB samples per block.  The weight rows of the
// block's 256 k's are one contiguous 256*c run: staged through LDS with coalesced loads (a thread reading its own row
// straight from global is a 4*c-byte stride across lanes), then read back at an odd row pitch (conflict-free).
constexpr int DB = 16;
__global__ __launch_bounds__(256) void heads_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wpr,
                                                          const float* __restrict__ wc, float* __restrict__ dx, int B,
                                                          int K, int c) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int cp = c | 1;            // odd row pitch: the per-thread row reads are conflict-free
  float* sw = dsm;                 // [256][cp]
  float* sdz = dsm + ((256 * cp + 3) / 4) * 4;  // [c + 1][DB] (transposed: the DB values of one j are 4 ds_read_b128)
  const int b0 = blockIdx.y * DB, k0 = blockIdx.x * 256;
  const int kn = min(256, K - k0);
  {  // flat coalesced copy; (row, col) of element i advance incrementally (no division per element)
    const int dq = 256 / c, dr = 256 - dq * c;
    int row = threadIdx.x / c, col = threadIdx.x - row * c;
    for (int i = threadIdx.x; i < kn * c; i += 256) {
      sw[row * cp + col] = wc[(long long)k0 * c + i];
      row += dq; col += dr;
      if (col >= c) { col -= c; ++row; }
    }
  }
  for (int i = threadIdx.x; i < DB * (c + 1); i += 256) {
    const int b = i / (c + 1), j = i - b * (c + 1);
    sdz[j * DB + b] = (b0 + b < B) ? dz[(long long)(b0 + b) * (c + 1) + j] : 0.f;
  }
  __syncthreads();
  const int k = k0 + threadIdx.x;
  if (k >= K) return;
  float acc[DB];
  const float wp = wpr[k];
#pragma unroll
  for (int q = 0; q < DB / 4; ++q) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(sdz + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[q * 4 + e] = d[e] * wp;
  }
  const float* wr = sw + threadIdx.x * cp;
  for (int j = 0; j < c; ++j) {
    const float wv = wr[j];
#pragma unroll
    for (int q = 0; q < DB / 4; ++q) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(sdz + (1 + j) * DB + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q * 4 + e] += d[e] * wv;
    }
  }
#pragma unroll
  for (int b = 0; b < DB; ++b)
    if (b0 + b < B) dx[(long long)(b0 + b) * K + k] = acc[b];
}

constexpr int WK = 64;  // k per block (wgrad)

// dW[k][j] = sum_b x[b][k] dz[b][j]: lane = k (x streamed once, coalesced); the 1+c gradients of sample b are one
// coalesced 164-B read (lane j) broadcast with v_readlane, acc[j] in registers.  Block owns 64 k's, its 4 waves
// split the batch and merge in wave order through LDS (deterministic).
__global__ __launch_bounds__(256) void heads_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                          float* __restrict__ dwpr, float* __restrict__ dbpr,
                                                          float* __restrict__ dwc, float* __restrict__ dbc, int B, int K,
                                                          int c, int accumulate) {
  constexpr int J = HC_MAX + 1;
  __shared__ float red[J + 1][WK + 1];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k0 = blockIdx.x * WK, k = k0 + lane;
  const bool kok = k < K;
  const int bq = (B + 3) / 4, bs = wid * bq, be = min(B, bs + bq);
  const bool do_bias = blockIdx.x == 0;
  float acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = 0.f;
  float bsum = 0.f;
  constexpr int U = 8;  // samples per trip: U x-loads + U dz-loads in flight (manual unroll: readlane blocks the pragma)
  for (int b = bs; b < be; b += U) {
    float xv[U], dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = b + u < be;
      xv[u] = (ok && kok) ? x[(long long)(b + u) * K + k] : 0.f;
      dv[u] = (ok && lane <= c) ? dz[(long long)(b + u) * (c + 1) + lane] : 0.f;  // lane j holds dz[b][j] (0 beyond c)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bsum += dv[u];
#pragma unroll
      for (int j = 0; j < J; ++j)
        acc[j] += xv[u] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv[u]), j));
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int j = 0; j < J; ++j) red[j][lane] = (w == 0 ? 0.f : red[j][lane]) + acc[j];
      red[J][lane] = (w == 0 ? 0.f : red[J][lane]) + bsum;
    }
    __syncthreads();
  }
  if (do_bias && threadIdx.x <= c) {
    float* o = threadIdx.x == 0 ? dbpr : dbc + (threadIdx.x - 1);
    *o = (accumulate ? *o : 0.f) + red[J][threadIdx.x];
  }
  for (int i = threadIdx.x; i < WK * (c + 1); i += 256) {  // consecutive i -> consecutive dwc addresses
    const int kk = i / (c + 1), j = i - kk * (c + 1);
    if (k0 + kk < K) {
      float* o = j == 0 ? dwpr + (k0 + kk) : dwc + (long long)(k0 + kk) * c + (j - 1);
      *o = (accumulate ? *o : 0.f) + red[j][kk];
    }
  }
}

}  // namespace

extern "C" int lg_heads_fwd_mfma_try(const float* x, const float* wpr, const float* wc, float* part, int B, int K, int c,
                                     int* nkc_out, void* stream);
extern "C" int lg_heads_wgrad_mfma_try(const float* x, const float* dz, float* dwpr, float* dbpr, float* dwc, float* dbc, int B,
                                       int K, int c, int accumulate, void* stream);
extern "C" int lg_heads_dgrad_mfma_try(const float* dz, const float* wpr, const float* wc, float* dx, int B, int K, int c,
                                       void* stream);

extern "C" size_t lg_heads_fwd_workspace_bytes(int B, int K, int c) {
  return (size_t)lg_cdiv(K, FKC) * (size_t)B * (size_t)(c + 1) * sizeof(float);
}

extern "C" int lg_heads_fwd(const float* x, const float* wpr, const float* bpr, const float* wc, const float* bc,
                            float* p, void* workspace, size_t ws_bytes, int B, int K, int c, void* stream) {
  LG_CHECK_ARG(x && wpr && bpr && wc && bc && p && workspace, "lg_heads_fwd: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && K % 4 == 0 && c >= 1 && c <= HC_MAX, "lg_heads_fwd: bad shape B=%d K=%d c=%d", B, K, c);
  LG_CHECK_ARG(ws_bytes >= lg_heads_fwd_workspace_bytes(B, K, c), "lg_heads_fwd: workspace too small (%zu bytes)", ws_bytes);
  int nkc = lg_cdiv(K, FKC);
  float* part = (float*)workspace;
  const int rc = lg_heads_fwd_mfma_try(x, wpr, wc, part, B, K, c, &nkc, stream);  // aligned shapes: fp32 matrix instruction
  if (rc != LG_OK && rc != LG_ERR_UNSUPPORTED) return rc;
  if (rc == LG_ERR_UNSUPPORTED) {
    hipLaunchKernelGGL(heads_fwd_kernel, dim3(nkc, lg_cdiv(B, FS)), dim3(256), 0, (hipStream_t)stream, x, wpr, wc, part,
                       B, K, c);
    LG_CHECK_LAUNCH("lg_heads_fwd");
  }
  hipLaunchKernelGGL(heads_final_kernel, dim3(lg_cdiv(B * (c + 1), 256)), dim3(256), 0, (hipStream_t)stream, part, bpr,
                     bc, p, B, c, nkc);
  LG_CHECK_LAUNCH("lg_heads_fwd(final)");
  return LG_OK;
}

extern "C" int lg_heads_dgrad(const float* dz, const float* wpr, const float* wc, float* dx, int B, int K, int c,
                              void* stream) {
  LG_CHECK_ARG(dz && wpr && wc && dx, "lg_heads_dgrad: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && c >= 1 && c <= HC_MAX, "lg_heads_dgrad: bad shape B=%d K=%d c=%d", B, K, c);
  {
    const int rc = lg_heads_dgrad_mfma_try(dz, wpr, wc, dx, B, K, c, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  dim3 grid(lg_cdiv(K, 256), lg_cdiv(B, DB));
  const size_t lds = (size_t)((256 * (c | 1) + 3) / 4 * 4 + DB * (c + 1)) * sizeof(float);
  hipLaunchKernelGGL(heads_dgrad_kernel, grid, dim3(256), lds, (hipStream_t)stream, dz, wpr, wc, dx, B, K, c);
  LG_CHECK_LAUNCH("lg_heads_dgrad");
  return LG_OK;
}

extern "C" int lg_heads_wgrad(const float* x, const float* dz, float* dwpr, float* dbpr, float* dwc, float* dbc, int B,
                              int K, int c, int accumulate, void* stream) {
  LG_CHECK_ARG(x && dz && dwpr && dbpr && dwc && dbc, "lg_heads_wgrad: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && K % 4 == 0 && c >= 1 && c <= HC_MAX, "lg_heads_wgrad: bad shape B=%d K=%d c=%d", B, K, c);
  {
    const int rc = lg_heads_wgrad_mfma_try(x, dz, dwpr, dbpr, dwc, dbc, B, K, c, accumulate, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  hipLaunchKernelGGL(heads_wgrad_kernel, dim3(lg_cdiv(K, WK)), dim3(256), 0, (hipStream_t)stream, x, dz, dwpr, dbpr,
                     dwc, dbc, B, K, c, accumulate);
  LG_CHECK_LAUNCH("lg_heads_wgrad");
  return LG_OK;
}
