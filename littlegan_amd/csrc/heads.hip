// Discriminator heads (/root/reference/model.py:62-63,70-72): Dense(1, sigmoid) and Dense(cond_dim, sigmoid) on the
// flattened last encoder map, fused into one [B, 1+c] output (column 0 = output_pr, 1.. = output_cond).
// K = init_dim^2 * conv_filter[0] (24576 at 128^2) is wide, N = 1+c <= 41 is tiny: the kernels put the N outputs on
// the LANES of a wave (weight rows [k][0..c) are contiguous -> one coalesced 164-B read per k) and broadcast the
// activation values from LDS (ds_read_b128 of 4 consecutive k, all lanes same address).
//   heads_fwd   : 4 samples per block, the 4 waves split K, in-block reduction, sigmoid
//   heads_dgrad : dx[b][k] = dz[b][:] . [wpr[k] | wc[k][:]]          (thread per k)
//   heads_wgrad : each wave owns 64 k's and sweeps the whole batch; x tile [16 b][256 k] through LDS
#include "lg_common.h"

namespace {

constexpr int HC_MAX = 40;  // max cond_dim (CelebA has 40 attributes); 1 + c <= 64 lanes

__device__ __forceinline__ float head_w(const float* __restrict__ wpr, const float* __restrict__ wc, int k, int j, int c) {
  return j == 0 ? wpr[k] : wc[(long long)k * c + (j - 1)];
}

constexpr int FB = 4;     // samples per block (fwd)
constexpr int KCH = 256;  // k per LDS chunk

// Both operands of a k-chunk go through LDS with wide coalesced loads: the weight rows wc[k0..k0+255][0..c) are one
// contiguous 40-KB run (read with float4), wpr[k0..] a 1-KB run; lane j then reads sw[k][j] (consecutive lanes ->
// consecutive banks) and the activation as a 4-k ds_read_b128 broadcast.
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wpr,
                                                        const float* __restrict__ bpr, const float* __restrict__ wc,
                                                        const float* __restrict__ bc, float* __restrict__ p, int B,
                                                        int K, int c) {
  extern __shared__ __attribute__((aligned(16))) float hsm[];
  float* xs = hsm;                 // [FB][KCH]
  float* swc = hsm + FB * KCH;     // [KCH][c]  (row k, c floats)
  float* swp = swc + KCH * c;      // [KCH]
  __shared__ float sred[4][FB][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b0 = blockIdx.x * FB;
  const bool act = lane <= c;
  float acc[FB];
#pragma unroll
  for (int b = 0; b < FB; ++b) acc[b] = 0.f;
  const bool vec_ok = (c % 4 == 0) && ((reinterpret_cast<size_t>(wc) & 15) == 0);
  for (int k0 = 0; k0 < K; k0 += KCH) {
    const int kn = min(KCH, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < FB * KCH / 4; i += 256) {
      const int b = i / (KCH / 4), k4 = (i % (KCH / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b0 + b < B && k0 + k4 < K) v = *reinterpret_cast<const f32x4*>(x + (long long)(b0 + b) * K + k0 + k4);
      *reinterpret_cast<f32x4*>(xs + b * KCH + k4) = v;
    }
    if (vec_ok) {  // 256*c/4 <= 2560 float4: all (up to 10 per thread) in flight before the first LDS store
      const f32x4* g = reinterpret_cast<const f32x4*>(wc + (long long)k0 * c);
      const int n4 = kn * c / 4;
      f32x4 v[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int i = u * 256 + threadIdx.x;
        v[u] = i < n4 ? g[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int i = u * 256 + threadIdx.x;
        if (i < n4) reinterpret_cast<f32x4*>(swc)[i] = v[u];
      }
    } else {
      for (int i = threadIdx.x; i < kn * c; i += 256) swc[i] = wc[(long long)k0 * c + i];
    }
    for (int i = threadIdx.x; i < KCH; i += 256) swp[i] = i < kn ? wpr[k0 + i] : 0.f;
    if (kn < KCH) for (int i = kn * c + threadIdx.x; i < KCH * c; i += 256) swc[i] = 0.f;
    __syncthreads();
    const int kw = wid * 64;  // this wave's 64 k of the chunk
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 4) {
      const int k = kw + kk;
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = !act ? 0.f : (lane == 0 ? swp[k + e] : swc[(k + e) * c + lane - 1]);
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + b * KCH + k);
        acc[b] += (xv[0] * w[0] + xv[1] * w[1]) + (xv[2] * w[2] + xv[3] * w[3]);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < FB; ++b) sred[wid][b][lane] = acc[b];
  __syncthreads();
  if (wid == 0 && act) {
    const float bias = lane == 0 ? bpr[0] : bc[lane - 1];
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      if (b0 + b < B) {
        const float z = ((sred[0][b][lane] + sred[1][b][lane]) + (sred[2][b][lane] + sred[3][b][lane])) + bias;
        p[(long long)(b0 + b) * (c + 1) + lane] = 1.f / (1.f + expf(-z));
      }
    }
  }
}

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

constexpr int WB = 16;  // samples per LDS tile (wgrad)

__global__ __launch_bounds__(256) void heads_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                          float* __restrict__ dwpr, float* __restrict__ dbpr,
                                                          float* __restrict__ dwc, float* __restrict__ dbc, int B, int K,
                                                          int c, int accumulate) {
  __shared__ __attribute__((aligned(16))) float xs[WB][KCH];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k0 = blockIdx.x * KCH, kw = wid * 64;
  const bool act = lane <= c;
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  for (int bb = 0; bb < B; bb += WB) {
    __syncthreads();
    for (int i = threadIdx.x; i < WB * KCH; i += 256) {
      const int b = i / KCH, kk = i - b * KCH;
      xs[b][kk] = (bb + b < B && k0 + kk < K) ? x[(long long)(bb + b) * K + k0 + kk] : 0.f;
    }
    __syncthreads();
    const int be = min(WB, B - bb);
    for (int b = 0; b < be; ++b) {
      const float dv = act ? dz[(long long)(bb + b) * (c + 1) + lane] : 0.f;
      bsum += dv;
#pragma unroll
      for (int kk = 0; kk < 64; kk += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[b][kw + kk]);
        acc[kk] += xv[0] * dv; acc[kk + 1] += xv[1] * dv; acc[kk + 2] += xv[2] * dv; acc[kk + 3] += xv[3] * dv;
      }
    }
  }
  if (blockIdx.x == 0 && wid == 0 && act) {
    float* o = lane == 0 ? dbpr : dbc + (lane - 1);
    *o = (accumulate ? *o : 0.f) + bsum;
  }
  if (!act) return;
#pragma unroll
  for (int kk = 0; kk < 64; ++kk) {
    const int k = k0 + kw + kk;
    if (k < K) {
      float* o = lane == 0 ? dwpr + k : dwc + (long long)k * c + (lane - 1);
      *o = (accumulate ? *o : 0.f) + acc[kk];
    }
  }
}

}  // namespace

extern "C" int lg_heads_fwd(const float* x, const float* wpr, const float* bpr, const float* wc, const float* bc,
                            float* p, int B, int K, int c, void* stream) {
  LG_CHECK_ARG(x && wpr && bpr && wc && bc && p, "lg_heads_fwd: null pointer");
  LG_CHECK_ARG(B > 0 && K > 0 && K % 4 == 0 && c >= 1 && c <= HC_MAX, "lg_heads_fwd: bad shape B=%d K=%d c=%d", B, K, c);
  const size_t lds = (size_t)(FB * KCH + KCH * c + KCH) * sizeof(float);
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(lg_cdiv(B, FB)), dim3(256), lds, (hipStream_t)stream, x, wpr, bpr, wc, bc, p, B,
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
  LG_CHECK_ARG(B > 0 && K > 0 && K % 4 == 0 && c >= 1 && c <= HC_MAX, "lg_heads_wgrad: bad shape B=%d K=%d c=%d", B, K, c);
  hipLaunchKernelGGL(heads_wgrad_kernel, dim3(lg_cdiv(K, KCH)), dim3(256), 0, (hipStream_t)stream, x, dz, dwpr, dbpr,
                     dwc, dbc, B, K, c, accumulate);
  LG_CHECK_LAUNCH("lg_heads_wgrad");
  return LG_OK;
}
