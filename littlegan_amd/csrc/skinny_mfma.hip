// The "skinny" fp32 GEMMs of the step on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
// accumulate — the same arithmetic as the VALU kernels of heads.hip / dense.hip, at twice their FLOP rate and without
// their LDS broadcast traffic, which is what bounded them: heads_fwd moved 64x the activation bytes through LDS).
//   heads_fwd   p[b][j]  = sum_k x[b][k] W[k][j]            M = b, N = j (<= 41: two 32-column tiles), contraction k split
//                                                           over blocks AND over the 4 waves of a block -> partials
//   heads_wgrad dW[k][j] = sum_b x[b][k] dz[b][j]           M = k, N = j, contraction b split over the 4 waves of a block
//   heads_dgrad dx[b][k] = sum_j dz[b][j] W[k][j]           M = b, N = k, contraction j (<= 41 -> 21 steps)
//   dense_fwd   y[b][n]  = sum_k x[b][k] w[k][n] + bias[n]  M = b, N = n (wide), contraction k (133 -> 67 steps)
//   dense_wgrad dw[k][n] = sum_b x[b][k] dy[b][n]           M = k (133 -> 5 tiles), N = n, contraction b
// W = [wpr | wc]: column 0 is dense_pr.kernel [K][1], columns 1..c dense_cond.kernel [K][c]  (/root/reference/model.py:62-63).
// Operand layout of the instruction: lane (r = lane & 31, h = lane >> 5) supplies A[row r][k = h] and B[k = h][col r];
// accumulator register e holds row (e & 3) + 8 (e >> 2) + 4 h, column r.  The contraction index may be visited in any
// order as long as A and B agree, so a lane that owns 8 consecutive k (two 16-B loads of a row) simply uses them in 8
// successive instructions: its partner lane (h ^ 1) owns the other 8 of the 16.
// Entry points return LG_ERR_UNSUPPORTED for ragged shapes; the callers then run the VALU kernels.
#include <stdlib.h>
#include "lg_common.h"

namespace {

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}
// pointer / stride / validity of head column j: W[k][j] = ok ? ptr[k * stride] : 0
struct HeadCol { const float* ptr; int stride; bool ok; };
__device__ __forceinline__ HeadCol head_col(const float* wpr, const float* wc, int j, int c) {
  HeadCol h;
  h.ok = j <= c;
  h.ptr = j == 0 ? wpr : wc + (h.ok ? j - 1 : 0);
  h.stride = j == 0 ? 1 : c;
  return h;
}

// ---------------------------------------------------------------------------------------------------------------------
// grid (K / 512, B / 32), 256 threads: wave w contracts k in [k0 + 128 w, +128); partials merged in wave order through LDS
__global__ __launch_bounds__(256) void heads_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wpr,
                                                             const float* __restrict__ wc, float* __restrict__ part, int B,
                                                             int K, int c) {
  __shared__ float sred[4][2][16][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.y * 32, kw = blockIdx.x * 512 + wid * 128 + 8 * h;
  const float* xr = x + (long long)(b0 + r) * K + kw;
  const HeadCol c0 = head_col(wpr, wc, r, c), c1 = head_col(wpr, wc, 32 + r, c);
  const bool two = c >= 32;  // (uniform) the second column tile holds real columns
  f32x16 acc0 = zero16(), acc1 = zero16();
  constexpr int GU = 2;  // groups of 16 k per trip (this lane owns 8 of each): 2 GU + 16 GU loads in flight (latency-bound; GU 1 / 2 / 4: 40 / 33 / 34 us at B=256)
  for (int g2 = 0; g2 < 8; g2 += GU) {
    f32x4 alo[GU], ahi[GU];
    float w0[GU][8], w1[GU][8];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int g = g2 + u;
      alo[u] = *reinterpret_cast<const f32x4*>(xr + 16 * g); ahi[u] = *reinterpret_cast<const f32x4*>(xr + 16 * g + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const long long k = kw + 16 * g + e;
        w0[u][e] = c0.ok ? c0.ptr[k * c0.stride] : 0.f;
        w1[u][e] = (two && c1.ok) ? c1.ptr[k * c1.stride] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = e < 4 ? alo[u][e & 3] : ahi[u][e & 3];
        acc0 = mfma2(a, w0[u][e], acc0);
        if (two) acc1 = mfma2(a, w1[u][e], acc1);
      }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) { sred[wid][0][e][lane] = acc0[e]; sred[wid][1][e][lane] = acc1[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int row = i >> 6, j = i & 63;
    if (j <= c) {
      const int nt = j >> 5, rr = j & 31, hh = (row >> 2) & 1, e = (row & 3) + 4 * (row >> 3), l = hh * 32 + rr;
      part[((long long)blockIdx.x * B + b0 + row) * (c + 1) + j] =
          (sred[0][nt][e][l] + sred[1][nt][e][l]) + (sred[2][nt][e][l] + sred[3][nt][e][l]);
    }
  }
}

// grid K / 32 blocks, 256 threads: the block owns 32 rows of dW, its 4 waves split the batch (16 samples per trip) and merge
// in wave order through LDS (deterministic)
__global__ __launch_bounds__(256) void heads_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                              float* __restrict__ dwpr, float* __restrict__ dbpr,
                                                              float* __restrict__ dwc, float* __restrict__ dbc, int B, int K,
                                                              int c, int accumulate) {
  __shared__ float red[2][16][64];
  __shared__ float rbias[2][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.x * 32;
  const bool ok0 = r <= c, ok1 = 32 + r <= c, two = c >= 32;
  float bs0 = 0.f, bs1 = 0.f;  // column sums of dz (bias gradients): lane (r, h) sees column r / 32 + r, samples of parity h
  const int bq = B / 4, bs = wid * bq;
  const float* xp = x + (long long)(bs + h) * K + k0 + r;          // A[row = k0 + r][b = 2 s + h]
  const float* d0 = dz + (long long)(bs + h) * (c + 1) + r;         // B[b = 2 s + h][col r]
  const float* d1 = d0 + 32;
  f32x16 acc0 = zero16(), acc1 = zero16();
  constexpr int SU = 16;  // contraction steps (sample pairs) per trip: 3 SU loads in flight
  for (int b = 0; b < bq; b += 2 * SU) {
    float a[SU], v0[SU], v1[SU];
#pragma unroll
    for (int s = 0; s < SU; ++s) {
      const bool ok = b + 2 * s < bq;
      a[s] = ok ? xp[(long long)(b + 2 * s) * K] : 0.f;
      v0[s] = (ok && ok0) ? d0[(long long)(b + 2 * s) * (c + 1)] : 0.f;
      v1[s] = (ok && two && ok1) ? d1[(long long)(b + 2 * s) * (c + 1)] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < SU; ++s) {
      acc0 = mfma2(a[s], v0[s], acc0);
      if (two) acc1 = mfma2(a[s], v1[s], acc1);
      bs0 += v0[s]; bs1 += v1[s];
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        red[0][e][lane] = (w == 0 ? 0.f : red[0][e][lane]) + acc0[e];
        red[1][e][lane] = (w == 0 ? 0.f : red[1][e][lane]) + acc1[e];
      }
      rbias[0][lane] = (w == 0 ? 0.f : rbias[0][lane]) + bs0;
      rbias[1][lane] = (w == 0 ? 0.f : rbias[1][lane]) + bs1;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {  // consecutive i -> consecutive dwc addresses of one row
    const int row = i >> 6, j = i & 63;
    if (j <= c) {
      const int nt = j >> 5, rr = j & 31, hh = (row >> 2) & 1, e = (row & 3) + 4 * (row >> 3);
      const int k = k0 + row;
      float* o = j == 0 ? dwpr + k : dwc + (long long)k * c + (j - 1);
      *o = (accumulate ? *o : 0.f) + red[nt][e][hh * 32 + rr];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x <= c) {  // bias gradients (every block holds them; one writes): even + odd samples
    const int j = threadIdx.x, nt = j >> 5, rr = j & 31;
    float* o = j == 0 ? dbpr : dbc + (j - 1);
    *o = (accumulate ? *o : 0.f) + (rbias[nt][rr] + rbias[nt][32 + rr]);
  }
}

// grid (K / 128, gy), 256 threads: wave w owns the 32 k's k0 + 32 w (its W rows stay in registers as B fragments) and
// walks the sample tiles blockIdx.y, blockIdx.y + gy, ...
__global__ __launch_bounds__(256) void heads_dgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ wpr,
                                                               const float* __restrict__ wc, float* __restrict__ dx, int B,
                                                               int K, int c) {
  constexpr int SMAX = 21;  // (1 + 40 + 1) / 2 contraction steps
  extern __shared__ float sdz[];  // the block's sample tiles, [tile][32][c + 1 (+1 if even: odd pitch, conflict-free reads)]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int kt = blockIdx.x * 128 + wid * 32;
  const int cp = (c + 1) | 1, nmt = (B / 32 - blockIdx.y + gridDim.y - 1) / gridDim.y;
  for (int t = 0; t < nmt; ++t) {  // a row of dz is read by 32 lanes of every wave: staged once, coalesced
    const float* src = dz + (long long)(blockIdx.y + t * gridDim.y) * 32 * (c + 1);
    for (int i = threadIdx.x; i < 32 * (c + 1); i += 256) {
      const int row = i / (c + 1), j = i - row * (c + 1);
      sdz[(t * 32 + row) * cp + j] = src[i];
    }
  }
  float bw[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    const int j = 2 * s + h;
    bw[s] = j <= c ? (j == 0 ? wpr[kt + r] : wc[(long long)(kt + r) * c + j - 1]) : 0.f;
  }
  __syncthreads();
  for (int t = 0; t < nmt; ++t) {
    const int mt = blockIdx.y + t * gridDim.y;
    const float* zr = sdz + (t * 32 + r) * cp + h;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < SMAX; ++s) acc = mfma2(2 * s + h <= c ? zr[2 * s] : 0.f, bw[s], acc);  // steps beyond the real columns multiply zeros
    float* o = dx + (long long)(mt * 32 + 4 * h) * K + kt + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[(long long)((e & 3) + 8 * (e >> 2)) * K] = acc[e];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dense_fwd: grid (N / 128, B / 32), 256 threads: wave w owns the 32 columns n0 + 32 w; the 32 x K block of x is the A
// operand (K <= 2 * SD), loaded once per wave
constexpr int SD = 72;  // contraction steps held in registers: K <= 144
__global__ __launch_bounds__(256) void dense_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int B,
                                                             int K, int N) {
  extern __shared__ float sx[];  // [32][K | 1]: the 32 rows of x are one contiguous run; odd pitch -> conflict-free column reads
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int n = blockIdx.x * 128 + wid * 32 + r, b0 = blockIdx.y * 32;
  const int nsteps = (K + 1) / 2, kp = K | 1;
  {
    const float* src = x + (long long)b0 * K;
    int row = threadIdx.x / K, col = threadIdx.x - row * K;
    const int dq = 256 / K, dr = 256 - dq * K;
    for (int i = threadIdx.x; i < 32 * K; i += 256) {
      sx[row * kp + col] = src[i];
      row += dq; col += dr;
      if (col >= K) { col -= K; ++row; }
    }
  }
  __syncthreads();
  const float* xr = sx + r * kp + h;
  const float* wp = w + (long long)h * N + n;
  const float bv = bias ? bias[n] : 0.f;
  f32x16 acc, acc2 = zero16();  // two accumulation chains: a dependent chain of 67 MFMAs leaves the pipe idle between issues
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = bv;
#pragma unroll 4
  for (int s = 0; s < nsteps; s += 2) {
    const bool ok = 2 * s + h < K, ok2 = 2 * s + 2 + h < K;
    const float a = ok ? xr[2 * s] : 0.f, a2 = ok2 ? xr[2 * s + 2] : 0.f;
    const float b = ok ? wp[(long long)(2 * s) * N] : 0.f, b2 = ok2 ? wp[(long long)(2 * s + 2) * N] : 0.f;
    acc = mfma2(a, b, acc);
    acc2 = mfma2(a2, b2, acc2);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
  float* o = y + (long long)(b0 + 4 * h) * N + n;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[(long long)((e & 3) + 8 * (e >> 2)) * N] = acc[e];
}

// dense_wgrad: grid N / 32 blocks, 256 threads: the block owns 32 columns and ALL MT = ceil(K / 32) <= 5 row tiles of dw;
// its 4 waves split the batch (8 samples per trip) and merge in wave order through LDS; db = column sums of dy
template <int MT>
__global__ __launch_bounds__(256) void dense_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dw, float* __restrict__ db, int B, int K,
                                                               int N, int accumulate) {
  __shared__ float red[MT][16][64];
  __shared__ float rb[64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int n = blockIdx.x * 32 + r;
  const int bq = B / 4, bs = wid * bq;
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = zero16();
  float bsum = 0.f;
  const float* dp = dy + (long long)(bs + h) * N + n;        // B[b = 2 s + h][col]
  const float* xp = x + (long long)(bs + h) * K + r;         // A[row k = 32 m + r][b = 2 s + h]
  constexpr int SU = 8;  // contraction steps (sample pairs) per trip: (1 + MT) SU loads in flight
  for (int b = 0; b < bq; b += 2 * SU) {
    float d[SU], a[SU][MT];
#pragma unroll
    for (int s = 0; s < SU; ++s) {
      const bool ok = b + 2 * s < bq;
      d[s] = ok ? dp[(long long)(b + 2 * s) * N] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[s][m] = (ok && 32 * m + r < K) ? xp[(long long)(b + 2 * s) * K + 32 * m] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < SU; ++s) {
      bsum += d[s];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma2(a[s][m], d[s], acc[m]);
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[m][e][lane] = (w == 0 ? 0.f : red[m][e][lane]) + acc[m][e];
      rb[lane] = (w == 0 ? 0.f : rb[lane]) + bsum;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < MT * 32 * 32; i += 256) {  // 32 consecutive threads -> one 128-B row segment of dw
    const int k = i >> 5, rr = i & 31;
    if (k < K) {
      const int m = k >> 5, row = k & 31, hh = (row >> 2) & 1, e = (row & 3) + 4 * (row >> 3);
      float* o = dw + (long long)k * N + blockIdx.x * 32 + rr;
      *o = (accumulate ? *o : 0.f) + red[m][e][hh * 32 + rr];
    }
  }
  if (db && threadIdx.x < 32) {  // lanes (r, 0) and (r, 1) held the even / odd samples of column n
    float* o = db + blockIdx.x * 32 + threadIdx.x;
    *o = (accumulate ? *o : 0.f) + (rb[threadIdx.x] + rb[32 + threadIdx.x]);
  }
}

bool skinny_on() {
  static int on = -1;
  if (on < 0) on = lg_env_flag("LG_NO_SKINNY_MFMA") ? 0 : 1;
  return on == 1;
}

}  // namespace

extern "C" int lg_heads_fwd_mfma_try(const float* x, const float* wpr, const float* wc, float* part, int B, int K, int c,
                                     int* nkc_out, void* stream) {
  if (!skinny_on() || B % 32 || K % 512 || c < 1 || c > 40) return LG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(heads_fwd_mfma_kernel, dim3(K / 512, B / 32), dim3(256), 0, (hipStream_t)stream, x, wpr, wc, part, B, K, c);
  LG_CHECK_LAUNCH("lg_heads_fwd(mfma)");
  *nkc_out = K / 512;
  return LG_OK;
}

extern "C" int lg_heads_wgrad_mfma_try(const float* x, const float* dz, float* dwpr, float* dbpr, float* dwc, float* dbc, int B,
                                       int K, int c, int accumulate, void* stream) {
  if (!skinny_on() || B % 64 || K % 32 || c < 1 || c > 40) return LG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(heads_wgrad_mfma_kernel, dim3(K / 32), dim3(256), 0, (hipStream_t)stream, x, dz, dwpr, dbpr, dwc, dbc, B, K, c,
                     accumulate);
  LG_CHECK_LAUNCH("lg_heads_wgrad(mfma)");
  return LG_OK;
}

extern "C" int lg_heads_dgrad_mfma_try(const float* dz, const float* wpr, const float* wc, float* dx, int B, int K, int c,
                                       void* stream) {
  if (!skinny_on() || B % 32 || K % 128 || c < 1 || c > 40) return LG_ERR_UNSUPPORTED;
  const int mt = B / 32, gy = mt >= 8 ? 4 : (mt >= 2 ? 2 : 1);
  const size_t lds = (size_t)((mt + gy - 1) / gy) * 32 * ((c + 1) | 1) * sizeof(float);
  if (lds > 48 * 1024) return LG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(heads_dgrad_mfma_kernel, dim3(K / 128, gy), dim3(256), lds, (hipStream_t)stream, dz, wpr, wc, dx, B, K, c);
  LG_CHECK_LAUNCH("lg_heads_dgrad(mfma)");
  return LG_OK;
}

extern "C" int lg_dense_fwd_mfma_try(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                                     void* stream) {
  if (!skinny_on() || B % 32 || N % 128 || K < 1 || K > 2 * SD) return LG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dense_fwd_mfma_kernel, dim3(N / 128, B / 32), dim3(256), (size_t)32 * (K | 1) * sizeof(float),
                     (hipStream_t)stream, x, w, bias, y, B, K, N);
  LG_CHECK_LAUNCH("lg_dense_fwd(mfma)");
  return LG_OK;
}

extern "C" int lg_dense_wgrad_mfma_try(const float* x, const float* dy, float* dw, float* db, int B, int K, int N, int accumulate,
                                       void* stream) {
  if (!skinny_on() || B % 32 || N % 32 || K < 1 || K > 160) return LG_ERR_UNSUPPORTED;
  const int mt = (K + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(N / 32);
#define LG_DW(M) hipLaunchKernelGGL(dense_wgrad_mfma_kernel<M>, grid, dim3(256), 0, st, x, dy, dw, db, B, K, N, accumulate)
  if (mt == 1) LG_DW(1); else if (mt == 2) LG_DW(2); else if (mt == 3) LG_DW(3); else if (mt == 4) LG_DW(4); else LG_DW(5);
#undef LG_DW
  LG_CHECK_LAUNCH("lg_dense_wgrad(mfma)");
  return LG_OK;
}
