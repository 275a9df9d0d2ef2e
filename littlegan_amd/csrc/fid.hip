// Activation statistics of the FID pass (/root/reference/fid.py:185-188): mu = mean over samples, sigma = np.cov(act,
// rowvar=False) (divisor N - 1) of an [N, D] fp32 activation matrix (D = 2048 Inception pool_3 features), in fp64 on the
// device.  The O(N D^2) part is the Gram matrix of the CENTRED activations, computed with the fp64 matrix instruction
// (v_mfma_f64_16x16x4_f64): a block owns a 64 x 64 tile of sigma (upper-triangular tiles only; the mirror is written by the
// same block), streams all N samples through LDS in slabs of 32 rows — converted to fp64 and centred on the way in — and each
// of its 4 waves accumulates a 32 x 32 sub-tile (2 x 2 MFMA tiles, 4 fp64 accumulators per lane each).
// The D x D matrix square root of the Fréchet distance (fid.py:144-163) stays on the host (scipy), as in the reference.
#include "lg_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TILE = 64, KS = 32, LDP = TILE + 2;  // LDS row pitch in doubles (+2: the k-strided fragment reads spread over banks)

// colsum[s][d] = sum over the s-th row range of act[:, d]   (fp64)
__global__ __launch_bounds__(256) void fid_colsum_kernel(const float* __restrict__ act, double* __restrict__ part, long long N,
                                                         int D, long long rows_per) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  const long long r0 = (long long)blockIdx.y * rows_per, r1 = r0 + rows_per < N ? r0 + rows_per : N;
  if (d >= D) return;
  double s = 0.0;
  for (long long n = r0; n < r1; ++n) s += (double)act[n * D + d];
  part[(long long)blockIdx.y * D + d] = s;
}
__global__ __launch_bounds__(256) void fid_mean_kernel(const double* __restrict__ part, double* __restrict__ mu, long long N, int D,
                                                       int nsplit) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += part[(long long)k * D + d];
  mu[d] = s / (double)N;
}

// sigma[i0..i0+63][j0..j0+63] (and its mirror) = sum_n (x[n][i] - mu[i]) (x[n][j] - mu[j]) / (N - 1)
__global__ __launch_bounds__(256) void fid_cov_kernel(const float* __restrict__ act, const double* __restrict__ mu,
                                                      double* __restrict__ sigma, long long N, int D, int ntile) {
  // upper-triangular tile index -> (ti <= tj)
  int t = blockIdx.x, ti = 0;
  while (t >= ntile - ti) { t -= ntile - ti; ++ti; }
  const int tj = ti + t;
  const int i0 = ti * TILE, j0 = tj * TILE;
  __shared__ double sA[KS * LDP], sB[KS * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;  // wave's 32 x 32 sub-tile
  const int l15 = lane & 15, lk = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  // staging: thread -> (row r = tid / 16 [+16], 4 consecutive columns c4 = (tid % 16) * 4)
  const int sr = tid >> 4, sc = (tid & 15) * 4;
  double mA[4], mB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    mA[q] = (i0 + sc + q < D) ? mu[i0 + sc + q] : 0.0;
    mB[q] = (j0 + sc + q < D) ? mu[j0 + sc + q] : 0.0;
  }
  for (long long n0 = 0; n0 < N; n0 += KS) {
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < KS; rr += 16) {
      const long long n = n0 + sr + rr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double va = 0.0, vb = 0.0;
        if (n < N) {
          if (i0 + sc + q < D) va = (double)act[n * D + i0 + sc + q] - mA[q];
          if (j0 + sc + q < D) vb = (double)act[n * D + j0 + sc + q] - mB[q];
        }
        sA[(sr + rr) * LDP + sc + q] = va;
        sB[(sr + rr) * LDP + sc + q] = vb;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < KS; k4 += 4) {
      double af[2], bf[2];  // A[row = l15][k = lk] = x[k][i], B[k = lk][col = l15] = x[k][j]
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = sA[(k4 + lk) * LDP + wr * 32 + a * 16 + l15];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = sB[(k4 + lk) * LDP + wc * 32 + b * 16 + l15];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
  // C/D layout of the f64 instruction: col = lane & 15, row = (lane >> 4) + 4 * reg
  const double inv = 1.0 / (double)(N - 1);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + wr * 32 + a * 16 + lk + 4 * e, j = j0 + wc * 32 + b * 16 + l15;
        if (i < D && j < D) {
          const double v = acc[a][b][e] * inv;
          sigma[(long long)i * D + j] = v;
          if (ti != tj) sigma[(long long)j * D + i] = v;
        }
      }
}

inline int fid_nsplit(long long N) {
  long long s = (N + 255) / 256;
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

}  // namespace

extern "C" size_t lg_fid_stats_workspace_bytes(long long N, int D) { return (size_t)fid_nsplit(N) * (size_t)D * sizeof(double); }

// mu[D], sigma[D][D] (fp64, device) of act[N][D] (fp32, device); N >= 2
extern "C" int lg_fid_stats(const float* act, long long N, int D, double* mu, double* sigma, void* workspace, size_t ws_bytes,
                            void* stream) {
  LG_CHECK_ARG(act && mu && sigma && workspace, "lg_fid_stats: null pointer");
  LG_CHECK_ARG(N >= 2 && D > 0 && D <= (1 << 16), "lg_fid_stats: bad shape N=%lld D=%d", N, D);
  LG_CHECK_ARG(ws_bytes >= lg_fid_stats_workspace_bytes(N, D), "lg_fid_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int ns = fid_nsplit(N);
  const long long rows_per = (N + ns - 1) / ns;
  hipLaunchKernelGGL(fid_colsum_kernel, dim3((D + 255) / 256, ns), dim3(256), 0, st, act, (double*)workspace, N, D, rows_per);
  LG_CHECK_LAUNCH("lg_fid_stats(colsum)");
  hipLaunchKernelGGL(fid_mean_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const double*)workspace, mu, N, D, ns);
  LG_CHECK_LAUNCH("lg_fid_stats(mean)");
  const int ntile = (D + TILE - 1) / TILE;
  hipLaunchKernelGGL(fid_cov_kernel, dim3(ntile * (ntile + 1) / 2), dim3(256), 0, st, act, (const double*)mu, sigma, N, D, ntile);
  LG_CHECK_LAUNCH("lg_fid_stats(cov)");
  return LG_OK;
}
