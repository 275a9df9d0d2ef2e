// InstanceNormalization(axis=None, eps=1e-3) of /root/reference/instance.py:105-128 as used at
// model.py:16,41,84,121: per-SAMPLE moments over all of (H,W,C), scalar gamma/beta, eps added to the
// std.  Because gamma/beta are scalars the whole op is a per-sample affine y = a_n*(x - mu_n) + beta with
// a_n = gamma/(sigma_n+eps); LeakyReLU(0.3) (model.py:24,50,100,130) and the decoder's skip add
// (model.py:46-47) are fused into the same elementwise pass.
//   stats : two-level reduction, block-local mean / M2 merged with Chan's formula, everything accumulated
//           in fp64 in a fixed order -> deterministic, no atomics.
//           stats[n][8] = {mu_hi, sigma, a, beta, mu_lo, 0, 0, 0}   (mu = mu_hi + mu_lo, float-float)
//   apply : y = [leaky]( a*(([leaky](x) - mu_hi) - mu_lo) + beta ) [+ skip]
//   bwd   : dx = a*(dz - m1 - c*m2'), c = x-mu, m1 = mean(dz), m2' = mean(dz*c)/((sigma+eps)*sigma);
//           dgamma = sum dz*c/(sigma+eps), dbeta = sum dz.
// Why fp64 sums and float-float means: mu, m1 and m2' are subtracted from EVERY element of a sample, so their
// rounding error is coherent; the later reductions of the step (bias / weight gradients = sums over up to
// 10^6 pixels of quantities whose mean was removed here) amplify a coherent 1e-7 error to 1e-3.
// All kernels are HBM-bound: 16-B vector loads, one pass over x for the moments (values kept in registers).
#include "lg_common.h"

#define LG_IN_EPS 1e-3f
#define LG_NSTAT 8

namespace {

constexpr int CHUNK = 4096;  // elements per block in the reduction passes (256 threads x 4 float4)
constexpr int EW_UNR = 4;    // float4 per thread per trip of the elementwise passes

// partial[n][chunk] = {count, mean, M2} (doubles)
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                            long long L, int nchunk, int pre_leaky, float alpha,
                                                            __bf16* __restrict__ x16out) {
  const int n = blockIdx.y, ch = blockIdx.x;
  const long long base = (long long)n * L + (long long)ch * CHUNK;
  const long long lim = L - (long long)ch * CHUNK;  // valid elements in this chunk
  __shared__ double sred[32];
  f32x4 v[4];
  double s = 0.0;
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = (q * 256 + threadIdx.x) * 4;
    v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e < lim) {
      v[q] = *reinterpret_cast<const f32x4*>(x + base + e);
      if (x16out) {  // bf16 activation path: the bf16 copy of the raw conv output is made in this same pass
        bf16x4 w;
        w[0] = (__bf16)v[q][0]; w[1] = (__bf16)v[q][1]; w[2] = (__bf16)v[q][2]; w[3] = (__bf16)v[q][3];
        *reinterpret_cast<bf16x4*>(x16out + base + e) = w;
      }
      if (pre_leaky) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[q][k] = lg_leaky(v[q][k], alpha);
      }
      s += ((double)v[q][0] + (double)v[q][1]) + ((double)v[q][2] + (double)v[q][3]);
      cnt += 4;
    }
  }
  double red[2] = {s, (double)cnt};
  lg_block_sum_d<2>(red, sred);
  __shared__ double s_mean;
  if (threadIdx.x == 0) s_mean = red[0] / red[1];
  __syncthreads();
  const double mean = s_mean;
  double m2 = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = (q * 256 + threadIdx.x) * 4;
    if (e < lim) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const double d = (double)v[q][k] - mean; m2 += d * d; }
    }
  }
  double red2[1] = {m2};
  lg_block_sum_d<1>(red2, sred);
  if (threadIdx.x == 0) {
    double* o = partial + ((long long)n * nchunk + ch) * 3;
    o[0] = red[1]; o[1] = mean; o[2] = red2[0];
  }
}

// one wave per sample: Chan merge of the chunk partials in fp64
__global__ __launch_bounds__(64) void stats_final_kernel(const double* __restrict__ partial, float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int nchunk) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const double* p = partial + (long long)n * nchunk * 3;
  double cnt = 0.0, sum = 0.0;
  for (int i = lane; i < nchunk; i += 64) { cnt += p[i * 3]; sum += p[i * 3] * p[i * 3 + 1]; }
  cnt = lg_wave_sum_d(cnt); sum = lg_wave_sum_d(sum);
  const double mean = sum / cnt;
  double m2 = 0.0;
  for (int i = lane; i < nchunk; i += 64) {
    const double d = p[i * 3 + 1] - mean;
    m2 += p[i * 3 + 2] + p[i * 3] * d * d;
  }
  m2 = lg_wave_sum_d(m2);
  if (lane == 0) {
    const double sigma = sqrt(m2 / cnt);
    const double a = (double)gamma[0] / (sigma + (double)LG_IN_EPS);
    float* o = stats + (long long)n * LG_NSTAT;
    const float mu_hi = (float)mean;
    o[0] = mu_hi; o[1] = (float)sigma; o[2] = (float)a; o[3] = beta[0];
    o[4] = (float)(mean - (double)mu_hi); o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
  }
}

// Non-temporal accesses of the element-wise passes, bf16 and fp32 path alike (bits: 1 backward-apply loads, 2 backward-apply stores, 4 apply loads, 8 apply stores).
// z, g and the skip map are dead behind these passes (z is read again only a whole tape later): loaded non-temporally they stop evicting
// what the neighbouring conv kernels live on (their sources and weights in L2 / the Infinity Cache).  Round 4, whole step on one box,
// variant builds, two rounds: 11.15 / 11.17 ms (0) -> 11.06 / 11.07 (1) -> 10.98 / 10.97 (5) = -1.6 %; the stores (3, 7: 11.07 / 10.98) add
// nothing there — although the passes in isolation (scripts/bench_norm.py) show the opposite: stores -9 .. -13 %, loads nothing.
// The same treatment of the final layer's input rows, the final weight gradient's activation operand and up_p16's source (their
// last readers) moved the step by nothing (10.79 - 10.86 against 10.81 - 10.90 ms): removed.
#ifndef LG_NORM_NT
#define LG_NORM_NT 5
#endif
// The bf16 apply / backward-apply passes walk their maps from the END (bits: 1 apply16p, 2 bwd_apply16; 4 / 8 their fp32 twins).  Their producers (the conv
// kernels) write a map front to back and their consumers read it front to back: a pass in between that starts at the END meets what the
// producer wrote last — still in the Infinity Cache when the map is larger than it — and leaves the START of its own output for the
// consumer's first reads.  Round 4, whole step, one box, three rounds: 10.79 / 10.76 / 10.72 ms (0) against 10.75 / 10.66 / 10.64 (3),
// captured 10.70 against 10.66; results unchanged except the bias column sums of a multi-trip backward pass (trip order).
#ifndef LG_NORM_REV
#define LG_NORM_REV 3
#endif
template <bool NT, typename V>
__device__ __forceinline__ V lg_ld(const V* q) { if constexpr (NT) return __builtin_nontemporal_load(q); else return *q; }

__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                    const float* __restrict__ skip, float* __restrict__ y,
                                                    __bf16* __restrict__ y16, long long L4, long long total4,
                                                    int pre_leaky, int post_leaky, float alpha) {
  // EW_UNR independent 16-B loads per thread per trip: a single load in flight per thread leaves the pass latency-bound
  // (measured 2.5-3.9 TB/s against 5.5+ for a streaming copy)
  const unsigned stride = gridDim.x * blockDim.x * EW_UNR, tot = (unsigned)total4, l4 = (unsigned)L4;
  const unsigned rb = (LG_NORM_REV & 4) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const int ntrip = (int)(((unsigned long long)tot + stride - 1) / stride);
  for (int tr = 0; tr < ntrip; ++tr) {  // block-contiguous 16-KB trips (LG_NORM_REV & 4: from the end of the map)
    const unsigned long long i0l = (unsigned long long)((LG_NORM_REV & 4) ? ntrip - 1 - tr : tr) * stride + rb * blockDim.x * EW_UNR + threadIdx.x;
    if (i0l >= tot) continue;
    const unsigned i0 = (unsigned)i0l;
    f32x4 v[EW_UNR], sk[EW_UNR];
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
      const unsigned i = i0 + u * 256;
      if (i < tot) {
        v[u] = lg_ld<(LG_NORM_NT & 4) != 0>(reinterpret_cast<const f32x4*>(x + (long long)i * 4));
        if (skip) sk[u] = lg_ld<(LG_NORM_NT & 4) != 0>(reinterpret_cast<const f32x4*>(skip + (long long)i * 4));
      }
    }
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
      const unsigned i = i0 + u * 256;
      if (i >= tot) break;
      const float* sp = stats + (long long)(i / l4) * LG_NSTAT;
      const float mu = sp[0], a = sp[2], b = sp[3], mul = sp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = v[u][k];
        if (pre_leaky) t = lg_leaky(t, alpha);
        t = a * ((t - mu) - mul) + b;  // (x-mu)/(sigma+eps)*gamma + beta, as instance.py:116-127 (no cancellation)
        if (post_leaky) t = lg_leaky(t, alpha);
        v[u][k] = t;
      }
      if (skip) v[u] += sk[u];
      if (y) *reinterpret_cast<f32x4*>(y + (long long)i * 4) = v[u];
      if (y16) {  // bf16 mirror = exactly the MFMA operand the consumers would round to themselves
        bf16x4 w;
        w[0] = (__bf16)v[u][0]; w[1] = (__bf16)v[u][1]; w[2] = (__bf16)v[u][2]; w[3] = (__bf16)v[u][3];
        *reinterpret_cast<bf16x4*>(y16 + (long long)i * 4) = w;
      }
    }
  }
}

template <bool NT = false>
__device__ __forceinline__ f32x4 load_g4(const void* g, long long i4, int g16) {  // 4 gradient values, fp32 or bf16 storage
  if (g16) {
    const bf16x4 w = lg_ld<NT>(reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(g) + i4));
    return f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
  }
  return lg_ld<NT>(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + i4));
}

// partial[n][chunk] = {sum dz, sum dz*c} (doubles)
__global__ __launch_bounds__(256) void bwd_partial_kernel(const float* __restrict__ x, const void* __restrict__ g, int g16,
                                                          const float* __restrict__ stats, double* __restrict__ partial,
                                                          long long L, int nchunk, int pre_leaky, int post_leaky,
                                                          float alpha) {
  // gridDim.x blocks per sample, each sweeping chunks blockIdx.x, blockIdx.x + gridDim.x, ... (few long-lived blocks
  // stream better than one short block per chunk), ONE block reduction at the end; nchunk = gridDim.x partial records
  const int n = blockIdx.y;
  const float* sp = stats + (long long)n * LG_NSTAT;
  const float mu = sp[0], a = sp[2], b = sp[3], mul = sp[4];
  __shared__ double sred[32];
  double s1 = 0.0, s2 = 0.0;
  for (long long c0 = (long long)blockIdx.x * CHUNK; c0 < L; c0 += (long long)gridDim.x * CHUNK) {
   const long long base = (long long)n * L + c0;
   const long long lim = L - c0;
#pragma unroll
   for (int q = 0; q < 4; ++q) {
    const int e = (q * 256 + threadIdx.x) * 4;
    if (e < lim) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + e);
      const f32x4 gv = load_g4(g, base + e, g16);
      // the 4 values of a quad are summed in fp32 (c = (x - mu_hi) - mu_lo is the float-float centred value the
      // forward pass uses, so no coherent error enters), the quads in fp64: the fp64 VALU rate, not HBM, bounded this pass
      float q1 = 0.f, q2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xx = xv[k];
        if (pre_leaky) xx = lg_leaky(xx, alpha);
        const float c = (xx - mu) - mul;
        float dz = gv[k];
        if (post_leaky) dz = (a * c + b > 0.f) ? dz : alpha * dz;
        q1 += dz;
        q2 += dz * c;
      }
      s1 += (double)q1;
      s2 += (double)q2;
    }
   }
  }
  double red[2] = {s1, s2};
  lg_block_sum_d<2>(red, sred);
  if (threadIdx.x == 0) {
    double* o = partial + ((long long)n * nchunk + blockIdx.x) * 2;
    o[0] = red[0]; o[1] = red[1];
  }
}

// bstats[n][4] = {m1_hi, m2'_hi, m1_lo, m2'_lo} ; gsum[n] = per-sample dgamma / dbeta contributions
__global__ __launch_bounds__(64) void bwd_final_kernel(const double* __restrict__ partial, const float* __restrict__ stats,
                                                       float* __restrict__ bstats, double* __restrict__ gsum,
                                                       long long L, int nchunk) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const double* p = partial + (long long)n * nchunk * 2;
  double s1 = 0.0, s2 = 0.0;
  for (int i = lane; i < nchunk; i += 64) { s1 += p[i * 2]; s2 += p[i * 2 + 1]; }
  s1 = lg_wave_sum_d(s1); s2 = lg_wave_sum_d(s2);
  if (lane == 0) {
    const double sigma = (double)stats[(long long)n * LG_NSTAT + 1], s = sigma + (double)LG_IN_EPS;
    const double m1 = s1 / (double)L, m2 = s2 / (double)L / (s * sigma);
    const float m1h = (float)m1, m2h = (float)m2;
    bstats[n * 4] = m1h; bstats[n * 4 + 1] = m2h;
    bstats[n * 4 + 2] = (float)(m1 - (double)m1h); bstats[n * 4 + 3] = (float)(m2 - (double)m2h);
    gsum[n * 2] = s2 / s;  // dgamma contribution
    gsum[n * 2 + 1] = s1;  // dbeta contribution
  }
}

// coef[n][8] = {mu_hi, mu_lo, a, beta, m1_hi, m2'_hi, m1_lo, m2'_lo}: everything a consumer needs to form the norm backward's dz
// of sample n from (z, g) on the fly (lg_bwdnorm8, lg_common.h) — the sums are merged exactly as bwd_final_kernel merges them
__global__ __launch_bounds__(64) void bwd_coef_kernel(const double* __restrict__ partial, const float* __restrict__ stats,
                                                      float* __restrict__ coef, long long L, int nchunk) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const double* p = partial + (long long)n * nchunk * 2;
  double s1 = 0.0, s2 = 0.0;
  for (int i = lane; i < nchunk; i += 64) { s1 += p[i * 2]; s2 += p[i * 2 + 1]; }
  s1 = lg_wave_sum_d(s1); s2 = lg_wave_sum_d(s2);
  if (lane == 0) {
    const float* sp = stats + (long long)n * LG_NSTAT;
    const double sigma = (double)sp[1], s = sigma + (double)LG_IN_EPS;
    const double m1 = s1 / (double)L, m2 = s2 / (double)L / (s * sigma);
    const float m1h = (float)m1, m2h = (float)m2;
    float* o = coef + (long long)n * 8;
    o[0] = sp[0]; o[1] = sp[4]; o[2] = sp[2]; o[3] = sp[3];
    o[4] = m1h; o[5] = m2h; o[6] = (float)(m1 - (double)m1h); o[7] = (float)(m2 - (double)m2h);
  }
}

__global__ __launch_bounds__(256) void bwd_affine_grad_kernel(const double* __restrict__ gsum, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int B, int accumulate) {
  __shared__ double sg[256], sb[256];
  double g = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) { g += gsum[i * 2]; b += gsum[i * 2 + 1]; }
  sg[threadIdx.x] = g; sb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sg[threadIdx.x] += sg[threadIdx.x + o]; sb[threadIdx.x] += sb[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dgamma[0] = (accumulate ? dgamma[0] : 0.f) + (float)sg[0];
    dbeta[0] = (accumulate ? dbeta[0] : 0.f) + (float)sb[0];
  }
}

// DB: also the column sums of dx over (sample, position) = the bias gradient of the conv that produced x ([.., C]
// channels innermost).  The launch makes gridDim.x*256 a multiple of C/4, so a thread meets the same 4 channels in
// every trip; threads of a block are merged in thread order into colpart[block][C] (deterministic), blocks by
// colsum_final (below).
template <bool DB>
__global__ __launch_bounds__(256) void bwd_apply_kernel(const float* __restrict__ x, const void* __restrict__ g, int g16,
                                                        const float* __restrict__ stats, const float* __restrict__ bstats,
                                                        float* __restrict__ dx, __bf16* __restrict__ dx16, long long L4,
                                                        long long total4, int pre_leaky, int post_leaky, float alpha,
                                                        float* __restrict__ colpart, int C4) {
  constexpr int UNR = DB ? 1 : EW_UNR;  // DB: a thread must stay on one channel quad -> plain grid stride
  const unsigned stride = gridDim.x * blockDim.x * UNR, tot = (unsigned)total4, l4 = (unsigned)L4;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const unsigned rb = (LG_NORM_REV & 8) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;   // (data and column-sum row alike)
  const int ntrip = (int)(((unsigned long long)tot + stride - 1) / stride);
  for (int tr = 0; tr < ntrip; ++tr) {
   const unsigned long long i0l = (unsigned long long)((LG_NORM_REV & 8) ? ntrip - 1 - tr : tr) * stride + rb * blockDim.x * UNR + threadIdx.x;
   if (i0l >= tot) continue;
   const unsigned i0 = (unsigned)i0l;
   f32x4 xs[UNR], gs[UNR];
#pragma unroll
   for (int u = 0; u < UNR; ++u) {
     const unsigned i = i0 + u * 256;
     if (i < tot) {
       xs[u] = lg_ld<(LG_NORM_NT & 1) != 0>(reinterpret_cast<const f32x4*>(x + (long long)i * 4));   // last reader of z and g
       gs[u] = load_g4<(LG_NORM_NT & 1) != 0>(g, (long long)i * 4, g16);
     }
   }
#pragma unroll
   for (int u = 0; u < UNR; ++u) {
    const unsigned i = i0 + u * 256;
    if (i >= tot) break;
    const int n = (int)(i / l4);
    const float* sp = stats + (long long)n * LG_NSTAT;
    const float mu = sp[0], a = sp[2], b = sp[3], mul = sp[4];
    const float m1 = bstats[n * 4], m2 = bstats[n * 4 + 1], m1l = bstats[n * 4 + 2], m2l = bstats[n * 4 + 3];
    const f32x4 xv = xs[u], gv = gs[u];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float xx = xv[k];
      if (pre_leaky) xx = lg_leaky(xx, alpha);
      const float c = (xx - mu) - mul;
      float dz = gv[k];
      if (post_leaky) dz = (a * c + b > 0.f) ? dz : alpha * dz;
      float d = a * ((((dz - m1) - m1l) - c * m2) - c * m2l);
      if (pre_leaky) d = (xv[k] > 0.f) ? d : alpha * d;
      o[k] = d;
    }
    if (dx) *reinterpret_cast<f32x4*>(dx + (long long)i * 4) = o;
    if (dx16) {
      bf16x4 w;
      w[0] = (__bf16)o[0]; w[1] = (__bf16)o[1]; w[2] = (__bf16)o[2]; w[3] = (__bf16)o[3];
      *reinterpret_cast<bf16x4*>(dx16 + (long long)i * 4) = w;
    }
    if constexpr (DB) csum += o;
   }
  }
  if constexpr (DB) {
    __shared__ f32x4 sacc[256];
    sacc[threadIdx.x] = csum;
    __syncthreads();
    const int q = threadIdx.x;  // channel quad
    if (q < C4) {
      const int base = (int)(((long long)rb * 256) % C4);
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int k = (q - base + C4) % C4; k < 256; k += C4) t += sacc[k];
      *reinterpret_cast<f32x4*>(colpart + ((long long)rb * C4 + q) * 4) = t;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16 activation path ("z16"): in the bf16 MFMA configuration the raw conv output z lives in HBM as bf16 only (the
// moments come from the fp32 accumulators of the conv epilogue, so they are unaffected), the skip tensors are the bf16
// mirrors the encoder wrote anyway, and the gradients arrive as bf16.  These kernels move 8 elements (16 B of bf16) per
// thread per access; fp32 operands, where one is mixed in, are two 16-B accesses.  L % 8 == 0.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
struct f32x8 { f32x4 lo, hi; };

template <bool IS16, bool NT = false>
__device__ __forceinline__ f32x8 load8(const void* p, long long i8) {  // elements [8*i8, 8*i8 + 8)
  f32x8 r;
  if constexpr (IS16) {
    const bf16x8v* q = reinterpret_cast<const bf16x8v*>(reinterpret_cast<const __bf16*>(p) + i8 * 8);
    const bf16x8v w = NT ? __builtin_nontemporal_load(q) : *q;
    r.lo = f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
    r.hi = f32x4{(float)w[4], (float)w[5], (float)w[6], (float)w[7]};
  } else {
    const float* f = reinterpret_cast<const float*>(p) + i8 * 8;
    r.lo = *reinterpret_cast<const f32x4*>(f);
    r.hi = *reinterpret_cast<const f32x4*>(f + 4);
  }
  return r;
}
template <bool NT = false>
__device__ __forceinline__ void store8_bf16(__bf16* p, long long i8, const f32x8& v) {
  bf16x8v w;
  w[0] = (__bf16)v.lo[0]; w[1] = (__bf16)v.lo[1]; w[2] = (__bf16)v.lo[2]; w[3] = (__bf16)v.lo[3];
  w[4] = (__bf16)v.hi[0]; w[5] = (__bf16)v.hi[1]; w[6] = (__bf16)v.hi[2]; w[7] = (__bf16)v.hi[3];
  if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<bf16x8v*>(p + i8 * 8));
  else *reinterpret_cast<bf16x8v*>(p + i8 * 8) = w;
}
__device__ __forceinline__ void store8_f32(float* p, long long i8, const f32x8& v) {
  *reinterpret_cast<f32x4*>(p + i8 * 8) = v.lo;
  *reinterpret_cast<f32x4*>(p + i8 * 8 + 4) = v.hi;
}

#ifndef LG_EW8_UNR
#define LG_EW8_UNR 2   // r3 sweep (same box, B=256 maps): 2 beats 4 and 8 by 3 - 8 % on apply, apply + skip and the backward apply (more waves resident)
#endif
#ifndef LG_DB_UNR
#define LG_DB_UNR 2
#endif
constexpr int EW8_UNR = LG_EW8_UNR;  // 16-B accesses in flight per thread and stream

// y = leaky(a*((z - mu_hi) - mu_lo) + beta) [+ skip]  from the bf16 z;  SK: 0 none, 1 fp32 skip, 2 bf16 skip
template <int SK>
__global__ __launch_bounds__(256) void apply16_kernel(const __bf16* __restrict__ x, const float* __restrict__ stats,
                                                      const void* __restrict__ skip, float* __restrict__ y,
                                                      __bf16* __restrict__ y16, long long L8, long long total8,
                                                      int pre_leaky, int post_leaky, float alpha) {
  const unsigned stride = gridDim.x * blockDim.x * EW8_UNR, tot = (unsigned)total8, l8 = (unsigned)L8;
  for (unsigned i0 = blockIdx.x * blockDim.x * EW8_UNR + threadIdx.x; i0 < tot; i0 += stride) {
    f32x8 v[EW8_UNR], sk[EW8_UNR];
#pragma unroll
    for (int u = 0; u < EW8_UNR; ++u) {
      const unsigned i = i0 + u * 256;
      if (i < tot) {
        v[u] = load8<true>(x, i);
        if constexpr (SK == 1) sk[u] = load8<false>(skip, i);
        if constexpr (SK == 2) sk[u] = load8<true>(skip, i);
      }
    }
#pragma unroll
    for (int u = 0; u < EW8_UNR; ++u) {
      const unsigned i = i0 + u * 256;
      if (i >= tot) break;
      const float* sp = stats + (long long)(i / l8) * LG_NSTAT;
      const float mu = sp[0], a = sp[2], b = sp[3], mul = sp[4];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = k < 4 ? v[u].lo[k & 3] : v[u].hi[k & 3];
        if (pre_leaky) t = lg_leaky(t, alpha);
        t = a * ((t - mu) - mul) + b;
        if (post_leaky) t = lg_leaky(t, alpha);
        if constexpr (SK != 0) t += k < 4 ? sk[u].lo[k & 3] : sk[u].hi[k & 3];
        if (k < 4) v[u].lo[k & 3] = t; else v[u].hi[k & 3] = t;
      }
      if (y) store8_f32(y, i, v[u]);
      if (y16) store8_bf16(y16, i, v[u]);
    }
  }
}

// apply16 with the statistics FINALISED IN THE SAME LAUNCH: grid (blocks per sample, B); the first wave of every block merges the
// sample's moment partials {count, mean, M2} exactly as stats_final_kernel does (same order, same fp64 arithmetic: the record is
// bit-identical), block 0 of a sample writes the record for the later consumers (backward, fused conv epilogues).  One launch and
// one kernel boundary less per normalised map (22 per C3 step); post-LeakyReLU form only (pre_leaky = 0).
template <int SK>
__global__ __launch_bounds__(256) void apply16p_kernel(const __bf16* __restrict__ x, const double* __restrict__ partial, int nchunk,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats, const void* __restrict__ skip,
                                                       float* __restrict__ y, __bf16* __restrict__ y16, unsigned L8,
                                                       int post_leaky, float alpha) {
  __shared__ float sst[8];
  // (LG_NORM_REV & 1: samples and chunks from the end — the producer wrote the end of the map last, the consumer reads its start first)
  const int n = (LG_NORM_REV & 1) ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const unsigned bx = (LG_NORM_REV & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const double* p = partial + (long long)n * nchunk * 3;
    double cnt = 0.0, sum = 0.0;
    for (int i = lane; i < nchunk; i += 64) { cnt += p[i * 3]; sum += p[i * 3] * p[i * 3 + 1]; }
    cnt = lg_wave_sum_d(cnt); sum = lg_wave_sum_d(sum);
    const double mean = sum / cnt;
    double m2 = 0.0;
    for (int i = lane; i < nchunk; i += 64) {
      const double d = p[i * 3 + 1] - mean;
      m2 += p[i * 3 + 2] + p[i * 3] * d * d;
    }
    m2 = lg_wave_sum_d(m2);
    if (lane == 0) {
      const double sigma = sqrt(m2 / cnt);
      const double a = (double)gamma[0] / (sigma + (double)LG_IN_EPS);
      const float mu_hi = (float)mean;
      sst[0] = mu_hi; sst[1] = (float)sigma; sst[2] = (float)a; sst[3] = beta[0]; sst[4] = (float)(mean - (double)mu_hi);
      if (bx == 0) {
        float* o = stats + (long long)n * LG_NSTAT;
        o[0] = mu_hi; o[1] = (float)sigma; o[2] = (float)a; o[3] = beta[0];
        o[4] = (float)(mean - (double)mu_hi); o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
      }
    }
  }
  __syncthreads();
  const float mu = sst[0], a = sst[2], b = sst[3], mul = sst[4];
  const long long base = (long long)n * L8;
  const unsigned stride = gridDim.x * blockDim.x * EW8_UNR;
  const int ntrip = (int)((L8 + stride - 1) / stride);
  for (int tr = 0; tr < ntrip; ++tr) {
    const unsigned i0 = (unsigned)((LG_NORM_REV & 1) ? ntrip - 1 - tr : tr) * stride + bx * blockDim.x * EW8_UNR + threadIdx.x;
    if (i0 >= L8) continue;
    f32x8 v[EW8_UNR], sk[EW8_UNR];
#pragma unroll
    for (int u = 0; u < EW8_UNR; ++u) {
      const unsigned i = i0 + u * 256;
      if (i < L8) {
        v[u] = load8<true, (LG_NORM_NT & 4) != 0>(x, base + i);
        if constexpr (SK == 1) sk[u] = load8<false, (LG_NORM_NT & 4) != 0>(skip, base + i);
        if constexpr (SK == 2) sk[u] = load8<true, (LG_NORM_NT & 4) != 0>(skip, base + i);
      }
    }
#pragma unroll
    for (int u = 0; u < EW8_UNR; ++u) {
      const unsigned i = i0 + u * 256;
      if (i >= L8) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = k < 4 ? v[u].lo[k & 3] : v[u].hi[k & 3];
        t = a * ((t - mu) - mul) + b;
        if (post_leaky) t = lg_leaky(t, alpha);
        if constexpr (SK != 0) t += k < 4 ? sk[u].lo[k & 3] : sk[u].hi[k & 3];
        if (k < 4) v[u].lo[k & 3] = t; else v[u].hi[k & 3] = t;
      }
      if (y) store8_f32(y, base + i, v[u]);
      if (y16) store8_bf16<(LG_NORM_NT & 8) != 0>(y16, base + i, v[u]);
    }
  }
}

// partial[n][blk] = {sum dz, sum dz*c} from the bf16 z; G16: gradient stored as bf16
template <bool G16>
__global__ __launch_bounds__(256) void bwd_partial16_kernel(const __bf16* __restrict__ x, const void* __restrict__ g,
                                                            const float* __restrict__ stats, double* __restrict__ partial,
                                                            long long L, int nchunk, int pre_leaky, int post_leaky,
                                                            float alpha) {
  const int n = blockIdx.y;
  const float* sp = stats + (long long)n * LG_NSTAT;
  const float mu = sp[0], a = sp[2], b = sp[3], mul = sp[4];
  __shared__ double sred[32];
  double s1 = 0.0, s2 = 0.0;
  constexpr int CH8 = 2 * CHUNK;  // elements per block trip (256 threads x 4 x 8)
  for (long long c0 = (long long)blockIdx.x * CH8; c0 < L; c0 += (long long)gridDim.x * CH8) {
    const long long base8 = ((long long)n * L + c0) / 8;
    const long long lim = L - c0;
    f32x8 xv[4], gv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = (q * 256 + threadIdx.x) * 8;
      if (e < lim) { xv[q] = load8<true>(x, base8 + e / 8); gv[q] = load8<G16>(g, base8 + e / 8); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = (q * 256 + threadIdx.x) * 8;
      if (e < lim) {
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float xx = k < 4 ? xv[q].lo[k & 3] : xv[q].hi[k & 3];
          if (pre_leaky) xx = lg_leaky(xx, alpha);
          const float c = (xx - mu) - mul;
          float dz = k < 4 ? gv[q].lo[k & 3] : gv[q].hi[k & 3];
          if (post_leaky) dz = (a * c + b > 0.f) ? dz : alpha * dz;
          q1 += dz;
          q2 += dz * c;
        }
        s1 += (double)q1;
        s2 += (double)q2;
      }
    }
  }
  double red[2] = {s1, s2};
  lg_block_sum_d<2>(red, sred);
  if (threadIdx.x == 0) {
    double* o = partial + ((long long)n * nchunk + blockIdx.x) * 2;
    o[0] = red[0]; o[1] = red[1];
  }
}

// dz = a*(dy' - m1 - c*m2') from the bf16 z; DB: + column sums (bias gradient); the launch makes gridDim.x*256 a multiple
// of C/8, so a thread meets the same 8 channels in every trip
template <bool DB, bool G16>
__global__ __launch_bounds__(256) void bwd_apply16_kernel(const __bf16* __restrict__ x, const void* __restrict__ g,
                                                          const float* __restrict__ stats, const float* __restrict__ bstats,
                                                          float* __restrict__ dx, __bf16* __restrict__ dx16, long long L8,
                                                          long long total8, int pre_leaky, int post_leaky, float alpha,
                                                          float* __restrict__ colpart, int C8) {
  constexpr int UNR = DB ? LG_DB_UNR : EW8_UNR;  // DB: the extra loads of a trip sit gridDim.x*256 units apart (same channel octet)
  const unsigned tot = (unsigned)total8, l8 = (unsigned)L8;
  const unsigned ustep = DB ? gridDim.x * blockDim.x : 256u;                  // distance between a thread's units of one trip
  const unsigned stride = DB ? gridDim.x * blockDim.x * UNR : gridDim.x * blockDim.x * UNR;
  const unsigned rb = (LG_NORM_REV & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;   // (the block's data AND its column-sum row: results unchanged)
  const unsigned first = DB ? rb * blockDim.x + threadIdx.x : rb * blockDim.x * UNR + threadIdx.x;
  f32x8 csum;
  csum.lo = f32x4{0.f, 0.f, 0.f, 0.f}; csum.hi = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntrip = (int)(((unsigned long long)tot + stride - 1) / stride);
  for (int tr = 0; tr < ntrip; ++tr) {
    const unsigned long long i0l = (unsigned long long)((LG_NORM_REV & 2) ? ntrip - 1 - tr : tr) * stride + first;
    if (i0l >= tot) continue;
    const unsigned i0 = (unsigned)i0l;
    f32x8 xs[UNR], gs[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned i = i0 + u * ustep;
      if (i < tot) { xs[u] = load8<true, (LG_NORM_NT & 1) != 0>(x, i); gs[u] = load8<G16, (LG_NORM_NT & 1) != 0>(g, i); }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned i = i0 + u * ustep;
      if (i >= tot) break;
      const int n = (int)(i / l8);
      const float* sp = stats + (long long)n * LG_NSTAT;
      const float mu = sp[0], a = sp[2], b = sp[3], mul = sp[4];
      const float m1 = bstats[n * 4], m2 = bstats[n * 4 + 1], m1l = bstats[n * 4 + 2], m2l = bstats[n * 4 + 3];
      f32x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float x0 = k < 4 ? xs[u].lo[k & 3] : xs[u].hi[k & 3];
        float xx = x0;
        if (pre_leaky) xx = lg_leaky(xx, alpha);
        const float c = (xx - mu) - mul;
        float dz = k < 4 ? gs[u].lo[k & 3] : gs[u].hi[k & 3];
        if (post_leaky) dz = (a * c + b > 0.f) ? dz : alpha * dz;
        float d = a * ((((dz - m1) - m1l) - c * m2) - c * m2l);
        if (pre_leaky) d = (x0 > 0.f) ? d : alpha * d;
        if (k < 4) o.lo[k & 3] = d; else o.hi[k & 3] = d;
      }
      if (dx) store8_f32(dx, i, o);
      if (dx16) store8_bf16<(LG_NORM_NT & 2) != 0>(dx16, i, o);
      if constexpr (DB) { csum.lo += o.lo; csum.hi += o.hi; }
    }
  }
  if constexpr (DB) {
    __shared__ f32x4 sacc[512];
    sacc[2 * threadIdx.x] = csum.lo; sacc[2 * threadIdx.x + 1] = csum.hi;
    __syncthreads();
    const int q = threadIdx.x;  // channel octet
    if (q < C8) {
      const int base = (int)(((long long)rb * 256) % C8);
      f32x4 tl = {0.f, 0.f, 0.f, 0.f}, th = {0.f, 0.f, 0.f, 0.f};
      for (int k = (q - base + C8) % C8; k < 256; k += C8) { tl += sacc[2 * k]; th += sacc[2 * k + 1]; }
      *reinterpret_cast<f32x4*>(colpart + ((long long)rb * C8 + q) * 8) = tl;
      *reinterpret_cast<f32x4*>(colpart + ((long long)rb * C8 + q) * 8 + 4) = th;
    }
  }
}

// db[c] (+)= sum_k part[k][c]: 16 columns x 64 row groups per block (1024 threads: with 16 groups a thread walked 128 rows of
// the <= 2046, 12.7 us per launch, seven launches per step), merged in group order (deterministic)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ db, int nb,
                                                            int C, int accumulate) {
  __shared__ float sr[64][17];
  const int cl = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C) {
    int k = gq;
    for (; k + 192 < nb; k += 256) {
      const float a = partial[(long long)k * C + c], b = partial[(long long)(k + 64) * C + c];
      const float e = partial[(long long)(k + 128) * C + c], f = partial[(long long)(k + 192) * C + c];
      s += (a + b) + (e + f);
    }
    for (; k < nb; k += 64) s += partial[(long long)k * C + c];
  }
  sr[gq][cl] = s;
  __syncthreads();
  if (gq == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 64; q += 4) t += (sr[q][cl] + sr[q + 1][cl]) + (sr[q + 2][cl] + sr[q + 3][cl]);
    db[c] = (accumulate ? db[c] : 0.f) + t;
  }
}

// Grid caps of the element-wise passes (blocks of 256 threads walking the map in trips).  LG_NORM_MAXBLK / LG_NORM_MAXBLK_DB: A/B of the cap
// (round 5, scripts/probe/norm_maxblk.sh): fewer, longer-lived blocks against many short ones.
inline long long ew_cap() {
  static long long v = 0;
  if (!v) { const char* e = getenv("LG_NORM_MAXBLK"); v = e && atoi(e) >= 64 ? atoi(e) : 8192; if (v > 8192) v = 8192; }
  return v;
}
inline long long db_cap(long long hard) {
  static long long v = 0;
  if (!v) { const char* e = getenv("LG_NORM_MAXBLK_DB"); v = e && atoi(e) >= 64 ? atoi(e) : hard; if (v > hard) v = hard; }
  return v;
}
// WHOLE ROUNDS (round 5).  These grids are larger than what the chip holds at once (blocks of 256 threads, 4 .. 8 per CU by their
// VGPRs), so they run in rounds — and a last partial round costs a whole trip time for a fraction of the work: the bias-sum form of
// the backward apply (71 VGPRs: 7 blocks per CU = 1792 resident) was launched as 2046 blocks = one round + 254 blocks, 14 % of a
// round at the price of one (`scripts/probe/norm_maxblk.sh`: 1533 = six per CU beat both 2046 and 1023; C3 step -0.9 %).  A grid
// beyond one round is cut to a whole number of rounds of the kernel's own residency (occupancy query, once per kernel).
// LG_NO_NORM_ROUNDS=1 = the caps alone, as before.
extern "C" int lg_device_cus(void);
#define LG_RESIDENT_BLOCKS(kern)                                                                                       \
  ([]() -> long long {                                                                                                 \
    static long long r = 0;                                                                                            \
    if (!r) {                                                                                                          \
      int per_cu = 0;                                                                                                  \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 8; \
      r = (long long)per_cu * lg_device_cus();                                                                         \
    }                                                                                                                  \
    return r;                                                                                                          \
  }())
inline long long fit_rounds(long long nb, long long resident, long long hard, int unit) {
  static int off = -1;
  if (off < 0) off = lg_env_flag("LG_NO_NORM_ROUNDS") ? 1 : 0;
  if (nb > hard) nb = hard;
  if (!off && nb > resident) nb = nb / resident * resident;
  nb = nb / unit * unit;
  return nb < unit ? unit : nb;
}
inline int nchunks(long long L) { return (int)((L + CHUNK - 1) / CHUNK); }
inline int ew_blocks(long long total4) {
  long long b = (total4 + 256 * EW_UNR - 1) / (256 * EW_UNR);
  return (int)(b < ew_cap() ? (b > 0 ? b : 1) : ew_cap());
}
inline size_t part_bytes(int B, long long L) { return ((size_t)B * nchunks(L) * 3 * sizeof(double) + 255) / 256 * 256; }
inline size_t bst_bytes(int B) { return ((size_t)B * 4 * sizeof(float) + 255) / 256 * 256; }

}  // namespace

extern "C" int lg_instnorm_stats_stride(void) { return LG_NSTAT; }

extern "C" size_t lg_instnorm_workspace_bytes(int B, long long L) {
  const size_t gs = ((size_t)B * 2 * sizeof(double) + 255) / 256 * 256;
  return part_bytes(B, L) + bst_bytes(B) + gs;
}

// stats[B][8] = {mu_hi, sigma, a, beta, mu_lo, 0,0,0} of (pre_leaky ? leaky(x) : x), x = [B][L]
extern "C" int lg_instnorm_leaky_stats_z16(const float* x, float* stats, const float* gamma, const float* beta,
                                           void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky,
                                           float alpha, void* x16_out, void* stream);
extern "C" int lg_instnorm_leaky_stats(const float* x, float* stats, const float* gamma, const float* beta,
                                       void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky,
                                       float alpha, void* stream) {
  return lg_instnorm_leaky_stats_z16(x, stats, gamma, beta, workspace, ws_bytes, B, L, pre_leaky, alpha, nullptr, stream);
}
// x16_out (may be null): also writes bf16(x) there — the bf16 activation path's copy of a conv output whose kernel had no
// fused-moments epilogue (the fp32 x can then be dropped)
extern "C" int lg_instnorm_leaky_stats_z16(const float* x, float* stats, const float* gamma, const float* beta,
                                           void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky,
                                           float alpha, void* x16_out, void* stream) {
  LG_CHECK_ARG(x && stats && gamma && beta && workspace, "lg_instnorm_leaky_stats: null pointer");
  LG_CHECK_ARG(B > 0 && B <= 65535 && L > 0 && L % 4 == 0, "lg_instnorm_leaky_stats: bad shape B=%d L=%lld", B, L);
  LG_CHECK_ARG(ws_bytes >= lg_instnorm_workspace_bytes(B, L), "lg_instnorm_leaky_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nc = nchunks(L);
  hipLaunchKernelGGL(stats_partial_kernel, dim3(nc, B), dim3(256), 0, st, x, (double*)workspace, L, nc, pre_leaky, alpha,
                     (__bf16*)x16_out);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_stats(partial)");
  hipLaunchKernelGGL(stats_final_kernel, dim3(B), dim3(64), 0, st, (const double*)workspace, stats, gamma, beta, nc);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_stats(final)");
  return LG_OK;
}

// stats from moment partials [B][nparts][3] doubles {count, mean, M2} produced by a conv epilogue (conv_halo.hip)
extern "C" int lg_instnorm_stats_finalize(const void* partials, int nparts, float* stats, const float* gamma,
                                          const float* beta, int B, void* stream) {
  LG_CHECK_ARG(partials && stats && gamma && beta && nparts > 0 && B > 0, "lg_instnorm_stats_finalize: bad arguments");
  hipLaunchKernelGGL(stats_final_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const double*)partials, stats, gamma,
                     beta, nparts);
  LG_CHECK_LAUNCH("lg_instnorm_stats_finalize");
  return LG_OK;
}

// y = [post_leaky] (a_n * ([pre_leaky](x) - mu_n) + beta) [+ skip]
extern "C" int lg_instnorm_leaky_apply(const float* x, const float* stats, const float* skip, float* y, void* y16, int B,
                                       long long L, int pre_leaky, int post_leaky, float alpha, void* stream) {
  LG_CHECK_ARG(x && stats && (y || y16), "lg_instnorm_leaky_apply: null pointer");
  LG_CHECK_ARG(B > 0 && L > 0 && L % 4 == 0 && (long long)B * L / 4 < (1LL << 31), "lg_instnorm_leaky_apply: bad shape B=%d L=%lld", B, L);
  const long long total4 = (long long)B * L / 4;
  hipLaunchKernelGGL(apply_kernel, dim3((int)fit_rounds(ew_blocks(total4), LG_RESIDENT_BLOCKS(apply_kernel), ew_cap(), 1)), dim3(256), 0, (hipStream_t)stream, x, stats, skip, y,
                     (__bf16*)y16, L / 4, total4, pre_leaky, post_leaky, alpha);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_apply");
  return LG_OK;
}

// g = dL/d(apply output before skip); writes dx = dL/dx, (accumulates) dgamma, dbeta
// g: dL/dy as fp32, or as bf16 when g_is_bf16; dx and dx16: fp32 result and/or its bf16 mirror (at least one)
constexpr int DB_MAX_BLOCKS = 2046;  // multiple of 3: gridDim*256 must be a multiple of C/4 (C/4 = 96 for 384 channels)

extern "C" size_t lg_instnorm_bwd_db_workspace_bytes(int B, long long L, int C) {
  return lg_instnorm_workspace_bytes(B, L) + (size_t)DB_MAX_BLOCKS * (size_t)C * sizeof(float);
}

extern "C" int lg_instnorm_leaky_bwd_db(const float* x, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                                        float* dgamma, float* dbeta, float* db, int C, void* workspace, size_t ws_bytes, int B,
                                        long long L, int pre_leaky, int post_leaky, float alpha, int accumulate, void* stream);

extern "C" int lg_instnorm_leaky_bwd(const float* x, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                                     float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky,
                                     int post_leaky, float alpha, int accumulate, void* stream) {
  return lg_instnorm_leaky_bwd_db(x, stats, g, g_is_bf16, dx, dx16, dgamma, dbeta, nullptr, 0, workspace, ws_bytes, B, L,
                                  pre_leaky, post_leaky, alpha, accumulate, stream);
}

// as lg_instnorm_leaky_bwd; db (may be null): db[C] (+)= column sums of dx viewed as [B*L/C][C] — the bias gradient of the
// conv layer whose output x is (C = its channel count, innermost), taken in the same pass that writes dx
extern "C" int lg_instnorm_leaky_bwd_db(const float* x, const float* stats, const void* g, int g_is_bf16, float* dx, void* dx16,
                                        float* dgamma, float* dbeta, float* db, int C, void* workspace, size_t ws_bytes, int B,
                                        long long L, int pre_leaky, int post_leaky, float alpha, int accumulate, void* stream) {
  LG_CHECK_ARG(x && stats && g && (dx || dx16) && workspace, "lg_instnorm_leaky_bwd: null pointer");
  LG_CHECK_ARG(B > 0 && B <= 65535 && L > 0 && L % 4 == 0 && (long long)B * L / 4 < (1LL << 31),
               "lg_instnorm_leaky_bwd: bad shape B=%d L=%lld", B, L);
  LG_CHECK_ARG(ws_bytes >= lg_instnorm_workspace_bytes(B, L), "lg_instnorm_leaky_bwd: workspace too small");
  if (db) {
    LG_CHECK_ARG(C > 0 && C % 4 == 0 && C / 4 <= 256 && L % C == 0 && (256 % (C / 4) == 0 || 768 % (C / 4) == 0),
                 "lg_instnorm_leaky_bwd_db: unsupported channel count C=%d (L=%lld)", C, L);
    LG_CHECK_ARG(ws_bytes >= lg_instnorm_bwd_db_workspace_bytes(B, L, C), "lg_instnorm_leaky_bwd_db: workspace too small");
  }
  hipStream_t st = (hipStream_t)stream;
  int nc = 4096 / B;  // bwd_partial blocks per sample: ~4096 long-lived blocks in all (the buffer is sized for nchunks(L))
  if (nc < 1) nc = 1;
  if (nc > nchunks(L)) nc = nchunks(L);
  char* ws = (char*)workspace;
  double* partial = (double*)ws;
  float* bstats = (float*)(ws + part_bytes(B, L));
  double* gsum = (double*)(ws + part_bytes(B, L) + bst_bytes(B));
  hipLaunchKernelGGL(bwd_partial_kernel, dim3(nc, B), dim3(256), 0, st, x, g, g_is_bf16, stats, partial, L, nc, pre_leaky,
                     post_leaky, alpha);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd(partial)");
  hipLaunchKernelGGL(bwd_final_kernel, dim3(B), dim3(64), 0, st, (const double*)partial, stats, bstats, gsum, L, nc);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd(final)");
  if (dgamma && dbeta) {
    hipLaunchKernelGGL(bwd_affine_grad_kernel, dim3(1), dim3(256), 0, st, (const double*)gsum, dgamma, dbeta, B,
                       accumulate);
    LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd(affine)");
  }
  const long long total4 = (long long)B * L / 4;
  if (!db) {
    hipLaunchKernelGGL(bwd_apply_kernel<false>, dim3((int)fit_rounds(ew_blocks(total4), LG_RESIDENT_BLOCKS(bwd_apply_kernel<false>), ew_cap(), 1)), dim3(256), 0, st, x, g, g_is_bf16, stats,
                       (const float*)bstats, dx, (__bf16*)dx16, L / 4, total4, pre_leaky, post_leaky, alpha, nullptr, 0);
    LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd(apply)");
    return LG_OK;
  }
  const int C4 = C / 4, unit = 256 % C4 == 0 ? 1 : 3;  // blocks per period of the thread -> channel map
  long long nb = fit_rounds((total4 + 255) / 256, LG_RESIDENT_BLOCKS(bwd_apply_kernel<true>), db_cap(DB_MAX_BLOCKS), unit);
  float* colpart = (float*)(ws + lg_instnorm_workspace_bytes(B, L));
  hipLaunchKernelGGL(bwd_apply_kernel<true>, dim3((int)nb), dim3(256), 0, st, x, g, g_is_bf16, stats, (const float*)bstats, dx,
                     (__bf16*)dx16, L / 4, total4, pre_leaky, post_leaky, alpha, colpart, C4);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_db(apply)");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(1024), 0, st, (const float*)colpart, db, (int)nb, C,
                     accumulate);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_db(bias)");
  return LG_OK;
}

// ---- bf16 activation path: the same ops reading the conv output z as bf16 (see the kernels above) -------------------
// skip (may be null): fp32, or bf16 when skip_is_bf16
extern "C" int lg_instnorm_leaky_apply_z16(const void* z16, const float* stats, const void* skip, int skip_is_bf16, float* y,
                                           void* y16, int B, long long L, int pre_leaky, int post_leaky, float alpha,
                                           void* stream) {
  LG_CHECK_ARG(z16 && stats && (y || y16), "lg_instnorm_leaky_apply_z16: null pointer");
  LG_CHECK_ARG(B > 0 && L > 0 && L % 8 == 0 && (long long)B * L / 8 < (1LL << 31),
               "lg_instnorm_leaky_apply_z16: bad shape B=%d L=%lld", B, L);
  const long long total8 = (long long)B * L / 8;
  long long nb = (total8 + 256 * EW8_UNR - 1) / (256 * EW8_UNR);
  nb = fit_rounds(nb, !skip ? LG_RESIDENT_BLOCKS(apply16_kernel<0>) : !skip_is_bf16 ? LG_RESIDENT_BLOCKS(apply16_kernel<1>) : LG_RESIDENT_BLOCKS(apply16_kernel<2>), ew_cap(), 1);
  hipStream_t st = (hipStream_t)stream;
  const __bf16* x = (const __bf16*)z16;
  if (!skip)
    hipLaunchKernelGGL(apply16_kernel<0>, dim3((int)nb), dim3(256), 0, st, x, stats, skip, y, (__bf16*)y16, L / 8, total8,
                       pre_leaky, post_leaky, alpha);
  else if (!skip_is_bf16)
    hipLaunchKernelGGL(apply16_kernel<1>, dim3((int)nb), dim3(256), 0, st, x, stats, skip, y, (__bf16*)y16, L / 8, total8,
                       pre_leaky, post_leaky, alpha);
  else
    hipLaunchKernelGGL(apply16_kernel<2>, dim3((int)nb), dim3(256), 0, st, x, stats, skip, y, (__bf16*)y16, L / 8, total8,
                       pre_leaky, post_leaky, alpha);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_apply_z16");
  return LG_OK;
}

// lg_instnorm_stats_finalize + lg_instnorm_leaky_apply_z16 in ONE launch (see apply16p_kernel): `partials` = the [B][nparts][3]
// moment records a *_fwd_stats conv left behind; `stats` [B][8] receives the finished records (bit-identical to the two-call form)
extern "C" int lg_instnorm_leaky_apply_z16_p(const void* z16, const void* partials, int nparts, const float* gamma, const float* beta,
                                             float* stats, const void* skip, int skip_is_bf16, float* y, void* y16, int B,
                                             long long L, int post_leaky, float alpha, void* stream) {
  LG_CHECK_ARG(z16 && partials && gamma && beta && stats && (y || y16), "lg_instnorm_leaky_apply_z16_p: null pointer");
  LG_CHECK_ARG(nparts > 0 && B > 0 && B <= 65535 && L > 0 && L % 8 == 0 && L / 8 < (1LL << 31),
               "lg_instnorm_leaky_apply_z16_p: bad shape B=%d L=%lld nparts=%d", B, L, nparts);
  const long long L8 = L / 8;
  long long bps = (L8 + 256 * EW8_UNR * 4 - 1) / (256 * EW8_UNR * 4);   // ~4 trips per block: the merge is paid once per 32 KB of z
  if (bps * B > ew_cap()) bps = ew_cap() / B;
  if (bps < 1) bps = 1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)bps, (unsigned)B);
  const __bf16* x = (const __bf16*)z16;
  const double* pr = (const double*)partials;
  if (!skip)
    hipLaunchKernelGGL(apply16p_kernel<0>, grid, dim3(256), 0, st, x, pr, nparts, gamma, beta, stats, skip, y, (__bf16*)y16, (unsigned)L8, post_leaky, alpha);
  else if (!skip_is_bf16)
    hipLaunchKernelGGL(apply16p_kernel<1>, grid, dim3(256), 0, st, x, pr, nparts, gamma, beta, stats, skip, y, (__bf16*)y16, (unsigned)L8, post_leaky, alpha);
  else
    hipLaunchKernelGGL(apply16p_kernel<2>, grid, dim3(256), 0, st, x, pr, nparts, gamma, beta, stats, skip, y, (__bf16*)y16, (unsigned)L8, post_leaky, alpha);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_apply_z16_p");
  return LG_OK;
}

extern "C" int lg_instnorm_leaky_bwd_z16_p(const void* z16, const float* stats, const void* g, int g_is_bf16, float* dx,
                                           void* dx16, float* dgamma, float* dbeta, float* db, int C, const void* partials,
                                           int nparts_in, void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky,
                                           int post_leaky, float alpha, int accumulate, void* stream);
// as lg_instnorm_leaky_bwd_db with x given as bf16 (workspace: lg_instnorm_bwd_db_workspace_bytes)
extern "C" int lg_instnorm_leaky_bwd_z16(const void* z16, const float* stats, const void* g, int g_is_bf16, float* dx,
                                         void* dx16, float* dgamma, float* dbeta, float* db, int C, void* workspace,
                                         size_t ws_bytes, int B, long long L, int pre_leaky, int post_leaky, float alpha,
                                         int accumulate, void* stream) {
  return lg_instnorm_leaky_bwd_z16_p(z16, stats, g, g_is_bf16, dx, dx16, dgamma, dbeta, db, C, nullptr, 0, workspace, ws_bytes, B, L,
                                     pre_leaky, post_leaky, alpha, accumulate, stream);
}
// partials / nparts_in (optional): the per-sample sums {sum g', sum g' c} as [B][nparts_in][2] doubles, already produced by
// the epilogue of the conv that wrote g (lg_conv2d_s2_dgrad_nf, ...): the first (partial-sums) pass over z and g is skipped
extern "C" int lg_instnorm_leaky_bwd_z16_p(const void* z16, const float* stats, const void* g, int g_is_bf16, float* dx,
                                           void* dx16, float* dgamma, float* dbeta, float* db, int C, const void* partials,
                                           int nparts_in, void* workspace, size_t ws_bytes, int B, long long L, int pre_leaky,
                                           int post_leaky, float alpha, int accumulate, void* stream) {
  LG_CHECK_ARG(z16 && stats && g && (dx || dx16) && workspace, "lg_instnorm_leaky_bwd_z16: null pointer");
  LG_CHECK_ARG(B > 0 && B <= 65535 && L > 0 && L % 8 == 0 && (long long)B * L / 8 < (1LL << 31),
               "lg_instnorm_leaky_bwd_z16: bad shape B=%d L=%lld", B, L);
  LG_CHECK_ARG(ws_bytes >= lg_instnorm_workspace_bytes(B, L), "lg_instnorm_leaky_bwd_z16: workspace too small");
  // producer-fused sums (lg_nf_accum, lg_common.h) are those of the post-LeakyReLU form only; the caller must pass the alpha
  // the producer used (the record carries none)
  LG_CHECK_ARG(!(partials && nparts_in > 0) || (pre_leaky == 0 && post_leaky == 1),
               "lg_instnorm_leaky_bwd_z16_p: fused partial sums exist for pre_leaky=0, post_leaky=1 only (got %d, %d)", pre_leaky, post_leaky);
  if (db) {
    LG_CHECK_ARG(C > 0 && C % 8 == 0 && C / 8 <= 256 && L % C == 0 && (256 % (C / 8) == 0 || 768 % (C / 8) == 0),
                 "lg_instnorm_leaky_bwd_z16: unsupported channel count C=%d (L=%lld)", C, L);
    LG_CHECK_ARG(ws_bytes >= lg_instnorm_bwd_db_workspace_bytes(B, L, C), "lg_instnorm_leaky_bwd_z16: workspace too small");
  }
  hipStream_t st = (hipStream_t)stream;
  const __bf16* x = (const __bf16*)z16;
  int nc = 4096 / B;
  if (nc < 1) nc = 1;
  const int nch8 = (int)((L + 2 * CHUNK - 1) / (2 * CHUNK));
  if (nc > nch8) nc = nch8;   // nch8 <= nchunks(L): the partial buffer is large enough
  char* ws = (char*)workspace;
  double* partial = (double*)ws;
  float* bstats = (float*)(ws + part_bytes(B, L));
  double* gsum = (double*)(ws + part_bytes(B, L) + bst_bytes(B));
  if (partials && nparts_in > 0) {  // sums fused into the producer of g
    partial = (double*)partials;
    nc = nparts_in;
  } else {
    if (g_is_bf16)
      hipLaunchKernelGGL(bwd_partial16_kernel<true>, dim3(nc, B), dim3(256), 0, st, x, g, stats, partial, L, nc, pre_leaky,
                         post_leaky, alpha);
    else
      hipLaunchKernelGGL(bwd_partial16_kernel<false>, dim3(nc, B), dim3(256), 0, st, x, g, stats, partial, L, nc, pre_leaky,
                         post_leaky, alpha);
    LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_z16(partial)");
  }
  hipLaunchKernelGGL(bwd_final_kernel, dim3(B), dim3(64), 0, st, (const double*)partial, stats, bstats, gsum, L, nc);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_z16(final)");
  if (dgamma && dbeta) {
    hipLaunchKernelGGL(bwd_affine_grad_kernel, dim3(1), dim3(256), 0, st, (const double*)gsum, dgamma, dbeta, B, accumulate);
    LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_z16(affine)");
  }
  const long long total8 = (long long)B * L / 8;
  if (!db) {
    long long nb = (total8 + 256 * EW8_UNR - 1) / (256 * EW8_UNR);
    nb = fit_rounds(nb, g_is_bf16 ? LG_RESIDENT_BLOCKS((bwd_apply16_kernel<false, true>)) : LG_RESIDENT_BLOCKS((bwd_apply16_kernel<false, false>)), ew_cap(), 1);
    if (g_is_bf16)
      hipLaunchKernelGGL((bwd_apply16_kernel<false, true>), dim3((int)nb), dim3(256), 0, st, x, g, stats, (const float*)bstats, dx,
                         (__bf16*)dx16, L / 8, total8, pre_leaky, post_leaky, alpha, (float*)nullptr, 0);
    else
      hipLaunchKernelGGL((bwd_apply16_kernel<false, false>), dim3((int)nb), dim3(256), 0, st, x, g, stats, (const float*)bstats, dx,
                         (__bf16*)dx16, L / 8, total8, pre_leaky, post_leaky, alpha, (float*)nullptr, 0);
    LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_z16(apply)");
    return LG_OK;
  }
  const int C8 = C / 8, unit = 256 % C8 == 0 ? 1 : 3;
  long long nb = (total8 + 511) / 512;   // two units per thread per trip
  nb = fit_rounds(nb, g_is_bf16 ? LG_RESIDENT_BLOCKS((bwd_apply16_kernel<true, true>)) : LG_RESIDENT_BLOCKS((bwd_apply16_kernel<true, false>)), db_cap(DB_MAX_BLOCKS), unit);
  float* colpart = (float*)(ws + lg_instnorm_workspace_bytes(B, L));   // [nb][C] floats, nb <= DB_MAX_BLOCKS (+2)
  if (g_is_bf16)
    hipLaunchKernelGGL((bwd_apply16_kernel<true, true>), dim3((int)nb), dim3(256), 0, st, x, g, stats, (const float*)bstats, dx,
                       (__bf16*)dx16, L / 8, total8, pre_leaky, post_leaky, alpha, colpart, C8);
  else
    hipLaunchKernelGGL((bwd_apply16_kernel<true, false>), dim3((int)nb), dim3(256), 0, st, x, g, stats, (const float*)bstats, dx,
                       (__bf16*)dx16, L / 8, total8, pre_leaky, post_leaky, alpha, colpart, C8);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_z16(apply+bias)");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(1024), 0, st, (const float*)colpart, db, (int)nb, C,
                     accumulate);
  LG_CHECK_LAUNCH("lg_instnorm_leaky_bwd_z16(bias)");
  return LG_OK;
}

// The norm backward WITHOUT its apply pass: coef[B][8] (see bwd_coef_kernel) from the statistics records and the sums
// {sum g', sum g' c} the producer of g left as [B][nparts][2] doubles (lg_*_dgrad_nf).  The consumer of dz — the next data-gradient
// conv (lg_convT_s2_dgrad_bn) — forms dz = a (g' - m1 - c m2') while it stages its operand: dz is never written.  For tapes that
// ask the level for no weight gradient (dz then has ONE reader).  instance.py:105-128 differentiated; eager_trainer.py:158-163.
extern "C" int lg_instnorm_bwd_coef(const float* stats, const void* partials, int nparts, float* coef, int B, long long L,
                                    void* stream) {
  LG_CHECK_ARG(stats && partials && coef && nparts > 0 && B > 0 && B <= 65535 && L > 0, "lg_instnorm_bwd_coef: bad arguments");
  hipLaunchKernelGGL(bwd_coef_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const double*)partials, stats, coef, L, nparts);
  LG_CHECK_LAUNCH("lg_instnorm_bwd_coef");
  return LG_OK;
}
