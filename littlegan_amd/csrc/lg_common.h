// Shared device/host helpers for the LittleGAN gfx950 kernels.
// Wavefront = 64, MFMA 32x32 tiles, LDS rows padded by 16 B (conflict-free ds_read_b128).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LG_OK 0
#define LG_ERR_ARG (-1)
#define LG_ERR_LAUNCH (-2)
#define LG_ERR_UNSUPPORTED (-3)

#define LG_DT_F32 0
#define LG_DT_BF16 1

extern "C" void lg_set_error(const char* fmt, ...);
// Names the kernel template a conv / weight-gradient entry point has just launched (thread-local, static strings only):
// bench.py reads it back through lg_last_kernel() so that its roofline line names a KERNEL, not a class of kernels.
extern "C" void lg_note_kernel(const char* name);
extern "C" unsigned long long* lg_clock_census(void);   // runtime.hip: buffer of the in-kernel clock census, or null

// 1 if the environment variable is set to a non-empty value; read ONCE per process at its first use (runtime.hip), so a
// *_supported query and the launch it promises always agree.  `name` must be a string literal.
extern "C" int lg_env_flag(const char* name);
// CUs the persistent kernels may fill: device CUs minus those reserved for communication kernels (lg_set_reserved_cus)
extern "C" int lg_grid_cus(void);

#define LG_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      lg_set_error(__VA_ARGS__);           \
      return LG_ERR_ARG;                   \
    }                                      \
  } while (0)

#define LG_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      lg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return LG_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

// ---- XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin) ----
// Consecutive LOGICAL ids land on one XCD so neighbouring tiles share its L2.
__device__ __forceinline__ int lg_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float lg_leaky(float x, float a) { return x > 0.f ? x : a * x; }

// wave64 sum via DPP/shuffles
__device__ __forceinline__ float lg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double lg_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of NV floats (blockDim.x multiple of 64, <= 1024); result valid in thread 0.
template <int NV>
__device__ __forceinline__ void lg_block_sum(float (&v)[NV], float* smem /* >= NV*16 floats */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = lg_wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) smem[i * 16 + wid] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += smem[i * 16 + w];
      v[i] = s;
    }
  }
}

// block-wide sum of NV doubles; result valid in thread 0.  smem >= NV*16 doubles.
template <int NV>
__device__ __forceinline__ void lg_block_sum_d(double (&v)[NV], double* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = lg_wave_sum_d(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) smem[i * 16 + wid] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += smem[i * 16 + w];
      v[i] = s;
    }
  }
}

// ---- norm-backward sums fused into the epilogue of the data-gradient conv that PRODUCES the gradient g -------------
// The InstanceNormalization backward of the layer below needs, per sample, S1 = sum g' and S2 = sum g'*c with
// c = z - mu, g' = LeakyReLU'(a*c + b) * g  (norm.hip, bwd_partial).  z, the statistics record and g (as it is STORED, i.e.
// bf16) are all at hand where the conv writes g, so the separate pass that re-read z and g is dropped.
// part: [B][nparts][2] doubles, one record per (sample, block tile), merged by bwd_final_kernel.
struct LgNormFuse {
  const __bf16* z;     // raw conv output of the layer the gradient belongs to, same shape / layout as g
  const float* stats;  // its statistics records [B][8] = {mu_hi, sigma, a, beta, mu_lo, ...}
  double* part;
  float alpha;
  int nparts;
};
// 8 bf16 z -> 8 bf16 h, exactly apply16_kernel (norm.hip): t = a*((z - mu) - mul) + b; leaky; RNE
__device__ __forceinline__ u32x4 lg_norm8(const u32x4 z8, float mu, float mul, float a, float b, float alpha) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t0 = __builtin_bit_cast(float, z8[k] << 16), t1 = __builtin_bit_cast(float, z8[k] & 0xffff0000u);
    t0 = a * ((t0 - mu) - mul) + b; t1 = a * ((t1 - mu) - mul) + b;
    t0 = lg_leaky(t0, alpha); t1 = lg_leaky(t1, alpha);
    // ONE v_cvt_pk_bf16_f32 per pair (RNE, as the scalar cast): converting the halves separately and or-ing them together costs 4
    typedef __bf16 lg_bf16x2 __attribute__((ext_vector_type(2)));
#ifdef LG_NORM8_OLD   // A/B builds only
    const __bf16 h0 = (__bf16)t0, h1 = (__bf16)t1;
    o[k] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
#else
    o[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{t0, t1}, lg_bf16x2));
#endif
  }
  return o;
}

// Per-item records read inside a persistent loop (statistics, norm-backward coefficients: written by an EARLIER kernel, constant
// for this one): read through the constant address space.  A plain uniform load there is scalarised by hipcc only while it can
// prove that no store of the kernel may alias it — true in the prologue, false inside the item loop behind the output stores — and
// otherwise becomes a VECTOR load whose value is needed at once: `s_waitcnt vmcnt(0)` at every item boundary, which also drains the
// weight-fragment ring (seen in the .s of conv_down3's normalising forms, round 4).  From the constant space it is an s_load.
typedef const float __attribute__((address_space(4)))* lg_const_f32p;
__device__ __forceinline__ lg_const_f32p lg_as_const(const float* p) { return (lg_const_f32p)(p); }
// ... and every field of such a record is made wave-uniform EXPLICITLY (v_readfirstlane; folds away behind an s_load): the record then
// lives in SGPRs whatever load the compiler chose, and the arithmetic that consumes it takes scalar operands.  Round 5: with
// the record as per-lane VGPR copies (the vector-load build of conv_down3's BWDNORM form) hipcc paired the fields in 64-bit registers and
// formed `g' - m1` as a PACKED fp32 subtraction that selects the pair's high register for both results; on gfx950 with a second wave on
// the SIMD the LOW result of that instruction intermittently came out WITHOUT the subtraction (82 of 82 wrong operand elements read back
// through one-hot weights: an even element, bit-equal to the value with m1 not subtracted) — launch-to-launch different outputs.  The
// builds with scalar operands (s_load or readfirstlane), with the fields in the LOW registers of their pairs, or without packed fp32
// instructions never showed it; memory ordering is not involved (full waits in front of every instruction leave it in place).
__device__ __forceinline__ float lg_uniform(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

// 8 bf16 z, 8 bf16 g -> 8 bf16 dz of the InstanceNorm + LeakyReLU backward, exactly bwd_apply16_kernel (norm.hip) in its
// post-LeakyReLU form: c = (z - mu) - mul; g' = (a c + b > 0) ? g : alpha g; dz = a ((((g' - m1) - m1l) - c m2) - c m2l); RNE.
// co = the per-sample record {mu, mul, a, b, m1, m2, m1l, m2l} of lg_instnorm_bwd_coef
struct LgBwdCoef { float mu, mul, a, b, m1, m2, m1l, m2l; };
__device__ __forceinline__ u32x4 lg_bwdnorm8(const u32x4 z8, const u32x4 g8, const LgBwdCoef& co, float alpha) {
  u32x4 o;
  typedef __bf16 lg_bf16x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float z0 = __builtin_bit_cast(float, z8[k] << 16), z1 = __builtin_bit_cast(float, z8[k] & 0xffff0000u);
    const float g0 = __builtin_bit_cast(float, g8[k] << 16), g1 = __builtin_bit_cast(float, g8[k] & 0xffff0000u);
    const float c0 = (z0 - co.mu) - co.mul, c1 = (z1 - co.mu) - co.mul;
    const float p0 = (co.a * c0 + co.b > 0.f) ? g0 : alpha * g0, p1 = (co.a * c1 + co.b > 0.f) ? g1 : alpha * g1;
    const float d0 = co.a * ((((p0 - co.m1) - co.m1l) - c0 * co.m2) - c0 * co.m2l);
    const float d1 = co.a * ((((p1 - co.m1) - co.m1l) - c1 * co.m2) - c1 * co.m2l);
    o[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{d0, d1}, lg_bf16x2_));
  }
  return o;
}

// 8 consecutive channels of one pixel: g8 / z8 = 16 B of bf16 each
__device__ __forceinline__ void lg_nf_accum(const u32x4 g8, const u32x4 z8, float mu, float mul, float a, float b, float alpha,
                                            float& s1, float& s2) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float g0 = __builtin_bit_cast(float, g8[k] << 16), g1 = __builtin_bit_cast(float, g8[k] & 0xffff0000u);
    const float z0 = __builtin_bit_cast(float, z8[k] << 16), z1 = __builtin_bit_cast(float, z8[k] & 0xffff0000u);
    const float c0 = (z0 - mu) - mul, c1 = (z1 - mu) - mul;
    const float p0 = (a * c0 + b > 0.f) ? g0 : alpha * g0, p1 = (a * c1 + b > 0.f) ? g1 : alpha * g1;
    s1 += p0; s1 += p1;
    s2 = __builtin_fmaf(p0, c0, s2); s2 = __builtin_fmaf(p1, c1, s2);
  }
}

static inline int lg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
