// Implicit-GEMM convolution kernels for gfx950 (MFMA 32x32, wave64).
//
// One kernel body serves every 5x5 conv-shaped contraction of the LittleGAN step
// (reference: /root/reference/model.py:15 Conv2D s2 SAME, :39-40 Conv2DTranspose s2 SAME,
// :86-87 Conv2DTranspose s1 SAME + tanh, and their data-gradients):
//
//   DOWN  : out[n,y,x,:]        = sum_{ky,kx} src[n,2y+ky-1,2x+kx-1,:] . Wd[ky,kx]   (conv fwd / convT dgrad)
//   UP    : out[n,2y+py,2x+px,:] = sum_{taps of parity class (py,px)} src[n,y+dy,x+dx,:] . Wu[ky,kx]
//                                                                                   (convT fwd / conv dgrad)
//   S1T   : out[n,y,x,:]        = sum_{ky,kx} src[n,y+2-ky,x+2-kx,:] . Wu[ky,kx]     (stride-1 convT fwd)
//   PATCH : 3-channel source, out[n,y,x,:] = sum_ky  src_row_segment(15 floats) . Wp[ky]   (conv1 fwd, s1 dgrad)
//
// GEMM view: M = pixels of the "M grid", N = output channels, K = taps x source channels.
// A (activations) is gathered from fp32 NHWC global memory, optionally converted to bf16, and staged
// in LDS as rows of KCH*32 payload bytes (+16 B pad => conflict-free ds_read_b128).  B (weights) comes
// pre-packed in MFMA fragment order in the compute dtype (pack.hip).  Lane (r,h) of a wave reads
// 16-byte chunk (2q+h) of row r: for f32 that feeds four v_mfma_f32_32x32x2_f32 (k = 8q+4h+j), for
// bf16 one v_mfma_f32_32x32x16_bf16; A and B use the same k assignment, so the sum is exact.
// Register-prefetch double buffering: tile k+1's global loads are in flight during tile k's MFMAs.
#include <stdlib.h>
#include "lg_common.h"

namespace {

enum { MODE_DOWN = 0, MODE_UP = 1, MODE_S1T = 2, MODE_PATCH = 3 };

struct ConvParams {
  const float* src;
  const char* wp;
  const float* bias;
  float* out;
  __bf16* out16;  // if non-null the result is written as bf16 HERE instead of fp32 to `out`
  int B, Hs, Ws, Cs;  // source tensor [B,Hs,Ws,Cs]
  int Hm, Wm, M;      // M grid (rows = B*Hm*Wm)
  int Ho, Wo, N, Npad;
  int act;            // 0 none, 1 tanh
  int pstride, ppad;  // PATCH mode: source stride and pad-before
  int ntn;            // number of N tiles
};

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int ESZ = 4; };
template <> struct DT<__bf16> { static constexpr int ESZ = 2; };

__device__ __forceinline__ void tap_info(int mode, int cls, int t, int& dy, int& dx, int& widx) {
  if (mode == MODE_DOWN) {
    const int ky = t / 5, kx = t - ky * 5;
    dy = ky - 1; dx = kx - 1; widx = t;
  } else if (mode == MODE_S1T) {
    const int ky = t / 5, kx = t - ky * 5;
    dy = 2 - ky; dx = 2 - kx; widx = t;
  } else {  // UP: class (py,px); py==0 -> ky in {1,3}, py==1 -> ky in {0,2,4}; iy = qy + (py+1-ky)/2
    const int py = cls >> 1, px = cls & 1;
    const int nkx = px ? 3 : 2;
    const int a = t / nkx, b = t - a * nkx;
    const int ky = py ? 2 * a : 2 * a + 1;
    const int kx = px ? 2 * b : 2 * b + 1;
    dy = (py + 1 - ky) / 2;  // exact: numerator is even
    dx = (px + 1 - kx) / 2;
    widx = ky * 5 + kx;
  }
}

template <typename T, int MODE, int KCH, int WAVES_M, int WAVES_N, int MT, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int ESZ = DT<T>::ESZ;
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
  constexpr int ROWB = KCH * 32 + 16;           // LDS row bytes
  constexpr int KC = KCH * 32 / ESZ;            // k elements per tile
  constexpr bool PATCH = (MODE == MODE_PATCH);
  // A staging geometry (source is fp32)
  constexpr int LPR = PATCH ? 4 : KC / 4;       // threads per row
  constexpr int RPP = 256 / LPR;                // rows per pass
  constexpr int PA = BM / RPP;                  // passes
  static_assert(PA >= 1 && BM % RPP == 0, "bad A staging geometry");
  constexpr int BCH = BN * KCH * 2;             // 16-B chunks in the B tile
  constexpr int PB = (BCH + 255) / 256;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA0 = smem;
  char* sB0 = smem + 2 * A_BYTES;
  int* s_out = reinterpret_cast<int*>(smem + 2 * A_BYTES + 2 * B_BYTES);  // [BM] out pixel index or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  const int cls = (MODE == MODE_UP) ? (3 - (int)blockIdx.y) : 0;  // heavy (9-tap) class first
  const int nblk = gridDim.x;
  const int lb = lg_xcd_remap(blockIdx.x, nblk);
  const int tile_n = lb % p.ntn, tile_m = lb / p.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int HWm = p.Hm * p.Wm;
  const int sstr = (MODE == MODE_DOWN) ? 2 : (PATCH ? p.pstride : 1);
  const int py = cls >> 1, px = cls & 1;

  // ---- per-block row tables -------------------------------------------------
  for (int i = tid; i < BM; i += 256) {
    const int m = m0 + i;
    int o = -1;
    if (m < p.M) {
      const int n = m / HWm, rem = m - n * HWm;
      const int y = rem / p.Wm, x = rem - y * p.Wm;
      const int oy = (MODE == MODE_UP) ? 2 * y + py : y;
      const int ox = (MODE == MODE_UP) ? 2 * x + px : x;
      o = (n * p.Ho + oy) * p.Wo + ox;
    }
    s_out[i] = o;
  }
  // rows this thread stages
  const int arow = tid / LPR, alc = tid % LPR;
  int rowpix[PA], rowy[PA], rowx[PA];  // n*Hs*Ws (or -1), sstr*y, sstr*x
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int m = m0 + q * RPP + arow;
    if (m < p.M) {
      const int n = m / HWm, rem = m - n * HWm;
      const int y = rem / p.Wm, x = rem - y * p.Wm;
      rowpix[q] = n * p.Hs * p.Ws; rowy[q] = sstr * y; rowx[q] = sstr * x;
    } else {
      rowpix[q] = -1; rowy[q] = 0; rowx[q] = 0;
    }
  }

  int ntaps;
  if (PATCH) ntaps = 5;
  else if (MODE == MODE_UP) ntaps = (py ? 3 : 2) * (px ? 3 : 2);
  else ntaps = 25;
  const int cpt = PATCH ? 1 : p.Cs / KC;  // k tiles per tap
  const int nk = ntaps * cpt;

  f32x4 ra[PA];
  u32x4 rb[PB];

  auto load_tile = [&](int it) {
    const int t = it / cpt, cc = it - t * cpt;
    if constexpr (PATCH) {
      const int ky = t;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int sy = rowy[q] + ky - p.ppad;
        if (rowpix[q] >= 0 && (unsigned)sy < (unsigned)p.Hs) {
          const int sx0 = rowx[q] - p.ppad;
          const float* base = p.src + ((long long)rowpix[q] + (long long)sy * p.Ws) * 3;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = alc * 4 + e;         // pseudo channel kx*3+c, 15 = pad
            const int sx = sx0 + j / 3;
            if (j < 15 && (unsigned)sx < (unsigned)p.Ws) v[e] = base[sx0 * 3 + j];
          }
        }
        ra[q] = v;
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int c = q * 256 + tid;
        if (BCH % 256 == 0 || c < BCH) {
          const int row = c / (KCH * 2), ch = c % (KCH * 2);
          const char* g = p.wp + ((long long)(ky * p.Npad + n0 + row) * 16) * ESZ + ch * 16;
          rb[q] = *reinterpret_cast<const u32x4*>(g);
        }
      }
    } else {
      int dy, dx, widx;
      tap_info(MODE, cls, t, dy, dx, widx);
      const int c0 = cc * KC;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int sy = rowy[q] + dy, sx = rowx[q] + dx;
        if (rowpix[q] >= 0 && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws) {
          const float* g = p.src + ((long long)rowpix[q] + (long long)sy * p.Ws + sx) * p.Cs + c0 + alc * 4;
          v = *reinterpret_cast<const f32x4*>(g);
        }
        ra[q] = v;
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int c = q * 256 + tid;
        if (BCH % 256 == 0 || c < BCH) {
          const int row = c / (KCH * 2), ch = c % (KCH * 2);
          // fragment-ordered pack: 16-B piece (row n, k-chunk kc16) lives in block (n/32, kc16/2) at lane 32*(kc16&1) + n%32
          const int n = n0 + row, kc16 = (c0 * ESZ) / 16 + ch;
          const char* g = p.wp + ((((long long)widx * (p.Npad >> 5) + (n >> 5)) * (p.Cs * ESZ / 32) + (kc16 >> 1)) * 64 +
                                  (kc16 & 1) * 32 + (n & 31)) * 16;
          rb[q] = *reinterpret_cast<const u32x4*>(g);
        }
      }
    }
  };

  auto store_tile = [&](int buf) {
    char* sA = sA0 + buf * A_BYTES;
    char* sB = sB0 + buf * B_BYTES;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int row = q * RPP + arow;
      if constexpr (ESZ == 4) {
        *reinterpret_cast<f32x4*>(sA + row * ROWB + alc * 16) = ra[q];
      } else {
        bf16x4 w;
        w[0] = (__bf16)ra[q][0]; w[1] = (__bf16)ra[q][1]; w[2] = (__bf16)ra[q][2]; w[3] = (__bf16)ra[q][3];
        *reinterpret_cast<bf16x4*>(sA + row * ROWB + alc * 8) = w;
      }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int c = q * 256 + tid;
      if (BCH % 256 == 0 || c < BCH) {
        const int row = c / (KCH * 2), ch = c % (KCH * 2);
        *reinterpret_cast<u32x4*>(sB + row * ROWB + ch * 16) = rb[q];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) load_tile(it + 1);
    const char* sA = sA0 + buf * A_BYTES + (wm * MT * 32 + r) * ROWB + h * 16;
    const char* sB = sB0 + buf * B_BYTES + (wn * NT * 32 + r) * ROWB + h * 16;
#pragma unroll
    for (int q = 0; q < KCH; ++q) {
      if constexpr (ESZ == 4) {
        f32x4 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * ROWB + q * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * ROWB + q * 32);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      } else {
        bf16x8 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sA + i * 32 * ROWB + q * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sB + j * 32 * ROWB + q * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (it + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -------------
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + (wn * NT + j) * 32 + r;
    const bool cok = col < p.N;
    const float bv = (cok && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int o = s_out[row];
        if (cok && o >= 0) {
          float v = acc[i][j][e] + bv;
          if (p.act == 1) v = tanhf(v);
          if (p.out16) p.out16[(long long)o * p.N + col] = (__bf16)v;
          else p.out[(long long)o * p.N + col] = v;
        }
      }
    }
  }
}

template <typename T, int MODE, int KCH, int WAVES_M, int WAVES_N, int MT, int NT>
int launch(const ConvParams& p, hipStream_t st) {
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32, ROWB = KCH * 32 + 16;
  const size_t lds = 2 * (size_t)(BM + BN) * ROWB + BM * sizeof(int);
  ConvParams q = p;
  q.ntn = p.Npad / BN;
  const int ntm = lg_cdiv(p.M, BM);
  dim3 grid(ntm * q.ntn, MODE == MODE_UP ? 4 : 1);
  auto kern = conv_igemm_kernel<T, MODE, KCH, WAVES_M, WAVES_N, MT, NT>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
  lg_note_kernel(MODE == MODE_PATCH ? "conv_igemm_kernel<PATCH>" : MODE == MODE_UP ? "conv_igemm_kernel<UP>" : MODE == MODE_DOWN ? "conv_igemm_kernel<DOWN>" : "conv_igemm_kernel<S1T>");
  return LG_OK;
}

// BN choice shared with pack.hip / host: 128 if N%128==0, 64 if N%64==0, else 32 (N padded up).
template <typename T, int MODE, int KCH>
int dispatch_bn(const ConvParams& p, hipStream_t st) {
  if (p.Npad % 128 == 0) return launch<T, MODE, KCH, 2, 2, 2, 2>(p, st);
  if (p.Npad % 64 == 0) return launch<T, MODE, KCH, 2, 2, 2, 1>(p, st);
  return launch<T, MODE, KCH, 4, 1, 1, 1>(p, st);
}

template <int MODE>
int dispatch_dtype(const ConvParams& p, int dtype, hipStream_t st) {
  if (dtype == LG_DT_F32) {
    if (p.Cs % 32 == 0) return dispatch_bn<float, MODE, 4>(p, st);
    return LG_ERR_UNSUPPORTED;
  }
  if (p.Cs % 64 == 0) return dispatch_bn<__bf16, MODE, 4>(p, st);
  if (p.Cs % 32 == 0) return dispatch_bn<__bf16, MODE, 2>(p, st);
  return LG_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int lg_conv_halo_try(int mode, int dtype, const float* src, const void* src16, const void* wpack,
                                const float* bias, float* out, void* out16, int B, int Hm, int Wm, int Cs, int N, int act, void* spart,
                                size_t spart_bytes, int* nparts_out, void* stream);

static bool halo_enabled() {
  static int v = -1;
  if (v < 0) v = lg_env_flag("LG_NO_HALO") ? 0 : 1;  // A/B switch: LG_NO_HALO=1 forces the per-tap gather kernel
  return v == 1;
}

extern "C" int lg_npad(int n) {
  if (n % 128 == 0) return n;
  if (n % 64 == 0) return n;
  return (n + 31) / 32 * 32;
}

// Generic entry used by the typed C-ABI wrappers in capi.hip.
//   mode 0 DOWN: src [B,2Hm,2Wm,Cs] -> out [B,Hm,Wm,N]
//   mode 1 UP  : src [B,Hm,Wm,Cs]   -> out [B,2Hm,2Wm,N]
//   mode 2 S1T : src [B,Hm,Wm,Cs]   -> out [B,Hm,Wm,N]   (+ optional tanh)
//   mode 3 PATCH: src [B,Hs,Ws,3], stride s, pad p -> out [B,Hm,Wm,N]; wp = [5][Npad][16]
extern "C" int lg_conv_igemm_ex(int mode, int dtype, const float* src, const void* src16, const void* wpack,
                                const float* bias, float* out, void* out16, int B, int Hm, int Wm, int Cs, int N, int act,
                                int pstride, int ppad, void* spart, size_t spart_bytes, int* nparts_out, void* stream);
extern "C" int lg_conv_up4_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                               int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream);
extern "C" int lg_conv_up3_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                               int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream);
extern "C" int lg_conv_down3_try(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm,
                                 int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream);

extern "C" int lg_conv_igemm(int mode, int dtype, const float* src, const void* wpack, const float* bias, float* out,
                             int B, int Hm, int Wm, int Cs, int N, int act, int pstride, int ppad, void* stream) {
  return lg_conv_igemm_ex(mode, dtype, src, nullptr, wpack, bias, out, nullptr, B, Hm, Wm, Cs, N, act, pstride, ppad,
                          nullptr, 0, nullptr, stream);
}

// Same, with (a) an optional bf16 mirror `src16` of the source tensor (used by the bf16 halo kernel instead of
// re-reading and re-rounding fp32; `src` must still be valid for the fallback kernels) and (b) optional fused
// InstanceNorm moment partials (see lg_conv_halo_try); *nparts_out == 0 means the chosen kernel did not produce them.
// out16 (optional): write the result as bf16 there instead of fp32 to `out` (data gradients of the bf16 path).
extern "C" int lg_conv_igemm_ex(int mode, int dtype, const float* src, const void* src16, const void* wpack,
                                const float* bias, float* out, void* out16, int B, int Hm, int Wm, int Cs, int N, int act,
                                int pstride, int ppad, void* spart, size_t spart_bytes, int* nparts_out, void* stream) {
  if (nparts_out) *nparts_out = 0;
  LG_CHECK_ARG((src || src16) && wpack && (out || out16), "lg_conv_igemm: null pointer");
  LG_CHECK_ARG(B > 0 && Hm > 0 && Wm > 0 && Cs > 0 && N > 0, "lg_conv_igemm: bad shape B=%d Hm=%d Wm=%d Cs=%d N=%d", B, Hm, Wm, Cs, N);
  LG_CHECK_ARG(dtype == LG_DT_F32 || dtype == LG_DT_BF16, "lg_conv_igemm: bad dtype %d", dtype);
  // the exact-f32 kernels read the fp32 source and write the fp32 destination: bf16 mirrors alone are a bf16-dtype contract (a null
  // fp32 pointer here would be dereferenced on the device)
  LG_CHECK_ARG(dtype == LG_DT_BF16 || (src && out && !out16), "lg_conv_igemm: dtype f32 needs the fp32 source and destination (bf16 mirrors / bf16 output belong to dtype bf16)");
  LG_CHECK_ARG((long long)B * Hm * Wm < (1ll << 31) / 4, "lg_conv_igemm: M grid too large");
  ConvParams p{};
  p.src = src; p.wp = (const char*)wpack; p.bias = bias; p.out = out; p.out16 = (__bf16*)out16;
  p.B = B; p.Cs = Cs; p.Hm = Hm; p.Wm = Wm; p.M = B * Hm * Wm; p.N = N; p.Npad = lg_npad(N); p.act = act;
  p.pstride = pstride; p.ppad = ppad;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (mode == MODE_DOWN && dtype == LG_DT_BF16 && src16 && out16 && act == 0 && halo_enabled()) {
    // software-pipelined persistent kernel (conv_down3.hip) for the bf16 activation path where its tiling applies
    rc = lg_conv_down3_try(src16, wpack, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  if (mode == MODE_UP && dtype == LG_DT_BF16 && src16 && out16 && act == 0 && halo_enabled()) {
    // resident-halo persistent kernel with the four parity classes on concurrent waves (conv_up3.hip): small-N layers
    rc = lg_conv_up3_try(src16, wpack, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
    // the wide layers (N % 128 == 0) on 8 x 16 source tiles: blocks bound to one parity class (conv_up4.hip)
    rc = lg_conv_up4_try(src16, wpack, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  if (mode != MODE_PATCH && halo_enabled()) {  // LDS halo-tile kernel where the tiling covers the shape
    rc = lg_conv_halo_try(mode, dtype, src, src16, wpack, bias, out, out16, B, Hm, Wm, Cs, N, act, spart, spart_bytes,
                          nparts_out, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  LG_CHECK_ARG(src, "lg_conv_igemm: this shape needs the per-tap gather kernel, which reads the fp32 source (got only bf16)");
  switch (mode) {
    case MODE_DOWN:
      p.Hs = 2 * Hm; p.Ws = 2 * Wm; p.Ho = Hm; p.Wo = Wm;
      rc = dispatch_dtype<MODE_DOWN>(p, dtype, st);
      break;
    case MODE_UP:
      p.Hs = Hm; p.Ws = Wm; p.Ho = 2 * Hm; p.Wo = 2 * Wm;
      rc = dispatch_dtype<MODE_UP>(p, dtype, st);
      break;
    case MODE_S1T:
      p.Hs = Hm; p.Ws = Wm; p.Ho = Hm; p.Wo = Wm;
      rc = dispatch_dtype<MODE_S1T>(p, dtype, st);
      break;
    case MODE_PATCH:
      LG_CHECK_ARG(Cs == 3 && (pstride == 1 || pstride == 2), "lg_conv_igemm: PATCH needs Cs==3, stride 1|2");
      p.Hs = pstride * Hm; p.Ws = pstride * Wm; p.Ho = Hm; p.Wo = Wm;
      if (dtype == LG_DT_F32) rc = dispatch_bn<float, MODE_PATCH, 2>(p, st);
      else rc = dispatch_bn<__bf16, MODE_PATCH, 1>(p, st);
      break;
    default:
      lg_set_error("lg_conv_igemm: bad mode %d", mode);
      return LG_ERR_ARG;
  }
  if (rc == LG_ERR_UNSUPPORTED) {
    lg_set_error("lg_conv_igemm: unsupported channel count Cs=%d (need multiple of 32)", Cs);
    return rc;
  }
  LG_CHECK_LAUNCH("lg_conv_igemm");
  return LG_OK;
}
