// Halo-tile implicit-GEMM convolution for gfx950: the "LDS staging of image tiles" of the north star.
//
// Same contractions and operand conventions as conv_igemm.hip (modes DOWN / UP / S1T), but the A operand is
// no longer re-gathered from global memory for every filter tap.  A block owns a TH x TW tile of the M grid
// (or NI whole images when the map is smaller than 128 pixels).  For each chunk of KC source channels it
// stages the source HALO of that tile once —  (s*TH+E) x (s*TW+E) pixels, E = 2 (UP), 3 (DOWN), 4 (S1T),
// zero-filled outside the image (TF "SAME") — and then runs ALL taps out of LDS: the MFMA A fragment of
// (row m, tap t) is halo row  hb(m) + dy_t*HW + dx_t,  a per-lane base plus a wave-uniform offset.
// Global->LDS activation traffic drops by the tap-reuse factor (UP 9x..4x, DOWN 4.8x, S1T 13x); only the
// weight tile (B operand, [BN][KC] per tap, double-buffered, register-prefetched) is streamed per tap.
// Falls back (LG_ERR_UNSUPPORTED) to the per-tap gather kernel for shapes the tiling does not cover.
#include "lg_common.h"

namespace {

enum { MODE_DOWN = 0, MODE_UP = 1, MODE_S1T = 2 };

struct HaloParams {
  const float* src;
  const char* wp;
  const float* bias;
  float* out;
  int B, Hs, Ws, Cs;
  int Hm, Wm;
  int Ho, Wo, N, Npad;
  int act;
  int ntn;
  int TH, TW, NI, tpi_x, tpi;  // tile geometry: tiles per image along x, tiles per image
  int HH, HW, HROWS, nrows;    // halo geometry (per image) and total halo rows
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
  } else {
    const int py = cls >> 1, px = cls & 1;
    const int nkx = px ? 3 : 2;
    const int a = t / nkx, b = t - a * nkx;
    const int ky = py ? 2 * a : 2 * a + 1;
    const int kx = px ? 2 * b : 2 * b + 1;
    dy = (py + 1 - ky) / 2;
    dx = (px + 1 - kx) / 2;
    widx = ky * 5 + kx;
  }
}

template <typename T, int MODE, int KCH, int WAVES_M, int WAVES_N, int MT, int NT>
__global__ __launch_bounds__(256) void conv_halo_kernel(const HaloParams p) {
  constexpr int ESZ = DT<T>::ESZ;
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
  static_assert(BM == 128, "halo tiles are 128 rows");
  constexpr int ROWB = KCH * 32 + 16;
  constexpr int KC = KCH * 32 / ESZ;
  constexpr int LPR = KC / 4;        // threads per halo row (fp32 source, float4 each)
  constexpr int RPP = 256 / LPR;     // halo rows per pass
  constexpr int BCH = BN * KCH * 2;  // 16-B chunks in the B tile
  constexpr int PB = (BCH + 255) / 256;
  constexpr int B_BYTES = BN * ROWB;
  constexpr int SS = (MODE == MODE_DOWN) ? 2 : 1;
  constexpr int LO = (MODE == MODE_S1T) ? -2 : -1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sB0 = smem;                                   // [2][BN][ROWB]
  int* s_out = reinterpret_cast<int*>(smem + 2 * B_BYTES);      // [128]
  int* s_hoff = s_out + BM;                           // [nrows] source pixel index or -1
  char* sH = smem + 2 * B_BYTES + ((BM + p.nrows) * 4 + 15) / 16 * 16;   // [nrows][ROWB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int cls = (MODE == MODE_UP) ? (3 - (int)blockIdx.y) : 0;
  const int py = cls >> 1, px = cls & 1;
  const int lb = lg_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = lb % p.ntn, tile_m = lb / p.ntn;
  const int n0 = tile_n * BN;

  // tile origin
  int img0, y0, x0;
  if (p.NI == 1) {
    img0 = tile_m / p.tpi;
    const int tt = tile_m - img0 * p.tpi;
    y0 = (tt / p.tpi_x) * p.TH; x0 = (tt % p.tpi_x) * p.TW;
  } else {
    img0 = tile_m * p.NI; y0 = 0; x0 = 0;
  }
  const int THW = p.TH * p.TW;

  // ---- tables: output pixel per M row, source pixel per halo row ------------------------------
  if (tid < BM) {
    const int i = tid / THW, rem = tid - i * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    const int n = img0 + i, y = y0 + ly, x = x0 + lx;
    int o = -1;
    if (n < p.B) {
      const int oy = (MODE == MODE_UP) ? 2 * y + py : y;
      const int ox = (MODE == MODE_UP) ? 2 * x + px : x;
      o = (n * p.Ho + oy) * p.Wo + ox;
    }
    s_out[tid] = o;
  }
  for (int hr = tid; hr < p.nrows; hr += 256) {
    const int i = hr / p.HROWS, rem = hr - i * p.HROWS;
    const int hy = rem / p.HW, hx = rem - hy * p.HW;
    const int n = img0 + i, sy = SS * y0 + LO + hy, sx = SS * x0 + LO + hx;
    int o = -1;
    if (n < p.B && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws) o = (n * p.Hs + sy) * p.Ws + sx;
    s_hoff[hr] = o;
  }
  // per-lane halo base row of the MT fragments this wave reads
  int hb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = (wm * MT + i) * 32 + r;
    const int ii = m / THW, rem = m - ii * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    hb[i] = ii * p.HROWS + (SS * ly - LO) * p.HW + SS * lx - LO;
  }

  int ntaps;
  if (MODE == MODE_UP) ntaps = (py ? 3 : 2) * (px ? 3 : 2);
  else ntaps = 25;
  const int nchunk = p.Cs / KC;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int arow = tid / LPR, alc = tid % LPR;
  u32x4 rb[PB];

  auto load_b = [&](int t, int c0) {
    int dy, dx, widx;
    tap_info(MODE, cls, t, dy, dx, widx);
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int c = q * 256 + tid;
      if (BCH % 256 == 0 || c < BCH) {
        const int row = c / (KCH * 2), ch = c % (KCH * 2);
        const char* g = p.wp + ((long long)(widx * p.Npad + n0 + row) * p.Cs + c0) * ESZ + ch * 16;
        rb[q] = *reinterpret_cast<const u32x4*>(g);
      }
    }
  };
  auto store_b = [&](int buf) {
    char* sB = sB0 + buf * B_BYTES;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int c = q * 256 + tid;
      if (BCH % 256 == 0 || c < BCH) {
        const int row = c / (KCH * 2), ch = c % (KCH * 2);
        *reinterpret_cast<u32x4*>(sB + row * ROWB + ch * 16) = rb[q];
      }
    }
  };

  __syncthreads();  // tables visible

  for (int cc = 0; cc < nchunk; ++cc) {
    const int c0 = cc * KC;
    // ---- stage the halo of this channel chunk (4 rows in flight per thread) ------------------
    for (int hr0 = 0; hr0 < p.nrows; hr0 += 4 * RPP) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int hr = hr0 + u * RPP + arow;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hr < p.nrows) {
          const int o = s_hoff[hr];
          if (o >= 0) v[u] = *reinterpret_cast<const f32x4*>(p.src + (long long)o * p.Cs + c0 + alc * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int hr = hr0 + u * RPP + arow;
        if (hr < p.nrows) {
          if constexpr (ESZ == 4) {
            *reinterpret_cast<f32x4*>(sH + hr * ROWB + alc * 16) = v[u];
          } else {
            bf16x4 w;
            w[0] = (__bf16)v[u][0]; w[1] = (__bf16)v[u][1]; w[2] = (__bf16)v[u][2]; w[3] = (__bf16)v[u][3];
            *reinterpret_cast<bf16x4*>(sH + hr * ROWB + alc * 8) = w;
          }
        }
      }
    }
    load_b(0, c0);
    store_b(0);
    __syncthreads();

    for (int t = 0; t < ntaps; ++t) {
      const int buf = t & 1;
      if (t + 1 < ntaps) load_b(t + 1, c0);
      int dy, dx, widx;
      tap_info(MODE, cls, t, dy, dx, widx);
      const int toff = dy * p.HW + dx;
      const char* sB = sB0 + buf * B_BYTES + (wn * NT * 32 + r) * ROWB + h * 16;
#pragma unroll
      for (int q = 0; q < KCH; ++q) {
        if constexpr (ESZ == 4) {
          f32x4 a[MT], b[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(sH + (hb[i] + toff) * ROWB + h * 16 + q * 32);
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
          for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sH + (hb[i] + toff) * ROWB + h * 16 + q * 32);
#pragma unroll
          for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sB + j * 32 * ROWB + q * 32);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
      if (t + 1 < ntaps) store_b(buf ^ 1);
      __syncthreads();  // B[buf^1] complete; after the last tap: every wave is done with the halo
    }
  }

  // ---- epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----------
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
          p.out[(long long)o * p.N + col] = v;
        }
      }
    }
  }
}

constexpr int LDS_BUDGET = 78 * 1024;  // two blocks per CU (160 KiB)

template <typename T, int MODE, int KCH, int WAVES_M, int WAVES_N, int MT, int NT>
int launch(HaloParams p, hipStream_t st) {
  constexpr int BN = WAVES_N * NT * 32, ROWB = KCH * 32 + 16;
  const size_t lds = 2 * (size_t)BN * ROWB + ((128 + p.nrows) * 4 + 15) / 16 * 16 + (size_t)p.nrows * ROWB;
  if (lds > LDS_BUDGET) return LG_ERR_UNSUPPORTED;
  p.ntn = p.Npad / BN;
  const int ntm = p.NI == 1 ? p.B * p.tpi : lg_cdiv(p.B, p.NI);
  dim3 grid(ntm * p.ntn, MODE == MODE_UP ? 4 : 1);
  auto kern = conv_halo_kernel<T, MODE, KCH, WAVES_M, WAVES_N, MT, NT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  return LG_OK;
}

template <typename T, int MODE, int KCH>
int dispatch_bn(const HaloParams& p, hipStream_t st) {
  if (p.Npad % 128 == 0) return launch<T, MODE, KCH, 2, 2, 2, 2>(p, st);
  if (p.Npad % 64 == 0) return launch<T, MODE, KCH, 2, 2, 2, 1>(p, st);
  return launch<T, MODE, KCH, 4, 1, 1, 1>(p, st);
}

template <int MODE>
int dispatch(const HaloParams& p, int dtype, hipStream_t st) {
  int rc = LG_ERR_UNSUPPORTED;
  if (dtype == LG_DT_F32) {
    if (p.Cs % 32 == 0) rc = dispatch_bn<float, MODE, 4>(p, st);
    if (rc == LG_ERR_UNSUPPORTED && p.Cs % 16 == 0) rc = dispatch_bn<float, MODE, 2>(p, st);
  } else {
    if (p.Cs % 64 == 0) rc = dispatch_bn<__bf16, MODE, 4>(p, st);
    if (rc == LG_ERR_UNSUPPORTED && p.Cs % 32 == 0) rc = dispatch_bn<__bf16, MODE, 2>(p, st);
  }
  return rc;
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int lg_npad(int n);

// Returns LG_OK if the halo kernel was launched, LG_ERR_UNSUPPORTED if the caller must use the gather kernel.
extern "C" int lg_conv_halo_try(int mode, int dtype, const float* src, const void* wpack, const float* bias, float* out,
                                int B, int Hm, int Wm, int Cs, int N, int act, void* stream) {
  if (mode != MODE_DOWN && mode != MODE_UP && mode != MODE_S1T) return LG_ERR_UNSUPPORTED;
  HaloParams p{};
  p.TW = Wm < 16 ? Wm : 16;
  if (!pow2(p.TW) || Wm % p.TW) return LG_ERR_UNSUPPORTED;
  p.TH = Hm < 128 / p.TW ? Hm : 128 / p.TW;
  if (!pow2(p.TH) || Hm % p.TH) return LG_ERR_UNSUPPORTED;
  p.NI = 128 / (p.TH * p.TW);
  if (p.NI > 1 && (p.TH != Hm || p.TW != Wm)) return LG_ERR_UNSUPPORTED;
  p.tpi_x = Wm / p.TW; p.tpi = p.tpi_x * (Hm / p.TH);
  const int ss = mode == MODE_DOWN ? 2 : 1, ext = mode == MODE_UP ? 2 : (mode == MODE_DOWN ? 3 : 4);
  p.HH = ss * p.TH + ext; p.HW = ss * p.TW + ext; p.HROWS = p.HH * p.HW; p.nrows = p.NI * p.HROWS;
  p.src = src; p.wp = (const char*)wpack; p.bias = bias; p.out = out;
  p.B = B; p.Cs = Cs; p.Hm = Hm; p.Wm = Wm; p.N = N; p.Npad = lg_npad(N); p.act = act;
  p.Hs = ss * Hm; p.Ws = ss * Wm;
  p.Ho = mode == MODE_UP ? 2 * Hm : Hm; p.Wo = mode == MODE_UP ? 2 * Wm : Wm;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (mode == MODE_DOWN) rc = dispatch<MODE_DOWN>(p, dtype, st);
  else if (mode == MODE_UP) rc = dispatch<MODE_UP>(p, dtype, st);
  else rc = dispatch<MODE_S1T>(p, dtype, st);
  if (rc != LG_OK) return rc;
  LG_CHECK_LAUNCH("lg_conv_halo");
  return LG_OK;
}
