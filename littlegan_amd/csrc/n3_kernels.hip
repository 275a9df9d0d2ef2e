// The 3-channel ("image side") layers of LittleGAN: 2 % of the step's FLOPs but, run through 32-wide MFMA tiles,
// a quarter of its time.  N = 3 output channels waste 29/32 of every MFMA, so these kernels use the VALU for the
// two forward-shaped ops (weights are wave-uniform -> they stream through SGPRs, activations come from an LDS halo
// tile) and the exact-f32 MFMA for the weight gradients (all 25 taps per pixel tile, operands gathered from LDS).
//   n3_s1t_fwd : y[B,H,W,3] = tanh(convT_s1(x[B,H,W,C]) + b)            /root/reference/model.py:86-87,104
//   n3_up      : dimg[B,2H,2W,3] = conv2d_backprop_input(dz[B,H,W,C])   (gradient of Encoder.conv1, model.py:15)
//   n3_wgrad   : dW[5][5][3][C] (+)= sum big3[s*o + k - pad][c3] * small[o][c]   (conv1: s=2,pad=1; final: s=1,pad=2)
#include <stdlib.h>
#include "lg_common.h"

namespace {

// --------------------------------------------------------------------------------------------------------------
// forward-shaped, 16x16 pixel tile per 256-thread block, one thread per (tile) pixel
// --------------------------------------------------------------------------------------------------------------
constexpr int TS = 16;

template <int C>
__global__ __launch_bounds__(256) void n3_s1t_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int H, int W) {
  constexpr int HS = TS + 4, ROWF = C + 4;  // halo side, floats per halo pixel (16-B pad: conflict-free b128 reads)
  extern __shared__ __attribute__((aligned(16))) float halo[];  // [HS*HS][ROWF]
  const int tpx = W / TS, tpi = tpx * (H / TS);
  const int n = blockIdx.x / tpi, tt = blockIdx.x % tpi;
  const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;
  for (int i = threadIdx.x; i < HS * HS * (C / 4); i += 256) {
    const int hp = i / (C / 4), c4 = i % (C / 4);
    const int sy = y0 - 2 + hp / HS, sx = x0 - 2 + hp % HS;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
      v = *reinterpret_cast<const f32x4*>(x + ((long long)(n * H + sy) * W + sx) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(halo + hp * ROWF + c4 * 4) = v;
  }
  __syncthreads();
  const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
  // two channels per instruction (v_pk_fma_f32, exact fp32 products as before): even channels add up in lane 0 of a pair, odd ones in
  // lane 1 — the kernel was bound by its 2400 scalar FMAs per pixel (9.6 k issue cycles per wave against 0.8 k of LDS reads)
  f32x2 acc[3] = {f32x2{bias[0], 0.f}, f32x2{bias[1], 0.f}, f32x2{bias[2], 0.f}};
#pragma unroll 1
  for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      // y[o] = sum_k x[o + 2 - k] W[k]: halo pixel (ly + 2 + (2-ky), lx + 2 + (2-kx)) in halo coords offset by -2
      const float* hp = halo + ((ly + 4 - ky) * HS + (lx + 4 - kx)) * ROWF;
      const float* wt = w + (ky * 5 + kx) * 3 * C;  // [co][c], wave-uniform -> scalar loads
#pragma unroll
      for (int c4 = 0; c4 < C / 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(hp + c4 * 4);
#pragma unroll
        for (int co = 0; co < 3; ++co) {
          const float* wc = wt + co * C + c4 * 4;
          acc[co] = __builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{wc[0], wc[1]}, acc[co]);
          acc[co] = __builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{wc[2], wc[3]}, acc[co]);
        }
      }
    }
  }
  float* o = y + ((long long)(n * H + y0 + ly) * W + x0 + lx) * 3;
  o[0] = tanhf(acc[0][0] + acc[0][1]); o[1] = tanhf(acc[1][0] + acc[1][1]); o[2] = tanhf(acc[2][0] + acc[2][1]);
}

// one thread per SOURCE pixel q of a 16x16 tile; it owns the 2x2 output quad (4 parity classes x 3 channels)
template <int CK>  // channels per LDS chunk
__global__ __launch_bounds__(256) void n3_up_kernel(const float* __restrict__ src, const float* __restrict__ w,
                                                    float* __restrict__ out, int B, int H, int W, int C) {
  constexpr int HS = TS + 2, ROWF = CK + 4;
  extern __shared__ __attribute__((aligned(16))) float halo[];  // [HS*HS][ROWF]
  const int tpx = W / TS, tpi = tpx * (H / TS);
  const int n = blockIdx.x / tpi, tt = blockIdx.x % tpi;
  const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;
  const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
  f32x2 acc[4][3];   // (even channels, odd channels): v_pk_fma_f32, see n3_s1t_fwd_kernel
#pragma unroll
  for (int k = 0; k < 4; ++k) { acc[k][0] = f32x2{0.f, 0.f}; acc[k][1] = f32x2{0.f, 0.f}; acc[k][2] = f32x2{0.f, 0.f}; }
  for (int c0 = 0; c0 < C; c0 += CK) {
    __syncthreads();
    for (int i = threadIdx.x; i < HS * HS * (CK / 4); i += 256) {
      const int hp = i / (CK / 4), c4 = i % (CK / 4);
      const int sy = y0 - 1 + hp / HS, sx = x0 - 1 + hp % HS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(src + ((long long)(n * H + sy) * W + sx) * C + c0 + c4 * 4);
      *reinterpret_cast<f32x4*>(halo + hp * ROWF + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        // out[2q+p] += src[q + d] W[k],  p = 1 - (k & 1),  d = (p + 1 - k) / 2   (conv2d_backprop_input, SAME, s=2, k=5)
        const int py = 1 - (ky & 1), px = 1 - (kx & 1);
        const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
        const int cls = py * 2 + px;
        const float* hp = halo + ((ly + 1 + dy) * HS + (lx + 1 + dx)) * ROWF;
        const float* wt = w + ((ky * 5 + kx) * 3) * C + c0;  // [cb=3][cs=C] rows
#pragma unroll
        for (int c4 = 0; c4 < CK / 4; ++c4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(hp + c4 * 4);
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float* wc = wt + co * C + c4 * 4;
            f32x2 a = acc[cls][co];
            a = __builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{wc[0], wc[1]}, a);
            a = __builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{wc[2], wc[3]}, a);
            acc[cls][co] = a;
          }
        }
      }
    }
  }
  const int y = y0 + ly, xq = x0 + lx;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    float* o = out + ((long long)(n * 2 * H + 2 * y + py) * 2 * W + 2 * xq) * 3;
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int co = 0; co < 3; ++co) o[px * 3 + co] = acc[py * 2 + px][co][0] + acc[py * 2 + px][co][1];
  }
}

// --------------------------------------------------------------------------------------------------------------
// weight gradient of a 3-channel layer, all 25 taps at once.  GEMM: M = 75 (tap, c3) -> 3 MFMA row tiles,
// N = Cs (1 or 2 column tiles), K = pixels of the small grid.  A[i][k] is GATHERED from an LDS halo of the
// 3-channel tensor (per-lane row offset + per-k pixel base), B[k][j] is the small tensor's pixel row.
// Block = 4 waves, pixel tile 8x16: wave w reduces pixels [32w, 32w+32); the waves merge into one fp32 slab per block.
// --------------------------------------------------------------------------------------------------------------
// S16: `small` is read from its bf16 mirror (bf16 path; the values are widened exactly, the MFMA stays f32).
template <int NT, bool S16>
__global__ __launch_bounds__(256) void n3_wgrad_kernel(const float* __restrict__ big3, const float* __restrict__ small,
                                                       const __bf16* __restrict__ small16, float* __restrict__ slab,
                                                       int B, int H, int W, int s, int pad) {
  constexpr int Cs = NT * 32, TH = 8, TW = 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HH = s * TH + 4, HW = s * TW + 4;  // halo of the 3-channel tensor (taps span 5 pixels)
  float* sB = smem;                            // [128][Cs]
  float* sA = smem + TH * TW * Cs;             // [HH*HW*3] (+ 1 zero)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int tpx = W / TW, tpi = tpx * (H / TH), ntiles = B * tpi;
  const int Hb = s * H, Wb = s * W;
  int aoff[3];
  bool aval[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = i * 32 + r;  // (tap, c3)
    aval[i] = idx < 75;
    const int t = idx / 3, c3 = idx - t * 3;
    aoff[i] = aval[i] ? ((t / 5) * HW + (t % 5)) * 3 + c3 : 0;
  }
  f32x16 acc[3][NT];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / tpi, tt = tile - n * tpi;
    const int y0 = (tt / tpx) * TH, x0 = (tt % tpx) * TW;
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW * (Cs / 4); i += 256) {
      const int pix = i / (Cs / 4), c4 = i % (Cs / 4);
      const int yy = y0 + pix / TW, xx = x0 + pix % TW;
      const long long go = ((long long)(n * H + yy) * W + xx) * Cs + c4 * 4;
      f32x4 v;
      if constexpr (S16) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(small16 + go);
        v = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
      } else {
        v = *reinterpret_cast<const f32x4*>(small + go);
      }
      *reinterpret_cast<f32x4*>(sB + pix * Cs + c4 * 4) = v;
    }
    for (int i = threadIdx.x; i < HH * HW * 3; i += 256) {
      const int hp = i / 3, c3 = i - hp * 3;
      const int sy = s * y0 - pad + hp / HW, sx = s * x0 - pad + hp % HW;
      float v = 0.f;
      if ((unsigned)sy < (unsigned)Hb && (unsigned)sx < (unsigned)Wb) v = big3[((long long)(n * Hb + sy) * Wb + sx) * 3 + c3];
      sA[i] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int m = wid * 32 + 2 * kk + h;  // pixel of this lane's k
      const int ly = m / TW, lx = m % TW;
      const int pbase = ((s * ly) * HW + s * lx) * 3;
      float a[3], b[NT];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = aval[i] ? sA[pbase + aoff[i]] : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = sB[m * Cs + j * 32 + r];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // merge the four waves' partial sums in wave order through LDS (sB is dead), then one coalesced slab per block
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < 75) {
              float* q = sB + row * Cs + j * 32 + r;
              *q = (w == 0 ? 0.f : *q) + acc[i][j][e];
            }
          }
    }
  }
  __syncthreads();
  float* o = slab + (long long)blockIdx.x * 75 * Cs;
  for (int i = threadIdx.x; i < 75 * Cs / 4; i += 256)
    *reinterpret_cast<f32x4*>(o + i * 4) = *reinterpret_cast<const f32x4*>(sB + i * 4);
}

// bf16 path of the same contraction: small = bf16 mirror, staged verbatim as [pixel][channel] bf16 and read as MFMA B
// operands with ds_read_b64_tr_b16 (k = pixel is the row index) — v_mfma_f32_32x32x16_bf16, 16 pixels per instruction, fp32
// accumulate.  A k step = one tile row of 16 pixels.
// TH (8 | 16) = rows of a tile.  TH = 16 doubles the tile of the 32-channel stride-1 layer (the final layer's weight gradient).
// The 3-channel operand lives in LDS as the bf16 RGBx HALO of the tile (8 bytes per pixel: three channels + a zero, rounded once,
// RNE) and is read with the SAME transposing load (round 4): M is laid out as (ky, kx, c4) = 25 taps x 4, so the 16 columns of a
// 16-lane group are four taps, a lane supplies the address of ITS tap's pixel for one of four consecutive k (halo pixel
// (S ly + ky, S x' + kx): the rows of the "matrix" overlap, which the instruction does not mind), and the hardware transpose hands
// every lane the 8 pixels of its (tap, c4) row.  100 rows -> four 32-row tiles (c4 = 3 and rows >= 100 are zero / dropped at the merge).
// Before: 15 kx-shifted bf16 planes filled by scattering every halo value to the <= 5 planes it appears in — 15 - 27 two-byte LDS
// stores with their selects per thread and tile, the largest item of the kernel (ablation, conv1 weight gradient at 2B: 164 -> 76 us
// without it); now one 8-byte store per halo pixel, at the price of 4 instead of 3 row tiles of MFMAs in a kernel whose matrix pipe is
// 6 - 10 % busy.
#ifndef LG_N3W_DBG
#define LG_N3W_DBG 0   // compile-time ablation bits (timing only, results wrong): 1 no halo stores, 2 no 3-channel loads, 4 no wide-operand loads, 8 no MFMA, 16 no wide-operand LDS stores
#endif
// 64 channels (NT = 2): the waves split the CHANNELS as well as the pixels — wave w owns channel half w & 1 and tile rows 4 (w >> 1) ..
// + 3 — so a wave carries 64 accumulator registers instead of 128 and the kernel fits three blocks per CU (round 4: at 256 registers
// and two blocks every ablation that happened to free registers ran the conv1 weight gradient in ~100 instead of 164 us whatever it
// removed; held to 168 registers by launch bounds alone it spilled 73).
#ifndef LG_N3W_LB
#define LG_N3W_LB 3   // blocks per CU the 64-channel forms are compiled for (A/B builds)
#endif
template <int NT, int TH = 8, int S = 1>
__global__ __launch_bounds__(256, NT == 2 ? LG_N3W_LB : 1) void n3_wgrad16_kernel(const float* __restrict__ big3, const __bf16* __restrict__ small16,
                                                         float* __restrict__ slab, int B, int H, int W, int s_unused, int pad) {
  static_assert(TH == 8 || (TH == 16 && NT == 1 && S == 1), "tile heights");
  static_assert(S == 1 || S == 2, "strides");
  constexpr int s = S;
  constexpr int Cs = NT * 32, TW = 16;
  constexpr int RSB = Cs * 2 + 16;                       // bytes per pixel row of sB (16-B pad)
  constexpr int SB_BYTES = (TH * TW * RSB > 75 * Cs * 4) ? TH * TW * RSB : 75 * Cs * 4;  // also the merge buffer
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int HH = s * TH + 4, HW = s * TW + 4;
  constexpr int H_BYTES = HH * HW * 8;                   // the RGBx halo
  constexpr int ZOFF = H_BYTES;                          // then 64 zero bytes... (a transposed read takes 8 bytes here and 8 more s*32 bytes on)
  char* sB = smem16;                                       // [TH*16][RSB] bf16
  char* sH = smem16 + SB_BYTES;                            // halo [HH][HW] x 8 B | zero region
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int tpx = W / TW, tpi = tpx * (H / TH), ntiles = B * tpi;
  const int Hb = s * H, Wb = s * W;
  (void)s_unused;
  // ds_read_b64_tr_b16 addressing (as wgrad_igemm.hip): lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3
  const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int colbase = 16 * (g & 1) + 4 * lp;
  // A side: row tile i, this lane's column group = tap j = 8 i + 4 (g & 1) + lp, its rows = pixels x' = 8 (g >> 1) + lq (+ 4)
  int aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 8 * i + 4 * (g & 1) + lp, ky = j / 5, kx = j - 5 * ky;
    aoff[i] = j < 25 ? (ky * HW + s * (8 * (g >> 1) + lq) + kx) * 8 : ZOFF;
  }
  for (int i = threadIdx.x; i < (64 + s * 32) / 4; i += 256) reinterpret_cast<unsigned*>(sH + ZOFF)[i] = 0u;   // (the first barrier of the tile loop publishes it)
  constexpr bool CSPLIT = NT == 2;
  constexpr int NTW = CSPLIT ? 1 : NT;            // 32-channel column tiles per wave
  constexpr int KSW = CSPLIT ? TH / 2 : TH / 4;   // k steps (tile rows of 16 pixels) per wave
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int chw = CSPLIT ? (wu & 1) : 0, kw = CSPLIT ? (wu >> 1) : wu;
  f32x16 acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // the NEXT tile's operands are requested (global -> registers) before this tile's MFMAs and written to LDS behind them
  // (round 3: TWO tiles ahead in two named register sets costs the resident blocks that hide the rest: conv1 weight gradient 83 -> 102 us,
  //  final 177 -> 290 us at B = 256.  Dropped.)
  constexpr int NPB = TH * TW * (Cs / 8) / 256;          // 16-B pieces of the wide operand per thread (2 | 4)
  constexpr int NPA = (HH * HW + 255) / 256;             // halo pixels per thread: 1 (TH 8, S 1) | 2 (TH 16) | 3 (S 2)
  u32x4 rb[NPB];
  float ra[NPA][3];
  auto tile_load = [&](int tile) {
    const int n = tile / tpi, tt = tile - n * tpi;
    const int y0 = (tt / tpx) * TH, x0 = (tt % tpx) * TW;
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int i = threadIdx.x + u * 256, pix = i / (Cs / 8), c8 = i % (Cs / 8);
      const int yy = y0 + pix / TW, xx = x0 + pix % TW;
      if constexpr (!(LG_N3W_DBG & 4)) rb[u] = *reinterpret_cast<const u32x4*>(small16 + ((long long)(n * H + yy) * W + xx) * Cs + c8 * 8);
      else rb[u] = u32x4{(unsigned)yy, (unsigned)xx, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int hp = threadIdx.x + u * 256;
      const int sy = s * y0 - pad + hp / HW, sx = s * x0 - pad + hp % HW;
      ra[u][0] = 0.f; ra[u][1] = 0.f; ra[u][2] = 0.f;
      if constexpr (!(LG_N3W_DBG & 2)) {
        if (hp < HH * HW && (unsigned)sy < (unsigned)Hb && (unsigned)sx < (unsigned)Wb) {
          const float* q = big3 + ((long long)(n * Hb + sy) * Wb + sx) * 3;
          ra[u][0] = q[0]; ra[u][1] = q[1]; ra[u][2] = q[2];
        }
      } else ra[u][0] = (float)(sy + sx);
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int i = threadIdx.x + u * 256, pix = i / (Cs / 8), c8 = i % (Cs / 8);
      if constexpr (!(LG_N3W_DBG & 16)) *reinterpret_cast<u32x4*>(sB + pix * RSB + c8 * 16) = rb[u];
      else if (rb[u][0] == 0x12345u) *reinterpret_cast<u32x4*>(sB + pix * RSB + c8 * 16) = rb[u];
    }
#pragma unroll
    for (int u = 0; u < ((LG_N3W_DBG & 1) ? 0 : NPA); ++u) {
      const int hp = threadIdx.x + u * 256;
      if (hp < HH * HW)
        *reinterpret_cast<bf16x4*>(sH + hp * 8) = bf16x4{(__bf16)ra[u][0], (__bf16)ra[u][1], (__bf16)ra[u][2], (__bf16)0.f};
    }
  };
  if ((int)blockIdx.x < ntiles) tile_load(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    tile_store();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) tile_load(tile + gridDim.x);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      const int ly = KSW * kw + ks;  // tile row = the 16 pixels of this k step
      bf16x8 a[4], b[NTW];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (rows >= 100: both halves come out of the zero region; s * 32 bytes on is still inside it)
        const char* pa = sH + aoff[i] + (aoff[i] == ZOFF ? 0 : s * ly * HW * 8);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + s * 32));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
      const int row0 = ly * TW + 8 * (g >> 1) + lq;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const char* pb = sB + row0 * RSB + ((chw * NTW + j) * 32 + colbase) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb + 4 * RSB));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        b[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if constexpr (!(LG_N3W_DBG & 8)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          else acc[i][j][0] += (float)a[i][0] * (float)b[j][0];
        }
    }
    if constexpr ((LG_N3W_DBG & 1) != 0) { if (ra[0][0] == 12345.678f) sH[threadIdx.x] = 1; }   // keeps the 3-channel loads alive
  }
  float* mrg = reinterpret_cast<float*>(sB);  // merge the waves' pixel shares in order (sB is dead), one slab per block
  for (int w = 0; w < (CSPLIT ? 2 : 4); ++w) {
    __syncthreads();
    if (kw == w) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;   // (tap, c4): tap = m / 4, c4 = e & 3
            if (m < 100 && (e & 3) < 3) {
              float* q = mrg + ((m >> 2) * 3 + (e & 3)) * Cs + (chw * NTW + j) * 32 + r;
              *q = (w == 0 ? 0.f : *q) + acc[i][j][e];
            }
          }
    }
  }
  __syncthreads();
  float* o = slab + (long long)blockIdx.x * 75 * Cs;
  for (int i = threadIdx.x; i < 75 * Cs / 4; i += 256)
    *reinterpret_cast<f32x4*>(o + i * 4) = *reinterpret_cast<const f32x4*>(mrg + i * 4);
}

// dw[i] (+)= sum_k slab[k][i] : 16 outputs x 16 slab groups per block, merged in group order (deterministic)
__global__ __launch_bounds__(256) void n3_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                             int nslab, int n, int accumulate) {
  __shared__ float sr[16][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (o < n) {
    int k = g;
    for (; k + 48 < nslab; k += 64) {
      const float a = slab[(long long)k * n + o], b = slab[(long long)(k + 16) * n + o];
      const float c = slab[(long long)(k + 32) * n + o], d = slab[(long long)(k + 48) * n + o];
      s += (a + b) + (c + d);
    }
    for (; k < nslab; k += 16) s += slab[(long long)k * n + o];
  }
  sr[g][cl] = s;
  __syncthreads();
  if (g == 0 && o < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sr[q][cl];
    dw[o] = (accumulate ? dw[o] : 0.f) + t;
  }
}

// persistent blocks: Cs 64: 3 per CU since the channel-split waves of round 4 (168 registers; 512 / 768 / 1024 blocks: 173 / 138 / 180 us
// at 2B; before, at 256 registers, 2 per CU was best); Cs 32: 3 per CU (512 / 768 / 1024 / 1536: 121 / 112 / 127 / 120 us with its bias sums)
inline int wgrad_blocks(int ntiles, int Cs, bool f32_path = false) {
  static int f32 = -1, f64 = -1;
  if (f32 < 0) { const char* e = getenv("LG_N3W_CAP32"); f32 = e ? atoi(e) : 0; const char* g = getenv("LG_N3W_CAP64"); f64 = g ? atoi(g) : 0; }
  const int cap = Cs > 32 ? (f64 > 0 ? f64 : (f32_path ? 512 : 768)) : (f32 > 0 ? f32 : 768);   // (the fp32-source kernel at 64 channels: 41 KB of LDS, 2 per CU measured best)  // measured with the prefetch (256..2048): Cs 64: 128 / 82 / 109 / 99 / 113 / 130 us, Cs 32: 320 / 212 / 182 / 214 / 189 / 200 us
  return ntiles < cap ? ntiles : cap;
}

}  // namespace

// ---- entry points used by conv_igemm.hip / wgrad_igemm.hip / capi.hip (LG_ERR_UNSUPPORTED -> generic kernels) ----
extern "C" int lg_n3_s1t_fwd_try(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                 void* stream) {
  if (H % TS || W % TS || !bias || !w) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int grid = B * (H / TS) * (W / TS);
  if (C == 32) {
    const size_t lds = (size_t)(TS + 4) * (TS + 4) * (32 + 4) * 4;
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(n3_s1t_fwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL(n3_s1t_fwd_kernel<32>, dim3(grid), dim3(256), lds, st, x, w, bias, y, B, H, W);
  } else if (C == 64) {
    const size_t lds = (size_t)(TS + 4) * (TS + 4) * (64 + 4) * 4;
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(n3_s1t_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL(n3_s1t_fwd_kernel<64>, dim3(grid), dim3(256), lds, st, x, w, bias, y, B, H, W);
  } else {
    return LG_ERR_UNSUPPORTED;
  }
  LG_CHECK_LAUNCH("lg_n3_s1t_fwd");
  return LG_OK;
}

extern "C" int lg_n3_up_try(const float* src, const float* w, float* out, int B, int H, int W, int C, void* stream) {
  if (H % TS || W % TS || C % 32 || !w) return LG_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(TS + 2) * (TS + 2) * (32 + 4) * 4;
  hipLaunchKernelGGL(n3_up_kernel<32>, dim3(B * (H / TS) * (W / TS)), dim3(256), lds, (hipStream_t)stream, src, w, out, B,
                     H, W, C);
  LG_CHECK_LAUNCH("lg_n3_up");
  return LG_OK;
}

extern "C" size_t lg_n3_wgrad_workspace_bytes(int B, int H, int W, int Cs) {
  const int ntiles = B * (H / 8) * (W / 16);
  return (size_t)wgrad_blocks(ntiles > 0 ? ntiles : 1, Cs) * 75 * Cs * sizeof(float);
}

extern "C" int lg_n3_wgrad_try(const float* big3, const float* small, const void* small16, float* dw, void* workspace,
                               size_t ws_bytes, int B, int H, int W, int Cs, int s, int pad, int accumulate, void* stream) {
  if (H % 8 || W % 16 || (Cs != 32 && Cs != 64) || (s != 1 && s != 2)) return LG_ERR_UNSUPPORTED;
  if (ws_bytes < lg_n3_wgrad_workspace_bytes(B, H, W, Cs)) return LG_ERR_UNSUPPORTED;
  if (!big3 || (!small && !small16)) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  static int th8 = -1;
  if (th8 < 0) th8 = lg_env_flag("LG_N3W_TH8") ? 1 : 0;   // A/B switch
  static int n3w_f32 = -1;
  if (n3w_f32 < 0) n3w_f32 = lg_env_flag("LG_N3W_F32") ? 1 : 0;   // (cached per call site: the table lookup takes a mutex)
  const bool th16 = Cs == 32 && s == 1 && H % 16 == 0 && small16 && !n3w_f32 && !th8;   // 16-row tiles (bf16 path, final layer)
  const bool p16 = small16 && !n3w_f32;
  const int ntiles = B * (H / (th16 ? 16 : 8)) * (W / 16), nblk = wgrad_blocks(ntiles, Cs, !p16);   // (the workspace is sized for the larger grid)
  const size_t lds = (size_t)(128 * Cs + (s * 8 + 4) * (s * 16 + 4) * 3 + 4) * 4;
  const __bf16* s16 = (const __bf16*)small16;
  static bool a = false;
  if (!a) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(n3_wgrad_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(n3_wgrad_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    a = true;
  }
  float* slab = (float*)workspace;
  if (s16 && !n3w_f32) {  // bf16 path: bf16 MFMA straight from the mirror
    // LDS: the wide operand's tile (also the merge buffer) + the bf16 RGBx halo of the 3-channel operand and its zero region
    auto ldsz = [&](int cs, int th = 8) {
      const size_t sb = (size_t)th * 16 * (cs * 2 + 16), mg = (size_t)75 * cs * 4;
      return (sb > mg ? sb : mg) + (size_t)(s * th + 4) * (s * 16 + 4) * 8 + 64 + (size_t)s * 32 + 64;   // + the RGBx halo and its zero region
    };
    if (th16) hipLaunchKernelGGL((n3_wgrad16_kernel<1, 16, 1>), dim3(nblk), dim3(256), ldsz(32, 16), st, big3, s16, slab, B, H, W, s, pad);
    else if (Cs == 32 && s == 1) hipLaunchKernelGGL((n3_wgrad16_kernel<1, 8, 1>), dim3(nblk), dim3(256), ldsz(32), st, big3, s16, slab, B, H, W, s, pad);
    else if (Cs == 32) hipLaunchKernelGGL((n3_wgrad16_kernel<1, 8, 2>), dim3(nblk), dim3(256), ldsz(32), st, big3, s16, slab, B, H, W, s, pad);
    else if (s == 1) hipLaunchKernelGGL((n3_wgrad16_kernel<2, 8, 1>), dim3(nblk), dim3(256), ldsz(64), st, big3, s16, slab, B, H, W, s, pad);
    else hipLaunchKernelGGL((n3_wgrad16_kernel<2, 8, 2>), dim3(nblk), dim3(256), ldsz(64), st, big3, s16, slab, B, H, W, s, pad);
  } else if (Cs == 32) {
    if (s16) hipLaunchKernelGGL((n3_wgrad_kernel<1, true>), dim3(nblk), dim3(256), lds, st, big3, small, s16, slab, B, H, W, s, pad);
    else hipLaunchKernelGGL((n3_wgrad_kernel<1, false>), dim3(nblk), dim3(256), lds, st, big3, small, s16, slab, B, H, W, s, pad);
  } else {
    if (s16) hipLaunchKernelGGL((n3_wgrad_kernel<2, true>), dim3(nblk), dim3(256), lds, st, big3, small, s16, slab, B, H, W, s, pad);
    else hipLaunchKernelGGL((n3_wgrad_kernel<2, false>), dim3(nblk), dim3(256), lds, st, big3, small, s16, slab, B, H, W, s, pad);
  }
  LG_CHECK_LAUNCH("lg_n3_wgrad");
  lg_note_kernel((s16 && !n3w_f32) ? (th16 ? "n3_wgrad16_kernel<1,16>" : Cs == 32 ? "n3_wgrad16_kernel<1>" : "n3_wgrad16_kernel<2>") : "n3_wgrad_kernel<f32>");
  const int n = 75 * Cs;
  hipLaunchKernelGGL(n3_slab_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, st, (const float*)workspace, dw, nblk, n,
                     accumulate);
  LG_CHECK_LAUNCH("lg_n3_wgrad(reduce)");
  return LG_OK;
}
