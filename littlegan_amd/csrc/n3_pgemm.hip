// bf16-path kernels of the two forward-shaped 3-channel layers, as a "tap-product GEMM + shift-sum":
//
//   out[o][c3] = sum_{tap,c} src[o + d(tap)][c] * W[tap][c3][c]
//              = sum_tap P[o + d(tap)][tap][c3],      P[q][tap][c3] = sum_c src[q][c] * W[tap][c3][c]
//
// P is a dense GEMM  [pixels x C] x [C x 75]  — every MFMA column is a real (tap, c3) output instead of 3 useful
// columns out of 32 — and the sum over taps is 75 LDS reads + adds per pixel.  A block takes one 16x16 tile, keeps
// the bf16 source halo as MFMA A fragments in registers (loaded straight from the bf16 mirror, 16 B per lane), and
// runs one pass per filter ROW ky: the 16-column B fragment is (kx, c3) = 15 real columns, the 16x16 fp32 products
// go to LDS, and every thread adds the 5 x 3 values its pixel receives from that row.
//   s1t_fwd_p16 : y[B,H,W,3] = tanh(convT_s1(x16[B,H,W,C]) + b)           /root/reference/model.py:86-87,104
//   up_p16      : dimg[B,2H,2W,3] = conv2d_backprop_input(dz16[B,H,W,C])  (gradient of Encoder.conv1, model.py:15)
// Weights come from the verbatim fp32 copy in the pack (pack.hip, cb == 3) and are rounded to bf16 (RNE) here, the
// same rounding the packed MFMA operands get; accumulation is fp32 throughout.
#include <stdlib.h>
#include <type_traits>
#include "lg_common.h"

#ifndef LG_P16_DBG
#define LG_P16_DBG 0   // compile-time ablation bits (timing only, results wrong).  patch_p16: 1 no image loads, 2 no output stores, 4 no MFMA, 8 no moments;
#endif                 // up_p16: 16 no shift-sum reads, 32 no product writes, 64 no source loads, 128 no barriers inside a tile
namespace {

constexpr int TS = 16;  // tile side (pixels of the M grid)
constexpr int PR = 17;  // LDS row pitch of P (floats): 16 columns + 1 -> the shift-sum reads are conflict-free

__device__ __forceinline__ bf16x8 cvt8(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  bf16x8 v;
  v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
  v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
  return v;
}
__device__ __forceinline__ bf16x8 zero8() {
  return __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
}

// B fragments of filter row ky: column n = kx*3 + c3 (n == 15 is padding), k = channel.  16x16x32 layout: lane l holds
// B[k = 8(l>>4) + j][col l&15].  w is [5][5][3][C] fp32.
template <int C>
__device__ __forceinline__ void load_bfrags(const float* __restrict__ w, int lane, bf16x8 (&bf)[5][C / 32]) {
  const int n = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int kh = 0; kh < C / 32; ++kh)
      bf[ky][kh] = n < 15 ? cvt8(w + ((long long)(ky * 5) * 3 + n) * C + kh * 32 + g * 8) : zero8();
}

template <int C>
__global__ __launch_bounds__(256) void s1t_fwd_p16_kernel(const __bf16* __restrict__ x16, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B,
                                                          int H, int W) {
  constexpr int HS = TS + 4, NQ = HS * HS, NMT = NQ / 16, MTW = (NMT + 3) / 4, KH = C / 32;
  static_assert(NQ % 16 == 0, "halo pixels must fill whole MFMA row tiles");
  __shared__ float sP[NQ * PR];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int tpx = W / TS, tpi = tpx * (H / TS);
  const int n = blockIdx.x / tpi, tt = blockIdx.x % tpi;
  const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;

  bf16x8 bf[5][KH], af[MTW][KH];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {  // A fragments: lane l holds A[row l&15][k = 8(l>>4) + j]
    const int mt = wid + 4 * i, q = mt * 16 + r;
    const int sy = y0 - 2 + q / HS, sx = x0 - 2 + q % HS;
    const bool ok = mt < NMT && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
      af[i][kh] = ok ? *reinterpret_cast<const bf16x8*>(x16 + ((long long)(n * H + sy) * W + sx) * C + kh * 32 + g * 8) : zero8();
  }
  load_bfrags<C>(w, lane, bf);

  const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
  float acc[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int mt = wid + 4 * i;
      if (mt < NMT) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kh], bf[ky][kh], c, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) sP[(mt * 16 + 4 * g + e) * PR + r] = c[e];  // C: col = lane&15, row = 4(lane>>4)+e
      }
    }
    __syncthreads();
    // y[o] = sum_k x[o + 2 - k] W[k]: halo pixel (ly + 4 - ky, lx + 4 - kx), halo origin = tile origin - 2
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      const float* pp = sP + ((ly + 4 - ky) * HS + (lx + 4 - kx)) * PR + kx * 3;
      acc[0] += pp[0]; acc[1] += pp[1]; acc[2] += pp[2];
    }
    __syncthreads();
  }
  float* o = y + ((long long)(n * H + y0 + ly) * W + x0 + lx) * 3;
  o[0] = tanhf(acc[0]); o[1] = tanhf(acc[1]); o[2] = tanhf(acc[2]);
}

// one thread per SOURCE pixel q of a 16x16 tile; it owns the 2x2 output quad (4 parity classes x 3 channels)
template <int C>
#ifndef LG_UP16_SINGLE
#define LG_UP16_SINGLE 0   // A/B builds: 1 = one A-fragment set (no tile-ahead prefetch), compiled for three blocks per CU (168 VGPRs).
                           // Round 4, same box, LG_PATCH_GRID=768: 58.0 against 64.4 us at B = 256, 119.9 against 117.1 at 2B — not adopted
#endif
__global__ __launch_bounds__(256, LG_UP16_SINGLE ? 3 : 1) void up_p16_kernel(const __bf16* __restrict__ src16, const float* __restrict__ w,
                                                     float* __restrict__ out, int B, int H, int W) {
  constexpr int HS = TS + 2, NQ = HS * HS, NMT = (NQ + 15) / 16, MTW = (NMT + 3) / 4, KH = C / 32;
  __shared__ float sP[NMT * 16 * PR];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int tpx = W / TS, tpi = tpx * (H / TS), ntiles = B * tpi, G = gridDim.x;

  // persistent: the 5 x KH weight fragments (8 converted weights each per lane) are set up once per block, and the A
  // fragments of the block's NEXT tile are requested before this tile's passes (two register sets, loop unrolled by two)
  bf16x8 bf[5][KH];
  load_bfrags<C>(w, lane, bf);
  auto load_af = [&](int t, bf16x8 (&af)[MTW][KH]) {
    const int n = t / tpi, tt = t - n * tpi;
    const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int mt = wid + 4 * i, q = mt * 16 + r;
      const int sy = y0 - 1 + q / HS, sx = x0 - 1 + q % HS;
      const bool ok = q < NQ && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
        af[i][kh] = (ok && !(LG_P16_DBG & 64)) ? *reinterpret_cast<const bf16x8*>(src16 + ((long long)(n * H + sy) * W + sx) * C + kh * 32 + g * 8) : zero8();
    }
  };
  const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
  auto compute = [&](int t, const bf16x8 (&af)[MTW][KH]) {
    const int n = t / tpi, tt = t - n * tpi;
    const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;
    float acc[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[k][0] = 0.f; acc[k][1] = 0.f; acc[k][2] = 0.f; }
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int mt = wid + 4 * i;
        if (mt < NMT) {
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < KH; ++kh) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kh], bf[ky][kh], c, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (!(LG_P16_DBG & 32)) sP[(mt * 16 + 4 * g + e) * PR + r] = c[e];
            else if (c[e] == 12345.f) sP[(mt * 16 + 4 * g + e) * PR + r] = c[e];
          }
        }
      }
      if constexpr (!(LG_P16_DBG & 128)) __syncthreads();
      // out[2q+p] += src[q + d] W[k],  p = 1 - (k & 1),  d = (p + 1 - k) / 2   (conv2d_backprop_input, SAME, s=2, k=5)
      const int py = 1 - (ky & 1), dy = (py + 1 - ky) / 2;
#pragma unroll
      for (int kx = 0; kx < ((LG_P16_DBG & 16) ? 1 : 5); ++kx) {
        const int px = 1 - (kx & 1), dx = (px + 1 - kx) / 2;
        const float* pp = sP + ((ly + 1 + dy) * HS + (lx + 1 + dx)) * PR + kx * 3;
        acc[py * 2 + px][0] += pp[0]; acc[py * 2 + px][1] += pp[1]; acc[py * 2 + px][2] += pp[2];
      }
      if constexpr (!(LG_P16_DBG & 128)) __syncthreads();
    }
    const int yq = y0 + ly, xq = x0 + lx;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float* o = out + ((long long)(n * 2 * H + 2 * yq + py) * 2 * W + 2 * xq) * 3;
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int co = 0; co < 3; ++co) o[px * 3 + co] = acc[py * 2 + px][co];
    }
  };
#if LG_UP16_SINGLE
  bf16x8 afS[MTW][KH];
  for (int t = blockIdx.x; t < ntiles; t += G) { load_af(t, afS); compute(t, afS); }
  return;
#endif
  bf16x8 afA[MTW][KH], afB[MTW][KH];
  int t = blockIdx.x;
  if (t < ntiles) load_af(t, afA);
  for (; t < ntiles; t += 2 * G) {
    if (t + G < ntiles) load_af(t + G, afB);
    compute(t, afA);
    if (t + G < ntiles) {
      if (t + 2 * G < ntiles) load_af(t + 2 * G, afA);
      compute(t + G, afB);
    }
  }
}


// --------------------------------------------------------------------------------------------------------------
// "patch" layers: the 3-channel tensor is the SOURCE (K = 5*5*3 = 75, N = 32 or 64 wide).
//   S = 2, pad 1 : Encoder.conv1 forward        z[B,H,W,N]   = conv2d_s2(img[B,2H,2W,3]) + b        model.py:15
//   S = 1, pad 2 : final layer's data gradient  dx[B,H,W,N]  = sum_{k,co} dpre[i+k-2][co] W[k][co][:]  model.py:86-87
// Both are  out[o][n] = b[n] + sum_k patch(o)[k] * w[k*N + n],  k = (ky, kx, c3).  The image tile lives in LDS as bf16 RGBx
// pixels (8 bytes: three channels + a zero) and K is laid out as (ky, kx, c4) with SIX pixels per filter row (kx = 5 and c4 = 3
// carry zero weights): 5 x 24 = 120 -> 128 = four k-steps of 32.  The 8 elements of a lane's A fragment are then TWO NEIGHBOURING
// PIXELS of one tile row — one ds_read_b128 (S = 2: always 16-B aligned) or one ds_read2_b64 (S = 1) per k-step, no conversion.
// (Round 4.  Before: fp32 pixels of 12 bytes, K = 6 groups of 16 consecutive floats, 96 = three k-steps — a fragment was 8 scalar
//  LDS reads + 5 conversions + 4 byte shuffles, 39 of the ~95 instructions of a 16-pixel row segment in a kernel whose four waves
//  per SIMD fill 96 % of the issue slots; 8 MFMAs instead of 6 per segment are the price.)
// These layers are bound by their OUTPUT stream (N floats per pixel), so
// the C tiles go through LDS and leave as 16-B-per-lane stores of whole contiguous rows; the conv1 form also emits
// the per-block InstanceNorm moments {count, mean, M2} (same record as conv_halo.hip) so no pass re-reads z.
// --------------------------------------------------------------------------------------------------------------
template <int S, int N, bool OUT16, bool STATS, bool FUSE = false>
__global__ __launch_bounds__(256, 4) void patch_p16_kernel(const float* __restrict__ src, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        __bf16* __restrict__ out16, double* __restrict__ spart, int B,
                                                        int H, int W, int pad, LgNormFuse nf = LgNormFuse{}) {
  static_assert(!FUSE || (OUT16 && !STATS), "norm-backward sums: bf16 data-gradient form only");
  // 64 columns: the waves split the CHANNELS as well as the tile rows (wave w: channel half w & 1, rows 8 (w >> 1) .. + 7) — half the
  // resident weight fragments per wave, four blocks per CU instead of three (round 4: the kernel follows its occupancy, not its
  // instruction count — every ablation that freed registers gained, the rewrite of the fragment build alone did not)
  constexpr bool CSPLIT = N == 64;
  constexpr int NWC = CSPLIT ? N / 2 : N;          // channels per wave
  constexpr int NT = NWC / 16, HSIDE = S * (TS - 1) + 5, PW = (HSIDE + 2) & ~1, TROWS = HSIDE + 1, NKS = 4;
  constexpr int MTW = CSPLIT ? TS / 2 : TS / 4;  // m-tiles (tile rows of 16 pixels) per wave
  constexpr bool HAS_BIAS = S == 2;   // the stride-1 form is the final layer's data gradient: no bias term
  // TROWS x PW pixels of 8 bytes; row HSIDE and the pixels right of column HSIDE - 1 stay zero (touched by zero-weight slots only)
  __shared__ __attribute__((aligned(16))) bf16x4 tile[TROWS * PW];
  __shared__ __attribute__((aligned(16))) float cst[4][16 * NWC];
  __shared__ double sred[40];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
  const int tpx = W / TS, tpi = tpx * (H / TS), ntiles = B * tpi;
  const int Hs = S * H, Ws = S * W;
  const int chw = CSPLIT ? (wid & 1) : 0, rw = CSPLIT ? (wid >> 1) : wid;   // channel half, row group of this wave

  // B fragments (constant for the block): lane (n = l&15, g) holds k' = 32 ks + 8 g + j = 24 ky + 4 kx + c4
  bf16x8 bf[NKS][NT];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kp = 32 * ks + 8 * g + j, ky = kp / 24, rem = kp - 24 * ky, kx = rem >> 2, c4 = rem & 3;
        v[j] = (ky < 5 && kx < 5 && c4 < 3) ? (__bf16)w[(long long)(ky * 15 + kx * 3 + c4) * N + chw * NWC + nt * 16 + r] : (__bf16)0.f;
      }
      bf[ks][nt] = v;
    }
  // A fragment of k-step ks: pixels (S r + 2 kxp, + 1) of tile row S ly + ky, (ky, kxp) = (k' / 24, (k' % 24) / 8), k' = 32 ks + 8 g
  int aoff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int kp = 32 * ks + 8 * g, ky = kp / 24, kxp = (kp - 24 * ky) >> 3;
    aoff[ks] = ky * PW + S * r + 2 * kxp;
  }
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = bias ? bias[chw * NWC + nt * 16 + r] : 0.f;
  for (int i = threadIdx.x; i < TROWS * PW; i += 256) tile[i] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};

  // the halo of the NEXT tile is requested (global -> registers) before this tile's MFMAs and written to LDS behind them:
  // the block no longer sits out a global-memory latency per tile (measured: 66-71 % of the wave cycles were parked)
  constexpr int NPT = (HSIDE * HSIDE + 255) / 256;  // halo pixels per thread
  float hv[NPT][3];
  auto halo_load = [&](int t) {
    const int n = t / tpi, tt = t - n * tpi;
    const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
      const int i = threadIdx.x + u * 256, hy = i / HSIDE, hx = i - hy * HSIDE;
      const int sy = S * y0 - pad + hy, sx = S * x0 - pad + hx;
      hv[u][0] = 0.f; hv[u][1] = 0.f; hv[u][2] = 0.f;
      if (i < HSIDE * HSIDE && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws) {
        const float* q = src + ((long long)(n * Hs + sy) * Ws + sx) * 3;
        if constexpr (!(LG_P16_DBG & 1)) { hv[u][0] = q[0]; hv[u][1] = q[1]; hv[u][2] = q[2]; }
        else { hv[u][0] = (float)sy; hv[u][1] = (float)sx; }
      }
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < HSIDE * HSIDE) {
        const int hy = i / HSIDE, hx = i - hy * HSIDE;
        tile[hy * PW + hx] = bf16x4{(__bf16)hv[u][0], (__bf16)hv[u][1], (__bf16)hv[u][2], (__bf16)0.f};   // RNE, as the fragment build did
      }
    }
  };
  if ((int)blockIdx.x < ntiles) halo_load(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n = t / tpi, tt = t - n * tpi;
    const int y0 = (tt / tpx) * TS, x0 = (tt % tpx) * TS;
    __syncthreads();  // previous tile fully consumed (first tile: the zero fill is complete)
    halo_store();
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) halo_load(t + gridDim.x);

    // one m-tile (a tile row of 16 pixels) at a time: MFMAs -> + bias -> moments -> wave-private LDS transpose -> whole
    // contiguous rows out, 16 B per lane.  No block barrier in here: LDS operations of one wave execute in order.
    float nf1 = 0.f, nf2 = 0.f;  // FUSE: norm-backward sums of the gradient this tile writes (lg_common.h)
    (void)nf1; (void)nf2;
    f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};  // sum and sum of squares about `shift` (the first bias: close enough to the block mean)
    const float shift = STATS ? (bias ? bias[0] : 0.f) : 0.f;
    float* cw = cst[wid];
    // FUSE: the z pieces of the wave's MTW tile rows are requested HERE, before the matrix work — read where they are used, each
    // was a load waited for at once (one 1-KB request in flight per wave, 16 KB per CU, and the form cost 60 us over the plain one)
    constexpr int NQZ = FUSE ? 16 * N / 8 / 64 : 1;
    u32x4 zpre[FUSE ? MTW * NQZ : 1];
    (void)zpre;
    if constexpr (FUSE) {
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int q = 0; q < NQZ; ++q)
          zpre[i * NQZ + q] = *reinterpret_cast<const u32x4*>(nf.z + ((long long)(n * H + y0 + rw * MTW + i) * W + x0) * N + (q * 64 + lane) * 8);
    }
    auto mtile = [&](auto i_c) {
      const int i = i_c;   // FUSE: a compile-time constant (the prefetched pieces are registers, not a runtime-indexed array)
      const int ly = rw * MTW + i;
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 a[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x4* ap = tile + S * ly * PW + aoff[ks];
        if constexpr (S == 2) a[ks] = *reinterpret_cast<const bf16x8*>(ap);   // pixel index even: 16-B aligned
        else {
          const bf16x4 lo = ap[0], hi = ap[1];
          a[ks] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (!(LG_P16_DBG & 4)) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], bf[ks][nt], acc[nt], 0, 0, 0);
          else acc[nt][0] += (float)a[ks][0] * (float)bf[ks][nt][0];
        }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {   // two accumulator rows at a time: v_pk_add_f32 / v_pk_fma_f32 halve the VALU count
          f32x2 v = {acc[nt][2 * e2], acc[nt][2 * e2 + 1]};
          if constexpr (HAS_BIAS) v += f32x2{bv[nt], bv[nt]};
          if constexpr (STATS && !(LG_P16_DBG & 8)) {
            const f32x2 d = v - f32x2{shift, shift};
            s1v += d;
            s2v = __builtin_elementwise_fma(d, d, s2v);
          }
          cw[(4 * g + 2 * e2) * NWC + nt * 16 + r] = v[0];  // C layout: col = lane&15, row = 4(lane>>4)+e
          cw[(4 * g + 2 * e2 + 1) * NWC + nt * 16 + r] = v[1];
        }
      __builtin_amdgcn_wave_barrier();
      const long long o0 = ((long long)(n * H + y0 + ly) * W + x0) * N;  // 16 pixels x N contiguous
      if constexpr (OUT16) {
#pragma unroll
        for (int q = 0; q < 16 * NWC / 8 / 64; ++q) {
          const int idx = q * 64 + lane;
          const int gix = CSPLIT ? (idx / (NWC / 8)) * N + chw * NWC + (idx % (NWC / 8)) * 8 : idx * 8;   // element offset inside the 16-pixel row segment
          const f32x4 a = *reinterpret_cast<const f32x4*>(cw + idx * 8), b = *reinterpret_cast<const f32x4*>(cw + idx * 8 + 4);
          bf16x8 v;
          v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
          v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
          if constexpr (!(LG_P16_DBG & 2)) *reinterpret_cast<bf16x8*>(out16 + o0 + gix) = v;
          else if (v[0] == (__bf16)12345.f) *reinterpret_cast<bf16x8*>(out16 + o0 + gix) = v;
          if constexpr (FUSE) {
            const lg_const_f32p sp = lg_as_const(nf.stats + (long long)n * 8);   // scalar loads (lg_common.h)
            lg_nf_accum(__builtin_bit_cast(u32x4, v), zpre[i * NQZ + q], sp[0], sp[4], sp[2], sp[3], nf.alpha, nf1, nf2);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16 * NWC / 4 / 64; ++q) {
          const int idx = q * 64 + lane;
          const int gix = CSPLIT ? (idx / (NWC / 4)) * N + chw * NWC + (idx % (NWC / 4)) * 4 : idx * 4;
          *reinterpret_cast<f32x4*>(out + o0 + gix) = *reinterpret_cast<const f32x4*>(cw + idx * 4);
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    if constexpr (FUSE) {   // unrolled: the prefetched pieces are registers, not a runtime-indexed array
      mtile(std::integral_constant<int, 0>{}); mtile(std::integral_constant<int, 1>{});
      mtile(std::integral_constant<int, 2>{}); mtile(std::integral_constant<int, 3>{});
      static_assert(MTW == 4 && !CSPLIT, "four tile rows per wave");
    } else {
#pragma unroll 1
      for (int i = 0; i < MTW; ++i) mtile(i);   // a runtime row index: one copy of the body
    }

    if constexpr (FUSE) {  // one record per (sample, tile): S1 = sum g', S2 = sum g' c
      double red[2] = {(double)nf1, (double)nf2};
      lg_block_sum_d<2>(red, sred);
      if (threadIdx.x == 0) {
        double* o = nf.part + ((long long)n * tpi + tt) * 2;
        o[0] = red[0]; o[1] = red[1];
      }
    }
    if constexpr (STATS) {  // moments of this block's 256 x N outputs (one sample per block): {count, mean, M2}
      const double cnt = 256.0 * N;
      double red[2] = {(double)s1v[0] + (double)s1v[1], (double)s2v[0] + (double)s2v[1]};
      lg_block_sum_d<2>(red, sred);
      if (threadIdx.x == 0) {
        double* o = spart + ((long long)n * tpi + tt) * 3;
        const double md = red[0] / cnt;  // mean - shift
        o[0] = cnt; o[1] = (double)shift + md; o[2] = red[1] - cnt * md * md;
      }
    }
  }
}

}  // namespace

// ---- entry points (LG_ERR_UNSUPPORTED -> the caller falls back to the fp32-source kernels) ----
// persistent grid of the patch kernels: a block sets up 12-24 weight fragments (8 converted weights each per lane) and then
// walks its tiles with the next halo in flight, so it should own several tiles; 3 blocks fit a CU (160 VGPRs)
static int patch_grid(int ntiles, int per_cu) {  // per_cu: resident blocks per CU of the instantiation (VGPR-limited)
  static int cus = 0, forced = -1;
  if (!cus) {
    const char* e = getenv("LG_PATCH_GRID");
    forced = e ? atoi(e) : 0;
    int dev = 0;
    hipDeviceProp_t pr;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
  }
  const int nb = forced > 0 ? forced : per_cu * (cus < lg_grid_cus() ? cus : lg_grid_cus());
  return ntiles < nb ? ntiles : nb;
}

extern "C" int lg_n3_p16_supported(int H, int W, int C) { return (H % TS == 0 && W % TS == 0 && (C == 32 || C == 64)) ? 1 : 0; }

extern "C" int lg_n3_s1t_fwd_p16_try(const void* x16, const float* w, const float* bias, float* y, int B, int H, int W,
                                     int C, void* stream) {
  if (!lg_n3_p16_supported(H, W, C) || !x16 || !w || !bias) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(B * (H / TS) * (W / TS));
  if (C == 32) hipLaunchKernelGGL(s1t_fwd_p16_kernel<32>, grid, dim3(256), 0, st, (const __bf16*)x16, w, bias, y, B, H, W);
  else hipLaunchKernelGGL(s1t_fwd_p16_kernel<64>, grid, dim3(256), 0, st, (const __bf16*)x16, w, bias, y, B, H, W);
  LG_CHECK_LAUNCH("lg_n3_s1t_fwd_p16");
  lg_note_kernel("s1t_fwd_p16_kernel");
  return LG_OK;
}

extern "C" int lg_n3_up_p16_try(const void* src16, const float* w, float* out, int B, int H, int W, int C, void* stream) {
  if (!lg_n3_p16_supported(H, W, C) || !src16 || !w) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(patch_grid(B * (H / TS) * (W / TS), 2));  // 228-244 VGPRs: 2 blocks per CU (measured 4096 / 1024 / 768 / 512 blocks: 96 / 63 / 77 / 55 us)
  if (C == 32) hipLaunchKernelGGL(up_p16_kernel<32>, grid, dim3(256), 0, st, (const __bf16*)src16, w, out, B, H, W);
  else hipLaunchKernelGGL(up_p16_kernel<64>, grid, dim3(256), 0, st, (const __bf16*)src16, w, out, B, H, W);
  LG_CHECK_LAUNCH("lg_n3_up_p16");
  lg_note_kernel("up_p16_kernel");
  return LG_OK;
}

extern "C" int lg_n3_conv1_p16_supported(int H, int W, int N) { return (H % TS == 0 && W % TS == 0 && N == 64) ? 1 : 0; }

// conv1 forward from the fp32 image (bf16 MFMA), with the per-block InstanceNorm moments; *nparts = records per sample
extern "C" int lg_n3_conv1_fwd_p16_try(const float* img, const float* w, const float* bias, float* z, void* z16, int B, int H,
                                       int W, int N, void* spart, size_t spart_bytes, int* nparts, void* stream) {
  if (nparts) *nparts = 0;
  if (H % TS || W % TS || N != 64 || !img || !w || (!z && !z16)) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int tpi = (H / TS) * (W / TS), ntiles = B * tpi;
  const dim3 grid(patch_grid(ntiles, 4));  // channel-split waves, <= 128 VGPRs: 4 blocks per CU (round 3, 160 VGPRs: 4096 / 2048 / 1024 / 768 / 512 blocks: 93 / 75 / 72 / 59 / 66 us)
  const bool stats = spart && nparts && (size_t)B * tpi * 3 * sizeof(double) <= spart_bytes;
  if (z16) {  // bf16 activation path: z leaves as bf16 (the moments still come from the fp32 accumulators)
    if (stats) hipLaunchKernelGGL((patch_p16_kernel<2, 64, true, true>), grid, dim3(256), 0, st, img, w, bias, nullptr, (__bf16*)z16, (double*)spart, B, H, W, 1);
    else hipLaunchKernelGGL((patch_p16_kernel<2, 64, true, false>), grid, dim3(256), 0, st, img, w, bias, nullptr, (__bf16*)z16, nullptr, B, H, W, 1);
  } else if (stats) hipLaunchKernelGGL((patch_p16_kernel<2, 64, false, true>), grid, dim3(256), 0, st, img, w, bias, z, nullptr, (double*)spart, B, H, W, 1);
  else hipLaunchKernelGGL((patch_p16_kernel<2, 64, false, false>), grid, dim3(256), 0, st, img, w, bias, z, nullptr, nullptr, B, H, W, 1);
  LG_CHECK_LAUNCH("lg_n3_conv1_fwd_p16");
  lg_note_kernel("patch_p16_kernel<2,64>");
  if (stats) *nparts = tpi;
  return LG_OK;
}

// data gradient of the final stride-1 layer: dpre [B,H,W,3] fp32 -> dx [B,H,W,32] as bf16 (dx16) or fp32 (dx)
extern "C" int lg_n3_s1_dgrad_p16_nf_try(const float* dpre, const float* w, float* dx, void* dx16, int B, int H, int W, int N,
                                         const LgNormFuse* nf, size_t nf_bytes, int* nparts_out, void* stream);
extern "C" int lg_n3_s1_dgrad_p16_try(const float* dpre, const float* w, float* dx, void* dx16, int B, int H, int W, int N,
                                      void* stream) {
  return lg_n3_s1_dgrad_p16_nf_try(dpre, w, dx, dx16, B, H, W, N, nullptr, 0, nullptr, stream);
}
// nf (optional, bf16 output): also the norm-backward sums of the produced gradient, [B][*nparts_out][2] doubles
extern "C" int lg_n3_s1_dgrad_p16_nf_try(const float* dpre, const float* w, float* dx, void* dx16, int B, int H, int W, int N,
                                         const LgNormFuse* nf, size_t nf_bytes, int* nparts_out, void* stream) {
  if (nparts_out) *nparts_out = 0;
  if (H % TS || W % TS || N != 32 || !dpre || !w || (!dx && !dx16)) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int tpi_ = (H / TS) * (W / TS);
  const int ntiles = B * tpi_;
  const dim3 grid(patch_grid(ntiles, 4));  // 80-104 VGPRs (measured with the fused sums: 168 / 154 / 146 / 161 / 200 us)
  if (dx16 && nf && nf->z && nf->stats && nf->part && nparts_out && (size_t)B * tpi_ * 2 * sizeof(double) <= nf_bytes) {
    hipLaunchKernelGGL((patch_p16_kernel<1, 32, true, false, true>), grid, dim3(256), 0, st, dpre, w, nullptr, nullptr, (__bf16*)dx16, nullptr, B, H, W, 2, *nf);
    LG_CHECK_LAUNCH("lg_n3_s1_dgrad_p16(nf)");
    lg_note_kernel("patch_p16_kernel<1,32,nf>");
    *nparts_out = tpi_;
    return LG_OK;
  }
  if (dx16) hipLaunchKernelGGL((patch_p16_kernel<1, 32, true, false>), grid, dim3(256), 0, st, dpre, w, nullptr, nullptr, (__bf16*)dx16, nullptr, B, H, W, 2);
  else hipLaunchKernelGGL((patch_p16_kernel<1, 32, false, false>), grid, dim3(256), 0, st, dpre, w, nullptr, dx, nullptr, nullptr, B, H, W, 2);
  LG_CHECK_LAUNCH("lg_n3_s1_dgrad_p16");
  lg_note_kernel("patch_p16_kernel<1,32>");
  return LG_OK;
}
