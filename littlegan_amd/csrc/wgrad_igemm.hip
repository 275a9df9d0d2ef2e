// Weight-gradient contraction for the 5x5 stride-2 layers (and the 3-channel "patch" layers):
//   dW[ky,kx,cb,cs] = sum_{n,y,x} big[n, 2y+ky-1, 2x+kx-1, cb] * small[n, y, x, cs]
// big = the 2H-sized tensor (conv input x, or convT output-gradient), small = the H-sized one
// (conv output-gradient, or convT input).  Same memory layout for tf Conv2D (HWIO) and
// Conv2DTranspose (HWOI) kernels — /root/reference/model.py:15,39-40.
// GEMM view per tap: M = Cb, N = Cs, K = pixels (B*Hm*Wm), split-K over blocks; each block writes
// its own fp32 slab, a second kernel reduces slabs in fixed order (deterministic, no atomics).
// Both operands are "k-major" in memory (pixel rows, channels contiguous) which is exactly the
// v_mfma_f32_32x32x2_f32 operand shape: lane (r,h) reads LDS[pixel 2kk+h][channel r].
// bf16 variant: LDS holds bf16 [pixel][channel]; fragments come from ds_read_b64_tr_b16.
#include <stdlib.h>
#include "lg_common.h"

namespace {

struct WgradParams {
  const float* big;
  const float* small;
  const __bf16* big16;    // optional bf16 mirrors (SRC16 instantiations): same layouts
  const __bf16* small16;
  float* slab;
  int B, Hm, Wm, Cb, Cs, M;
  int Cbp;       // slab rows per tap (Cb, or 16 in patch mode)
  int chunk;     // pixels per split, multiple of KP
  int nti, ntj, ntaps;
  int pstride, ppad;
};

// pixels per k tile.  The 128x128 bf16 tile from the mirrors uses 32 (two MFMA k steps per barrier): its double-buffered
// operand image is then 34.8 KB instead of 69.6 KB and THREE blocks stay resident per CU (168 VGPRs) instead of two.
constexpr int wg_kp(bool bf16, bool src16, int mtnt) { return bf16 ? ((src16 && mtnt == 4) ? 32 : 64) : 32; }
// LDS row stride (bytes) of a [pixel][channels] bf16 operand image read with ds_read_b64_tr_b16: a 32-lane group reads
// 4 consecutive rows x 64 B (16 banks) each, so the rows must sit 16 banks apart: stride = 64 or 192 (mod 256).  The
// former 16-B pad left them 4 banks apart (measured: SQ_LDS_BANK_CONFLICT = 60 % of the LDS-active cycles).
// Measured: the 128x128 tile 251 -> 228 us with the 64-B pad; the smaller tiles got SLOWER with it (less LDS headroom,
// store conflicts), so they keep the 16-B pad.
constexpr int wg_rs(bool bf16, int ch, bool sq128) { return bf16 ? (sq128 ? ch * 2 + 64 : ch * 2 + 16) : ch * 4; }

template <bool BF16, bool PATCH, bool SRC16, int WI, int WJ, int MT, int NT>
__global__ __launch_bounds__(64 * WI * WJ) void wgrad_kernel(const WgradParams p) {
  static_assert(!SRC16 || (BF16 && !PATCH), "bf16 sources only for the bf16 MFMA, non-patch kernel");
  constexpr int NTHR = 64 * WI * WJ;
  constexpr int BI = WI * MT * 32, BJ = WJ * NT * 32;
  constexpr int KP = wg_kp(BF16, SRC16, MT * NT);  // pixels per k tile
  constexpr int RSA = wg_rs(BF16, BI, BI == 128 && BJ == 128);   // LDS row strides (bytes)
  constexpr int RSB = wg_rs(BF16, BJ, BI == 128 && BJ == 128);
  constexpr int LA = SRC16 ? BI / 8 : BI / 4, LB = SRC16 ? BJ / 8 : BJ / 4;  // 16-B lanes per row
  static_assert(NTHR % LA == 0 && NTHR % LB == 0, "staging geometry");
  constexpr int RA = NTHR / LA, RB = NTHR / LB;  // rows per pass
  constexpr int PA = KP / RA, PB = KP / RB;
  static_assert(PA >= 1 && PB >= 1 && KP % RA == 0 && KP % RB == 0, "staging geometry");
  constexpr int A_BYTES = KP * RSA, B_BYTES = KP * RSB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA0 = smem;
  char* sB0 = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wid / WJ, wj = wid % WJ;

  // logical block order: tap fastest, then channel tile, then pixel split; consecutive logical ids share an XCD
  // (lg_xcd_remap), so the 25 taps that re-read the same pixel chunk of `big` / `small` run together on one L2
  // instead of pulling the chunk from HBM 25 times.
  int bx = lg_xcd_remap(blockIdx.x, gridDim.x);
  const int t = bx % p.ntaps; bx /= p.ntaps;  // tap
  const int tj = bx % p.ntj; bx /= p.ntj;
  const int ti = bx % p.nti; bx /= p.nti;
  const int ky = PATCH ? t : t / 5, kx = PATCH ? 0 : t - (t / 5) * 5;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int split = bx;
  const int kbeg = split * p.chunk;
  const int kend = min(kbeg + p.chunk, p.M);
  const int nk = (kend - kbeg + KP - 1) / KP;
  const int HWm = p.Hm * p.Wm;
  const int Hb = (PATCH ? p.pstride : 2) * p.Hm, Wb = (PATCH ? p.pstride : 2) * p.Wm;

  const int arow = tid / LA, alc = tid % LA;
  const int brow = tid / LB, blc = tid % LB;
  f32x4 ra[PA], rb[PB];

  auto load_tile = [&](int it) {
    const int k0 = kbeg + it * KP;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int m = k0 + q * RA + arow;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < kend) {
        const int n = m / HWm, rem = m - n * HWm;
        const int y = rem / p.Wm, x = rem - y * p.Wm;
        if constexpr (PATCH) {
          const int sy = p.pstride * y + ky - p.ppad;
          if ((unsigned)sy < (unsigned)Hb) {
            const int sx0 = p.pstride * x - p.ppad;
            const float* base = p.big + ((long long)(n * Hb + sy) * Wb) * 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int j = alc * 4 + e;
              const int sx = sx0 + j / 3;
              if (j < 15 && (unsigned)sx < (unsigned)Wb) v[e] = base[sx0 * 3 + j];
            }
          }
        } else {
          const int sy = 2 * y + ky - 1, sx = 2 * x + kx - 1;
          if ((unsigned)sy < (unsigned)Hb && (unsigned)sx < (unsigned)Wb) {
            if constexpr (SRC16)
              v = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(p.big16 + ((long long)(n * Hb + sy) * Wb + sx) * p.Cb + i0 + alc * 8));
            else
              v = *reinterpret_cast<const f32x4*>(p.big + ((long long)(n * Hb + sy) * Wb + sx) * p.Cb + i0 + alc * 4);
          }
        }
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int m = k0 + q * RB + brow;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < kend) {
        if constexpr (SRC16) v = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(p.small16 + (long long)m * p.Cs + j0 + blc * 8));
        else v = *reinterpret_cast<const f32x4*>(p.small + (long long)m * p.Cs + j0 + blc * 4);
      }
      rb[q] = v;
    }
  };

  auto store_tile = [&](int buf) {
    char* sA = sA0 + buf * A_BYTES;
    char* sB = sB0 + buf * B_BYTES;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      char* d = sA + (q * RA + arow) * RSA;
      if constexpr (SRC16) {
        *reinterpret_cast<f32x4*>(d + alc * 16) = ra[q];
      } else if constexpr (BF16) {
        bf16x4 w;
        w[0] = (__bf16)ra[q][0]; w[1] = (__bf16)ra[q][1]; w[2] = (__bf16)ra[q][2]; w[3] = (__bf16)ra[q][3];
        *reinterpret_cast<bf16x4*>(d + alc * 8) = w;
      } else {
        *reinterpret_cast<f32x4*>(d + alc * 16) = ra[q];
      }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      char* d = sB + (q * RB + brow) * RSB;
      if constexpr (SRC16) {
        *reinterpret_cast<f32x4*>(d + blc * 16) = rb[q];
      } else if constexpr (BF16) {
        bf16x4 w;
        w[0] = (__bf16)rb[q][0]; w[1] = (__bf16)rb[q][1]; w[2] = (__bf16)rb[q][2]; w[3] = (__bf16)rb[q][3];
        *reinterpret_cast<bf16x4*>(d + blc * 8) = w;
      } else {
        *reinterpret_cast<f32x4*>(d + blc * 16) = rb[q];
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

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) load_tile(it + 1);
    const char* sA = sA0 + buf * A_BYTES;
    const char* sB = sB0 + buf * B_BYTES;
    if constexpr (!BF16) {
      const float* fa = reinterpret_cast<const float*>(sA) + wi * MT * 32 + r;
      const float* fb = reinterpret_cast<const float*>(sB) + wj * NT * 32 + r;
#pragma unroll 4
      for (int kk = 0; kk < KP / 2; ++kk) {
        const int pix = 2 * kk + h;
        float a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = fa[pix * BI + i * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = fb[pix * BJ + j * 32];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3
      // of a 4(row) x 16(col) block and receives column (lane&15) of the 4 rows.  For the 32x32x16 MFMA
      // lane (r,h) needs channel r at pixels 8h..8h+7 of the 16-pixel k step: two blocks (rows 8h..8h+3,
      // 8h+4..8h+7), columns 16*(r>>4) .. +15.
      const int g = lane >> 4;           // 16-lane group: (g&1) = channel half, (g>>1) = h
      const int lq = (lane & 15) >> 2, lp = lane & 3;
      const int colbase = 16 * (g & 1) + 4 * lp;
#pragma unroll
      for (int ks = 0; ks < KP / 16; ++ks) {
        const int row0 = ks * 16 + 8 * (g >> 1) + lq;
        bf16x8 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const char* pa = sA + row0 * RSA + ((wi * MT + i) * 32 + colbase) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 4 * RSA));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          a[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const char* pb = sB + row0 * RSB + ((wj * NT + j) * 32 + colbase) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb + 4 * RSB));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          b[j] = __builtin_bit_cast(bf16x8, v);
        }
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

  // slab[split][t][i][j]
  float* out = p.slab + ((long long)split * p.ntaps + t) * p.Cbp * p.Cs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = j0 + (wj * NT + j) * 32 + r;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = i0 + (wi * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < p.Cbp && col < p.Cs) out[(long long)row * p.Cs + col] = acc[i][j][e];
      }
    }
  }
}

// out[t][i<rows_v][j] (+)= sum_s slab[s][t][i][j]  (slab rows_p >= rows_v per tap)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int nsplit, int ntaps,
                                   int rows_p, int rows_v, int cols, int accumulate) {
  const long long n_out = (long long)ntaps * rows_v * cols;
  const long long slab_sz = (long long)ntaps * rows_p * cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
    const int j = (int)(o % cols);
    const long long rem = o / cols;
    const int i = (int)(rem % rows_v), t = (int)(rem / rows_v);
    const long long si = ((long long)t * rows_p + i) * cols + j;
    float s = accumulate ? out[o] : 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[k * slab_sz + si];
    out[o] = s;
  }
}

// same for slabs with no row padding (rows_p == rows_v): 16-B accesses, 4 slabs in flight per thread
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const f32x4* __restrict__ slab, f32x4* __restrict__ out,
                                                           int nsplit, long long n4, int accumulate) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 s = accumulate ? out[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
      const f32x4 a = slab[(long long)k * n4 + i], b = slab[(long long)(k + 1) * n4 + i];
      const f32x4 c = slab[(long long)(k + 2) * n4 + i], d = slab[(long long)(k + 3) * n4 + i];
      s += (a + b) + (c + d);
    }
    for (; k < nsplit; ++k) s += slab[(long long)k * n4 + i];
    out[i] = s;
  }
}

// db[c] (+)= sum_k partial[k][c]: 16 columns x 16 row groups per block, merged in group order (deterministic)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ db, int nb,
                                                           int C, int accumulate) {
  __shared__ float sr[16][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C) {
    int k = g;
    for (; k + 48 < nb; k += 64) {
      const float a = partial[(long long)k * C + c], b = partial[(long long)(k + 16) * C + c];
      const float e = partial[(long long)(k + 32) * C + c], f = partial[(long long)(k + 48) * C + c];
      s += (a + b) + (e + f);
    }
    for (; k < nb; k += 16) s += partial[(long long)k * C + c];
  }
  sr[g][cl] = s;
  __syncthreads();
  if (g == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sr[q][cl];
    db[c] = (accumulate ? db[c] : 0.f) + t;
  }
}

// column sums of a [M][C] matrix: partial[blk][C]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, const __bf16* __restrict__ x16,
                                                     float* __restrict__ partial, long long M, int C, long long rows_per_blk) {
  const int c4 = C / 4;                 // C % 4 == 0
  const int tpc = 256 / c4 > 0 ? 256 / c4 : 1;  // threads per column-quad along rows
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
  __shared__ f32x4 sred[256];
  for (int cq0 = 0; cq0 < c4; cq0 += 256) {
    const int cq = cq0 + (c4 >= 256 ? threadIdx.x : threadIdx.x % c4);
    const int rr = c4 >= 256 ? 0 : threadIdx.x / c4;
    const int nr = c4 >= 256 ? 1 : tpc;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (cq < c4 && rr < nr)
      for (long long m = r0 + rr; m < r1; m += nr) {
        if (x16) {
          const bf16x4 w = *reinterpret_cast<const bf16x4*>(x16 + m * C + cq * 4);
          s += f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
        } else {
          s += *reinterpret_cast<const f32x4*>(x + m * C + cq * 4);
        }
      }
    sred[threadIdx.x] = s;
    __syncthreads();
    if (rr == 0 && cq < c4) {
      for (int k = 1; k < nr; ++k)
        if (threadIdx.x + k * c4 < 256) s += sred[threadIdx.x + k * c4];
      *reinterpret_cast<f32x4*>(partial + (long long)blockIdx.x * C + cq * 4) = s;
    }
    __syncthreads();
  }
}

template <bool BF16, bool PATCH, bool SRC16, int WI, int WJ, int MT, int NT>
void launch_wgrad(WgradParams p, int nsplit, hipStream_t st) {
  constexpr int BI = WI * MT * 32, BJ = WJ * NT * 32, KP = wg_kp(BF16, SRC16, MT * NT);
  constexpr int RSA = wg_rs(BF16, BI, BI == 128 && BJ == 128), RSB = wg_rs(BF16, BJ, BI == 128 && BJ == 128);
  const size_t lds = 2 * (size_t)KP * (RSA + RSB);
  p.nti = lg_cdiv(p.Cbp, BI);
  p.ntj = lg_cdiv(p.Cs, BJ);
  auto kern = wgrad_kernel<BF16, PATCH, SRC16, WI, WJ, MT, NT>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(p.nti * p.ntj * p.ntaps * nsplit);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WI * WJ), lds, st, p);
  lg_note_kernel(PATCH ? "wgrad_kernel<PATCH>" : BF16 ? "wgrad_kernel<bf16,per-tap>" : "wgrad_kernel<f32,per-tap>");
}

struct TileSel { int bi, bj; };
inline TileSel pick_tile(int cbp, int cs) {
  TileSel s;
  s.bi = cbp % 128 == 0 ? 128 : (cbp % 64 == 0 ? 64 : 32);
  s.bj = cs % 128 == 0 ? 128 : (cs % 64 == 0 ? 64 : 32);
  if (s.bi == 64 && s.bj == 64) { /* 2x2 waves of 32x32 */ }
  return s;
}

// resident block slots of the whole chip for a tile shape (LDS-limited: 2 buffers of KP x (BI + BJ) operand rows)
inline int wgrad_slots(int bi, int bj, bool bf16_kernel) {
  const int KP = (bf16_kernel && bi == 128 && bj == 128) ? 32 : (bf16_kernel ? 64 : 32);  // wg_kp (mirror path assumed)
  const bool sq = bi == 128 && bj == 128;
  const size_t lds = 2 * (size_t)KP * (wg_rs(bf16_kernel, bi, sq) + wg_rs(bf16_kernel, bj, sq));
  int per_cu = (int)((160 * 1024) / lds);
  if (bf16_kernel && bi == 128 && bj == 128 && per_cu > 3) per_cu = 3;  // 168 VGPRs
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  return 256 * per_cu;
}

// split-K factor: the grid runs in rounds of `slots` blocks, each block costs its K range plus a fixed part (prologue,
// slab write) -> minimise rounds x (M/ns + K0); a 2.05-round grid wastes a third of the machine
inline int pick_nsplit(int tiles, int M, int KP, int slots) {
  const int maxs = M / (4 * KP) > 0 ? (M / (4 * KP) < 256 ? M / (4 * KP) : 256) : 1;
  const long long K0 = 1536;
  int best = 1;
  long long best_cost = -1;
  for (int ns = 1; ns <= maxs; ++ns) {
    const long long rounds = ((long long)tiles * ns + slots - 1) / slots;
    const long long cost = rounds * ((M + ns - 1) / ns + K0) + 24LL * ns;  // + the ordered reduce over ns slabs
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
  }
  return best;
}

}  // namespace

extern "C" size_t lg_n3_wgrad_workspace_bytes(int B, int H, int W, int Cs);
extern "C" int lg_n3_wgrad_try(const float* big3, const float* small, const void* small16, float* dw, void* workspace,
                               size_t ws_bytes, int B, int H, int W, int Cs, int s, int pad, int accumulate, void* stream);
extern "C" size_t lg_wgrad_at_workspace_bytes(int B, int Hm, int Wm, int cb, int cs);
extern "C" int lg_wgrad_at_try(const void* big16, const void* small16, void* workspace, size_t ws_bytes, int B, int Hm, int Wm,
                               int cb, int cs, int* nsplit_out, void* stream);
extern "C" size_t lg_wgrad_at32_workspace_bytes(int B, int Hm, int Wm, int cb, int cs);
extern "C" int lg_wgrad_at32_try(const float* big, const float* small, void* workspace, size_t ws_bytes, int B, int Hm, int Wm,
                                 int cb, int cs, int* nsplit_out, void* stream);
static size_t wgrad_ws_generic(int B, int Hm, int Wm, int cb, int cs, int dtype);

extern "C" size_t lg_wgrad_workspace_bytes(int B, int Hm, int Wm, int cb, int cs, int dtype) {
  size_t g = wgrad_ws_generic(B, Hm, Wm, cb, cs, dtype);
  if (cb == 3) { const size_t n = lg_n3_wgrad_workspace_bytes(B, Hm, Wm, cs); if (n > g) g = n; }
  else if (dtype == LG_DT_BF16) { const size_t n = lg_wgrad_at_workspace_bytes(B, Hm, Wm, cb, cs); if (n > g) g = n; }
  else { const size_t n = lg_wgrad_at32_workspace_bytes(B, Hm, Wm, cb, cs); if (n > g) g = n; }
  return g;
}

static size_t wgrad_ws_generic(int B, int Hm, int Wm, int cb, int cs, int dtype) {
  const int KP = dtype == LG_DT_BF16 ? 64 : 32;
  const int cbp = cb == 3 ? 16 : cb, ntaps = cb == 3 ? 5 : 25;
  TileSel ts = pick_tile(cb == 3 ? 32 : cbp, cs);
  const int tiles = lg_cdiv(cbp, ts.bi) * lg_cdiv(cs, ts.bj) * ntaps;
  const int ns = pick_nsplit(tiles, B * Hm * Wm, KP, wgrad_slots(ts.bi, ts.bj, dtype == LG_DT_BF16 && cb != 3));
  return (size_t)ns * ntaps * cbp * cs * sizeof(float);
}

// dW[5][5][cb][cs] (+)= big (x) small ; big [B,s*Hm,s*Wm,cb], small [B,Hm,Wm,cs].
// cb == 3 selects the patch form with source stride `pstride` and pad-before `ppad`
// (conv1: 2,1 ; stride-1 final layer: 1,2); otherwise stride 2 / pad 1.
extern "C" int lg_conv_wgrad_m16(const float* big, const void* big16, const float* small, const void* small16, float* dw,
                                 void* workspace, size_t ws_bytes, int B, int Hm, int Wm, int cb, int cs, int pstride,
                                 int ppad, int accumulate, int dtype, void* stream);

extern "C" int lg_conv_wgrad(const float* big, const float* small, float* dw, void* workspace, size_t ws_bytes,
                             int B, int Hm, int Wm, int cb, int cs, int pstride, int ppad, int accumulate, int dtype,
                             void* stream) {
  return lg_conv_wgrad_m16(big, nullptr, small, nullptr, dw, workspace, ws_bytes, B, Hm, Wm, cb, cs, pstride, ppad,
                           accumulate, dtype, stream);
}

// big16 / small16 (optional, both or none): bf16 mirrors of the operands, consumed by the bf16 MFMA kernel directly
extern "C" int lg_conv_wgrad_m16(const float* big, const void* big16, const float* small, const void* small16, float* dw,
                                 void* workspace, size_t ws_bytes, int B, int Hm, int Wm, int cb, int cs, int pstride,
                                 int ppad, int accumulate, int dtype, void* stream) {
  LG_CHECK_ARG(dw && workspace && ((big && small) || (big16 && small16 && dtype == LG_DT_BF16 && cb != 3) ||
                                   (cb == 3 && big && small16 && dtype == LG_DT_BF16)),
               "lg_conv_wgrad: null pointer (fp32 operands may be omitted only when both bf16 mirrors are given; "
               "3-channel layers: the bf16 mirror of `small` alone)");
  LG_CHECK_ARG(B > 0 && Hm > 0 && Wm > 0 && cs % 32 == 0 && (cb == 3 || cb % 32 == 0),
               "lg_conv_wgrad: bad shape B=%d Hm=%d Wm=%d cb=%d cs=%d", B, Hm, Wm, cb, cs);
  LG_CHECK_ARG(ws_bytes >= lg_wgrad_workspace_bytes(B, Hm, Wm, cb, cs, dtype), "lg_conv_wgrad: workspace too small");
  if (cb == 3 && !lg_env_flag("LG_NO_N3")) {  // all-taps 3-channel kernel (n3_kernels.hip) where its tiling applies
    const int rc = lg_n3_wgrad_try(big, small, dtype == LG_DT_BF16 ? small16 : nullptr, dw, workspace, ws_bytes, B, Hm, Wm,
                                   cs, pstride, ppad, accumulate, stream);
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  LG_CHECK_ARG((big && small) || (cb != 3 && big16 && small16), "lg_conv_wgrad: this shape needs the fp32 operands");
  if (cb != 3 && dtype == LG_DT_BF16 && big16 && small16) {  // all-taps kernel (wgrad_at.hip) on the 16x16 maps and larger
    int ns_at = 0;
    const int rc = lg_wgrad_at_try(big16, small16, workspace, ws_bytes, B, Hm, Wm, cb, cs, &ns_at, stream);
    if (rc == LG_OK) {
      const long long n4 = 25LL * cb * cs / 4;
      const int rb4 = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
      hipLaunchKernelGGL(slab_reduce4_kernel, dim3(rb4), dim3(256), 0, (hipStream_t)stream, (const f32x4*)workspace, (f32x4*)dw,
                         ns_at, n4, accumulate);
      LG_CHECK_LAUNCH("lg_conv_wgrad(reduce)");
      return LG_OK;
    }
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  if (cb != 3 && dtype == LG_DT_F32 && big && small && pstride == 2 && ppad == 1) {  // exact-f32 path: all-taps kernel (wgrad_at32.hip)
    int ns_at = 0;
    const int rc = lg_wgrad_at32_try(big, small, workspace, ws_bytes, B, Hm, Wm, cb, cs, &ns_at, stream);
    if (rc == LG_OK) {
      const long long n4 = 25LL * cb * cs / 4;
      const int rb4 = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
      hipLaunchKernelGGL(slab_reduce4_kernel, dim3(rb4), dim3(256), 0, (hipStream_t)stream, (const f32x4*)workspace, (f32x4*)dw,
                         ns_at, n4, accumulate);
      LG_CHECK_LAUNCH("lg_conv_wgrad(reduce)");
      return LG_OK;
    }
    if (rc != LG_ERR_UNSUPPORTED) return rc;
  }
  const bool patch = cb == 3;
  const bool bf16 = dtype == LG_DT_BF16 && !patch;  // patch layers stay on the exact f32 MFMA
  const int KP = bf16 ? 64 : 32;
  WgradParams p{};
  p.big = big; p.small = small; p.slab = (float*)workspace;
  p.big16 = bf16 ? (const __bf16*)big16 : nullptr; p.small16 = bf16 ? (const __bf16*)small16 : nullptr;
  p.B = B; p.Hm = Hm; p.Wm = Wm; p.Cb = cb; p.Cs = cs; p.M = B * Hm * Wm;
  p.Cbp = patch ? 16 : cb; p.ntaps = patch ? 5 : 25; p.pstride = pstride; p.ppad = ppad;
  TileSel ts = pick_tile(patch ? 32 : p.Cbp, cs);
  const int tiles = lg_cdiv(p.Cbp, ts.bi) * lg_cdiv(cs, ts.bj) * p.ntaps;
  const int KPws = dtype == LG_DT_BF16 ? 64 : 32;  // workspace sized with the caller's dtype
  int ns = pick_nsplit(tiles, p.M, KPws, wgrad_slots(ts.bi, ts.bj, bf16));
  p.chunk = (lg_cdiv(p.M, ns) + KP - 1) / KP * KP;
  ns = lg_cdiv(p.M, p.chunk);
  hipStream_t st = (hipStream_t)stream;
#define LG_WG(BF, PT, WI, WJ, MT, NT) launch_wgrad<BF, PT, false, WI, WJ, MT, NT>(p, ns, st)
#define LG_WG16(WI, WJ, MT, NT) launch_wgrad<true, false, true, WI, WJ, MT, NT>(p, ns, st)
  if (patch) {
    if (ts.bj == 128) LG_WG(false, true, 1, 2, 1, 2); else if (ts.bj == 64) LG_WG(false, true, 1, 2, 1, 1); else LG_WG(false, true, 1, 1, 1, 1);
  } else if (!bf16) {
    if (ts.bi == 128 && ts.bj == 128) LG_WG(false, false, 2, 2, 2, 2);
    else if (ts.bi == 128 && ts.bj == 64) LG_WG(false, false, 2, 2, 2, 1);
    else if (ts.bi == 128) LG_WG(false, false, 4, 1, 1, 1);
    else if (ts.bi == 64 && ts.bj == 128) LG_WG(false, false, 2, 2, 1, 2);
    else if (ts.bi == 64 && ts.bj == 64) LG_WG(false, false, 2, 2, 1, 1);
    else if (ts.bi == 64) LG_WG(false, false, 2, 1, 1, 1);
    else if (ts.bj == 128) LG_WG(false, false, 1, 2, 1, 2);
    else if (ts.bj == 64) LG_WG(false, false, 1, 2, 1, 1);
    else LG_WG(false, false, 1, 1, 1, 1);
  } else if (p.big16 && p.small16) {
    if (ts.bi == 128 && ts.bj == 128) LG_WG16(2, 2, 2, 2);
    else if (ts.bi == 128 && ts.bj == 64) LG_WG16(2, 2, 2, 1);
    else if (ts.bi == 128) LG_WG16(4, 1, 1, 1);
    else if (ts.bi == 64 && ts.bj == 128) LG_WG16(2, 2, 1, 2);
    else if (ts.bi == 64 && ts.bj == 64) LG_WG16(2, 2, 1, 1);
    else if (ts.bi == 64) LG_WG16(2, 1, 1, 1);
    else if (ts.bj == 128) LG_WG16(1, 2, 1, 2);
    else if (ts.bj == 64) LG_WG16(1, 2, 1, 1);
    else LG_WG16(1, 1, 1, 1);
  } else {
    if (ts.bi == 128 && ts.bj == 128) LG_WG(true, false, 2, 2, 2, 2);
    else if (ts.bi == 128 && ts.bj == 64) LG_WG(true, false, 2, 2, 2, 1);
    else if (ts.bi == 128) LG_WG(true, false, 4, 1, 1, 1);
    else if (ts.bi == 64 && ts.bj == 128) LG_WG(true, false, 2, 2, 1, 2);
    else if (ts.bi == 64 && ts.bj == 64) LG_WG(true, false, 2, 2, 1, 1);
    else if (ts.bi == 64) LG_WG(true, false, 2, 1, 1, 1);
    else if (ts.bj == 128) LG_WG(true, false, 1, 2, 1, 2);
    else if (ts.bj == 64) LG_WG(true, false, 1, 2, 1, 1);
    else LG_WG(true, false, 1, 1, 1, 1);
  }
#undef LG_WG
#undef LG_WG16
  LG_CHECK_LAUNCH("lg_conv_wgrad");
  const int rows_v = patch ? 15 : cb;
  const long long n_out = (long long)p.ntaps * rows_v * cs;
  if (rows_v == p.Cbp && n_out % 4 == 0 && (reinterpret_cast<size_t>(dw) & 15) == 0) {
    const long long n4 = n_out / 4;
    const int rb4 = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(rb4), dim3(256), 0, st, (const f32x4*)workspace, (f32x4*)dw, ns, n4,
                       accumulate);
  } else {
    const int rblocks = (int)((n_out + 255) / 256 < 2048 ? (n_out + 255) / 256 : 2048);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rblocks), dim3(256), 0, st, (const float*)workspace, dw, ns, p.ntaps,
                       p.Cbp, rows_v, cs, accumulate);
  }
  LG_CHECK_LAUNCH("lg_conv_wgrad(reduce)");
  return LG_OK;
}

extern "C" size_t lg_bias_grad_workspace_bytes(long long M, int C) {
  long long nb = (M + 255) / 256;
  if (nb > 512) nb = 512;
  return (size_t)nb * C * sizeof(float);
}

extern "C" int lg_bias_grad_m16(const float* dy, const void* dy16, float* db, void* workspace, size_t ws_bytes, long long M,
                                int C, int accumulate, void* stream);
// db[C] (+)= column sums of dy[M][C]
extern "C" int lg_bias_grad(const float* dy, float* db, void* workspace, size_t ws_bytes, long long M, int C,
                            int accumulate, void* stream) {
  return lg_bias_grad_m16(dy, nullptr, db, workspace, ws_bytes, M, C, accumulate, stream);
}
// same; if dy16 (bf16 [M][C]) is given it is read instead of dy (dy may then be null)
extern "C" int lg_bias_grad_m16(const float* dy, const void* dy16, float* db, void* workspace, size_t ws_bytes, long long M,
                                int C, int accumulate, void* stream) {
  LG_CHECK_ARG((dy || dy16) && db && workspace, "lg_bias_grad: null pointer");
  LG_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 4096, "lg_bias_grad: bad shape M=%lld C=%d", M, C);
  LG_CHECK_ARG(ws_bytes >= lg_bias_grad_workspace_bytes(M, C), "lg_bias_grad: workspace too small");
  long long nb = (M + 255) / 256;
  if (nb > 512) nb = 512;
  const long long rpb = (M + nb - 1) / nb;
  nb = (M + rpb - 1) / rpb;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_kernel, dim3((int)nb), dim3(256), 0, st, dy, (const __bf16*)dy16, (float*)workspace, M, C, rpb);
  LG_CHECK_LAUNCH("lg_bias_grad");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace, db, (int)nb, C,
                     accumulate);
  LG_CHECK_LAUNCH("lg_bias_grad(reduce)");
  return LG_OK;
}
