// Halo-tile implicit-GEMM convolution for gfx950: the "LDS staging of image tiles" of the north star.
//
// Same contractions and operand conventions as conv_igemm.hip (modes DOWN / UP / S1T), but the A operand is
// no longer re-gathered from global memory for every filter tap.  A block owns a TH x TW tile of the M grid
// (or NI whole images when the map is smaller than 128 pixels).  For each chunk of KC source channels it
// stages the source HALO of that tile once —  (s*TH+E) x (s*TW+E) pixels, E = 2 (UP), 3 (DOWN), 4 (S1T),
// zero-filled outside the image (TF "SAME") — and then runs ALL taps out of LDS: the MFMA A fragment of
// (row m, tap t) is halo row  hb(m) + dy_t*HW + dx_t,  a per-lane base plus a wave-uniform offset.
// Global->LDS activation traffic drops by the tap-reuse factor (UP 9x..4x, DOWN 4.8x, S1T 13x).  The weights (B
// operand) never touch LDS: the pack is stored in MFMA fragment order (pack.hip), so each wave fetches the 1-KiB
// operand of an MFMA with one coalesced global_load_dwordx4 per lane, two taps ahead, and the tap loop has NO
// barrier — the only block-wide synchronisation is the halo restage once per channel chunk.
// Falls back (LG_ERR_UNSUPPORTED) to the per-tap gather kernel for shapes the tiling does not cover.
#include <stdlib.h>
#include "lg_common.h"

extern "C" int lg_device_cus(void);

namespace {

enum { MODE_DOWN = 0, MODE_UP = 1, MODE_S1T = 2 };
// DOWN from the bf16 mirror with 32-channel chunks keeps its halo rows UNPADDED (64 B) and XOR-swizzled instead of
// padded to 80 B: 665 rows x 64 B = 42.5 KB, so THREE blocks fit a CU's 160 KB (the padded image allowed two), and the
// 128x32 wave tile of that variant is compiled for 3 waves per SIMD (161 VGPRs, no spills).  The A fragments of
// 8 consecutive lanes sit in rows R, R+2, .. (stride-2 conv): rows R and R+4 share their banks, so the 16-B slot is
// XORed with (row >> 2) & 3 -> conflict-free ds_read_b128.
// (selected per launch as the W3 template flag: worth it when the grid fills 3 slots per CU in fewer rounds)
template <typename T, int MODE, int KCH, bool SRC16, bool RES, int WAVES_M, int MT, int NT>
constexpr bool halo_w3_ok() {
  return SRC16 && !RES && sizeof(T) == 2 &&
         ((MODE == MODE_DOWN && KCH == 2 && ((WAVES_M == 1 && MT == 4 && NT == 1) || (WAVES_M == 2 && MT == 2 && NT == 1))) ||
          (MODE == MODE_UP && KCH == 4 && WAVES_M == 1 && MT == 4 && NT == 1));  // the tiles that fit 168 VGPRs (nearly) unspilled
}

struct HaloParams {
  const float* src;
  const __bf16* src16;         // optional bf16 mirror of src (same NHWC layout); used by the SRC16 instantiations
  const char* wp;
  const float* bias;
  float* out;
  __bf16* out16;  // if non-null the result is written as bf16 HERE instead of fp32 to `out`
  int B, Hs, Ws, Cs;
  int Hm, Wm;
  int Ho, Wo, N, Npad;
  int act;
  int ntn;
  int TH, TW, NI, tpi_x, tpi;  // tile geometry: tiles per image along x, tiles per image
  int HH, HW, HROWS, nrows;    // halo geometry (per image) and total halo rows
  int HWH, HWP;                // W3 / DOWN: half and full pitch (pixels) of a de-interleaved halo pixel row
  int dbg;                     // ablation switches for scripts/bench_layer.py (LG_DBG env; 0 in production)
  double* spart;               // fused InstanceNorm moments: [B][nparts][3] = {count, mean, M2} per block, or null
  int nparts;
  int dry;                     // 1: only answer whether this kernel covers the shape (no launch)
  int res_budget;              // LDS bytes the resident-halo (RES) variant may use; 0 = variant off
  int cfg;                     // tile-shape switches (LG_CFG env, A/B)
  int* nparts_host;            // host-side: receives nparts of the launched tiling (moments epilogue on)
  int epi_rows;                // rows per pass of the LDS-transposed epilogue (0 = straight from the accumulators)
  int ksplit;                  // 1, or 2: blockIdx.z takes one half of the channel chunks and ADDS its tile into the zeroed output (see launch)
  int clsorder;                // UP: 1 = blockIdx.y -> class 3, 2, 0, 1 instead of 3, 2, 1, 0 (see the kernel)
  int ntm, mfast;              // row tiles of the grid; mfast = 1: consecutive blocks of an XCD walk the ROW tiles of one column tile (see the kernel)
};

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int ESZ = 4; };
template <> struct DT<__bf16> { static constexpr int ESZ = 2; };

// ds_read_b128 is serviced in four NON-contiguous 16-lane groups: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same
// +32 (MI355X_MICROARCH.md, LDS).  A group is conflict-free when its 16 lanes hit 16 distinct 16-B slots of the 256-B
// bank row — e.g. 16 CONSECUTIVE halo rows of the padded image.  Which tile pixel an MFMA row stands for is free, so
// rows are assigned such that each lane group owns one 16-pixel tile row: MFMA row r of a 32-row fragment is tile
// pixel pix32(r) (group 0 -> pixels 0..15, group 1 -> pixels 16..31).  With the identity mapping a group straddled two
// tile rows and two of its sixteen slots collided (measured: SQ_LDS_BANK_CONFLICT = 53 % of the LDS-active cycles).
__device__ __forceinline__ int pix32(int r) {
  const int q = r >> 2, lo = r & 3;                  // q = 0..7
  // group 0: q in {0,3,5,6} -> slots 0,1,2,3 ; group 1: q in {1,2,4,7} -> slots 0,1,2,3 (ranked), + 16
  const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;     // parity of q: 0 -> group 0, 1 -> group 1
  const int rank = odd ? ((q == 1) ? 0 : (q == 2) ? 1 : (q == 4) ? 2 : 3) : ((q == 0) ? 0 : (q == 3) ? 1 : (q == 5) ? 2 : 3);
  return odd * 16 + rank * 4 + lo;
}

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

// RES ("resident"): UP mode from the bf16 mirror with the halo of ALL Cs channels staged once; the block then runs the
// four parity classes one after the other out of that image (grid.y == 1): a quarter of the source traffic and no
// barrier at all inside the (class, chunk, tap) loops.
template <typename T, int MODE, int KCH, bool DBUF, bool SRC16, bool RES, int WAVES_M, int WAVES_N, int MT, int NT, bool W3 = false>
#ifndef LG_EXP_W3OCC
#define LG_EXP_W3OCC 3
#endif
__global__ __launch_bounds__(256, (W3 ? LG_EXP_W3OCC : 2)) void conv_halo_kernel(const HaloParams p) {
  static_assert(!W3 || halo_w3_ok<T, MODE, KCH, SRC16, RES, WAVES_M, MT, NT>(), "W3: DOWN, 32-channel chunks, bf16 mirror, 128x32 / 64x32 wave tiles");
  static_assert(!SRC16 || (sizeof(T) == 2 && !DBUF), "bf16 source only with bf16 MFMA, single-buffered halo");
  static_assert(!RES || (MODE == MODE_UP && (SRC16 || sizeof(T) == 4)), "resident halo: UP mode, from the bf16 mirror or (round 5) exact f32");
  constexpr int ESZ = DT<T>::ESZ;
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
  static_assert(BM == 128, "halo tiles are 128 rows");
  constexpr int KC = KCH * 32 / ESZ;
  constexpr bool SWZ = W3 && MODE == MODE_DOWN;
  const int ROWB = RES ? p.Cs * ESZ + 16 : (SWZ ? KCH * 32 : KCH * 32 + 16);    // LDS bytes per halo row (RES: all channels)
  const int LPR = RES ? (SRC16 ? p.Cs / 8 : p.Cs / 4) : (SRC16 ? KC / 8 : KC / 4);  // threads per halo row (16 B each: 4 fp32 or 8 bf16 channels)
  const int RPP = 256 / LPR;         // halo rows per pass
  constexpr int SS = (MODE == MODE_DOWN) ? 2 : 1;
  constexpr int LO = (MODE == MODE_S1T) ? -2 : -1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NCLS = RES ? 4 : 1;
  int* s_out = reinterpret_cast<int*>(smem);          // [NCLS][128] output pixel per M row (RES: per class)
  int* s_hoff = s_out + NCLS * BM;                    // [nrows] source pixel index or -1
  char* sH = smem + ((NCLS * BM + p.nrows) * 4 + 15) / 16 * 16;   // [DBUF ? 2 : 1][nrows][ROWB]
  const int HBYTES = p.nrows * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  // UP: blockIdx.y -> parity class, heaviest first (9, 6, 6, 4 taps).  A grid of about one round (512 blocks: the exact-f32 layers at
  // B = 64) is placed breadth first, so the blocks of y and y + 2 share a CU: with the order 3, 2, 1, 0 that is (9 + 6 | 6 + 4) taps per
  // SIMD; clsorder puts the 4-tap class third — (9 + 4 | 6 + 6) (round 5).
  int cls = (MODE == MODE_UP) ? (p.clsorder ? ((0x1023 >> (4 * (int)blockIdx.y)) & 3) : 3 - (int)blockIdx.y) : 0;  // RES: set per pass of the class loop below
  int py = cls >> 1, px = cls & 1;
  const int lb = lg_xcd_remap(blockIdx.x, gridDim.x);
  // lb -> (row tile, column tile).  Column tile fastest (the first form): the blocks an XCD holds at one time share their SOURCE tiles and
  // stream the WHOLE weight tensor between them — fine while that fits the XCD's 4 MB of L2.  mfast: row tile fastest — an XCD's blocks share
  // one or two column slices of the weights and read disjoint source tiles (launch() sets it where the weights are the larger stream).
  const int tile_n = p.mfast ? lb / p.ntm : lb % p.ntn, tile_m = p.mfast ? lb - tile_n * p.ntm : lb / p.ntn;
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
    const int mp = (tid & ~31) + pix32(tid & 31);  // table index = MFMA row; its tile pixel
    const int i = mp / THW, rem = mp - i * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    const int n = img0 + i, y = y0 + ly, x = x0 + lx;
    if constexpr (RES) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s_out[c * BM + tid] = n < p.B ? (n * p.Ho + 2 * y + (c >> 1)) * p.Wo + 2 * x + (c & 1) : -1;
    } else {
      int o = -1;
      if (n < p.B) {
        const int oy = (MODE == MODE_UP) ? 2 * y + py : y;
        const int ox = (MODE == MODE_UP) ? 2 * x + px : x;
        o = (n * p.Ho + oy) * p.Wo + ox;
      }
      s_out[tid] = o;
    }
  }
  for (int hr = tid; hr < p.nrows; hr += 256) {
    const int i = hr / p.HROWS, rem = hr - i * p.HROWS;
    int hy, hx;
    bool hv = true;
    if constexpr (SWZ) {  // de-interleaved image: a halo pixel row holds its even columns, then its odd columns (HWH each)
      hy = rem / p.HWP;
      const int hxp = rem - hy * p.HWP;
      hx = hxp < p.HWH ? 2 * hxp : 2 * (hxp - p.HWH) + 1;
      hv = hx < p.HW;
    } else {
      hy = rem / p.HW; hx = rem - hy * p.HW;
    }
    const int n = img0 + i, sy = SS * y0 + LO + hy, sx = SS * x0 + LO + hx;
    int o = -1;
    if (hv && n < p.B && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws) o = (n * p.Hs + sy) * p.Ws + sx;
    s_hoff[hr] = o;
  }
  // per-lane halo base row of the MT fragments this wave reads
  int hb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = (wm * MT + i) * 32 + pix32(r);
    const int ii = m / THW, rem = m - ii * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    if constexpr (SWZ) hb[i] = ii * p.HROWS + 2 * ly * p.HWP + lx;  // tap (ky, kx) adds ky*HWP + (kx>>1) + (kx&1)*HWH
    else hb[i] = ii * p.HROWS + (SS * ly - LO) * p.HW + SS * lx - LO;
  }

  int ntaps;
  if (MODE == MODE_UP) ntaps = (py ? 3 : 2) * (px ? 3 : 2);
  else ntaps = 25;
  if (p.dbg & 8) ntaps = 1;  // ablation: one tap per chunk -> fixed per-block cost (timing only)
  // ksplit == 2 (exact-f32 DOWN on the 8 x 8 maps): this block contracts chunks [c_lo, c_lo + nchunk) only
  const int nchunk_all = p.Cs / KC;
  const int c_lo = p.ksplit > 1 ? (int)blockIdx.z * nchunk_all / p.ksplit : 0;
  const int nchunk = p.ksplit > 1 ? ((int)blockIdx.z + 1) * nchunk_all / p.ksplit - c_lo : nchunk_all;

  f32x16 acc[MT][NT];

  const int arow = tid / LPR, alc = tid % LPR;
  constexpr int NB = NT * KCH;  // B fragments (1 KiB each, 16 B per lane) of one tap for this wave's NT column tiles
  // NSETS register sets of B fragments = the prefetch distance in taps; ~96 VGPRs of fragments in flight whatever the
  // tap's size, so that the distance covers the L2 latency also for the short taps of KCH == 2 (8 MFMAs per tap)
#ifdef LG_EXP_NSETS
  constexpr int NSETS = (W3 && MODE == MODE_DOWN) ? LG_EXP_NSETS : ((W3 && MODE == MODE_UP) ? 2 : ((MT * NT == 4 && NB == 4) ? 6 : 3));
#else
  constexpr int NSETS = (W3 && MODE == MODE_UP) ? 2 : ((MT * NT == 4 && NB == 4) ? 6 : 3);  // deeper only where registers allow
#endif
  u32x4 fb[NSETS][NB];

  int nit = nchunk * ntaps;  // flat (chunk, tap) iteration space
  const int KB = p.Cs * ESZ / 32, N32 = p.Npad >> 5;
  const int nt0 = (n0 >> 5) + wn * NT;

  // B operand straight from the fragment-ordered pack: block (tap, n32, kb), lane-contiguous 16 B -> one coalesced
  // 1-KiB global load per MFMA operand, no LDS round trip and no barrier in the tap loop.
  auto load_frags = [&](u32x4 (&fb)[NB], int it) {
    const int cc = it / ntaps, t = it - cc * ntaps;
    int dy, dx, widx;
    tap_info(MODE, cls, t, dy, dx, widx);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < KCH; ++q) {
        const char* g = p.wp + ((((long long)widx * N32 + nt0 + j) * KB + (c_lo + cc) * KCH + q) * 64 + lane) * 16;
        fb[j * KCH + q] = *reinterpret_cast<const u32x4*>(g);
      }
  };
  auto stage_halo = [&](int c0, char* sH) {  // synchronous: up to SU halo rows in flight per thread (one L2/HBM latency
    constexpr int SU = 12;                   // per SU*RPP rows instead of one per 4*RPP)
    for (int hr0 = 0; hr0 < p.nrows; hr0 += SU * RPP) {
      if constexpr (SRC16) {  // bf16 mirror: 16 B = 8 channels, straight into the LDS image (no conversion); RES: c0 == 0
        u32x4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int hr = hr0 + u * RPP + arow;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          if (hr < p.nrows) {
            const int o = s_hoff[hr];
            if (o >= 0) v[u] = *reinterpret_cast<const u32x4*>(p.src16 + (long long)o * p.Cs + c0 + alc * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int hr = hr0 + u * RPP + arow;
          if (hr < p.nrows) *reinterpret_cast<u32x4*>(sH + hr * ROWB + ((SWZ ? (alc ^ ((hr >> 2) & 3)) : alc) << 4)) = v[u];
        }
      } else {
        f32x4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int hr = hr0 + u * RPP + arow;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (hr < p.nrows) {
            const int o = s_hoff[hr];
            if (o >= 0) v[u] = *reinterpret_cast<const f32x4*>(p.src + (long long)o * p.Cs + c0 + alc * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
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
    }
  };
  auto compute = [&](int it, const u32x4 (&fb)[NB], const char* sH) {
    const int cc = it / ntaps, t = it - cc * ntaps;
    int dy, dx, widx;
    tap_info(MODE, cls, t, dy, dx, widx);
    // SWZ (DOWN): stride-2 columns are consecutive rows of the de-interleaved image -> a ds_read_b128 lane group (16
    // consecutive pixels of a tile row) reads 16 consecutive 64-B rows = 16 distinct slots with the (row>>2)&3 XOR
    const int toff = SWZ ? (dy + 1) * p.HWP + ((dx + 1) >> 1) + ((dx + 1) & 1) * p.HWH : dy * p.HW + dx;
    if constexpr (RES) sH += cc * (KCH * 32);  // this chunk's channels inside the resident row
#pragma unroll
    for (int q = 0; q < KCH; ++q) {
      if constexpr (ESZ == 4) {
        f32x4 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(sH + (hb[i] + toff) * ROWB + h * 16 + q * 32);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], __builtin_bit_cast(f32x4, fb[j * KCH + q])[e],
                                                               acc[i][j], 0, 0, 0);
      } else {
        bf16x8 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = hb[i] + toff;
          if constexpr (SWZ) a[i] = *reinterpret_cast<const bf16x8*>(sH + row * ROWB + (((2 * q + h) ^ ((row >> 2) & 3)) << 4));
          else a[i] = *reinterpret_cast<const bf16x8*>(sH + row * ROWB + h * 16 + q * 32);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], __builtin_bit_cast(bf16x8, fb[j * KCH + q]),
                                                                acc[i][j], 0, 0, 0);
      }
    }
  };
  // Interleaved halo prefetch (DBUF): while the taps of chunk c run out of halo buffer `hcur`, the halo of chunk c+1
  // is fetched in `upt` row-units per tap (registers for one tap, then into the other buffer) -> no exposed staging.
  constexpr int UMAX = 4;
  const int NU = (p.nrows + RPP - 1) / RPP;                                   // row units (one float4 per thread each)
  const int upt = ntaps > 1 ? (NU + ntaps - 2) / (ntaps - 1) : NU;            // issued during taps 0 .. ntaps-2
  f32x4 hreg[UMAX];
  auto halo_issue = [&](int c0, int u0) {
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      if (u < upt) {
        const int hr = (u0 + u) * RPP + arow;
        hreg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hr < p.nrows) {
          const int o = s_hoff[hr];
          if (o >= 0) hreg[u] = *reinterpret_cast<const f32x4*>(p.src + (long long)o * p.Cs + c0 + alc * 4);
        }
      }
    }
  };
  auto halo_commit = [&](char* dst, int u0) {
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      if (u < upt) {
        const int hr = (u0 + u) * RPP + arow;
        if (hr < p.nrows) {
          if constexpr (ESZ == 4) {
            *reinterpret_cast<f32x4*>(dst + hr * ROWB + alc * 16) = hreg[u];
          } else {
            bf16x4 w;
            w[0] = (__bf16)hreg[u][0]; w[1] = (__bf16)hreg[u][1]; w[2] = (__bf16)hreg[u][2]; w[3] = (__bf16)hreg[u][3];
            *reinterpret_cast<bf16x4*>(dst + hr * ROWB + alc * 8) = w;
          }
        }
      }
    }
  };

  // one tap: MFMAs out of `cur`, then refill `cur` with the fragments of tap it+NSETS (its MFMAs have been issued, in
  // order, so the registers are free).  Chunk boundary: every wave must be done with the old halo (barrier).
  int hsel = 0;
  auto iteration = [&](int it, u32x4 (&cur)[NB]) {
    const int cc = it / ntaps, t = it - cc * ntaps;
    char* hcur = sH + hsel * HBYTES;
    if constexpr (DBUF) {
      if (cc + 1 < nchunk && !(p.dbg & 4)) {
        char* hnext = sH + (hsel ^ 1) * HBYTES;
        if (t > 0) halo_commit(hnext, (t - 1) * upt);
        if (t + 1 < ntaps || ntaps == 1) halo_issue((c_lo + cc + 1) * KC, t * upt);
        if (ntaps == 1) halo_commit(hnext, 0);
      }
    }
    if (!(p.dbg & 1)) compute(it, cur, hcur);
    if (it + NSETS < nit && !(p.dbg & 2)) load_frags(cur, it + NSETS);
    if (!RES && t + 1 == ntaps && it + 1 < nit && !(p.dbg & 4)) {
      __syncthreads();
      if constexpr (DBUF) {
        hsel ^= 1;
      } else {
        stage_halo((c_lo + cc + 1) * KC, sH);
        __syncthreads();
      }
    }
  };

  __syncthreads();  // tables visible
  if constexpr (RES) {
    stage_halo(0, sH);
    __syncthreads();
  }
  double* sred = reinterpret_cast<double*>(RES ? s_hoff : s_out);  // reduction scratch of the moments epilogue (16-B aligned)
#pragma unroll 1
  for (int pass = 0; pass < NCLS; ++pass) {
    if constexpr (RES) {
      cls = 3 - pass; py = cls >> 1; px = cls & 1;
      ntaps = (py ? 3 : 2) * (px ? 3 : 2);
      if (p.dbg & 8) ntaps = 1;
      nit = nchunk * ntaps;
    }
    const int* so = s_out + (RES ? cls * BM : 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int u = 0; u < NSETS; ++u)
      if (u < nit) load_frags(fb[u], u);
    if constexpr (!RES) {
      stage_halo(c_lo * KC, sH);
      __syncthreads();
    }
    for (int it = 0; it < nit; it += NSETS) {
#pragma unroll
      for (int u = 0; u < NSETS; ++u)
        if (it + u < nit) iteration(it + u, fb[u]);
    }

    // ---- epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----------
    // TEPI: the block's waves own column slices of the same pixel rows, so straight from the accumulators a pixel row
    // (N x 4 B) would leave as WAVES_N separate 128-B pieces at different times.  The finished tile goes through the
    // (now dead) halo region of LDS instead, epi_rows rows at a time, and leaves as whole rows, 16 B per lane.
    constexpr bool TEPI = WAVES_N > 1 && !DBUF;  // RES: the halo stays live across the classes -> its own C region behind it
    bool tepi_done = false;
    if constexpr (TEPI) {
      if (p.epi_rows > 0 && p.ksplit <= 1) {
        tepi_done = true;
        constexpr int CP = BN + 4;  // row pitch (floats)
        float* C = reinterpret_cast<float*>(RES ? sH + (p.nrows * ROWB + 15) / 16 * 16 : sH);
        const int RP = p.epi_rows;  // 32 or 64
        for (int pass = 0; pass < BM / RP; ++pass) {
          __syncthreads();  // halo (first pass) / previous pass fully consumed
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int fr0 = (wm * MT + i) * 32;
            if (fr0 / RP == pass) {
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                const int cl = (wn * NT + j) * 32 + r, col = n0 + cl;
                const float bv = (col < p.N && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  float v = acc[i][j][e] + bv;
                  if (p.act == 1) v = tanhf(v);
                  C[(fr0 % RP + (e & 3) + 8 * (e >> 2) + 4 * h) * CP + cl] = v;
                }
              }
            }
          }
          __syncthreads();
          for (int idx = tid; idx < RP * (BN / 4); idx += 256) {
            const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            const int o = so[pass * RP + row], col = n0 + c4 * 4;
            if (o >= 0 && col < p.N && !(p.dbg & 16)) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(C + row * CP + c4 * 4);
              if (p.out16) {
                bf16x4 w;
                w[0] = (__bf16)v[0]; w[1] = (__bf16)v[1]; w[2] = (__bf16)v[2]; w[3] = (__bf16)v[3];
                *reinterpret_cast<bf16x4*>(p.out16 + (long long)o * p.N + col) = w;
              } else {
                *reinterpret_cast<f32x4*>(p.out + (long long)o * p.N + col) = v;
              }
            }
          }
        }
      }
    }
    if constexpr (sizeof(T) == 4 && !RES && !DBUF) {
      // ksplit: the two halves of the contraction meet in the output, which the launcher zeroed: 0 + a + b and 0 + b + a are the same
      // float (IEEE addition commutes; with THREE summands it would not), so the result does not depend on which block comes first.
      // Bias rides with half 0; no activation, no fused moments (launch() declines both).
      if (p.ksplit > 1) {
        tepi_done = true;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = n0 + (wn * NT + j) * 32 + r;
          const bool cok = col < p.N;
          const float bv = (cok && p.bias && blockIdx.z == 0) ? p.bias[col] : 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
              const int o = so[row];
              if (cok && o >= 0 && !(p.dbg & 16)) unsafeAtomicAdd(p.out + (long long)o * p.N + col, acc[i][j][e] + bv);
            }
          }
        }
      }
    }
    if (!tepi_done)
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
          const int o = so[row];
          if (cok && o >= 0 && !(p.dbg & 16)) {
            float v = acc[i][j][e] + bv;
            if (p.act == 1) v = tanhf(v);
            if (p.out16) p.out16[(long long)o * p.N + col] = (__bf16)v;
            else p.out[(long long)o * p.N + col] = v;
          }
        }
      }
    }

    // ---- fused InstanceNormalization moments of this block's output tile (one sample per block: NI == 1) ----------
    // {count, mean, M2 about the block mean}, merged per sample with Chan's formula by stats_final_kernel (norm.hip):
    // the separate pass that re-read the whole conv output for its moments is gone.
    if (p.spart && !(p.dbg & 32) && p.NI == 2) {
      // two samples per block (8 x 8 maps): 32-row fragment f = wm*MT + i belongs to sample f >> 1; one record per sample
      float s2[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wn * NT + j) * 32 + r;
        if (col < p.N) {
          const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s2[((wm * MT + i) >> 1) & 1] += acc[i][j][e] + bv;
        }
      }
      __syncthreads();
      const int ncol = min(BN, p.N - n0);
      const double cnt = 64.0 * (double)ncol;
      double red[2] = {(double)s2[0], (double)s2[1]};
      lg_block_sum_d<2>(red, sred);
      if (tid == 0) { sred[32] = red[0] / cnt; sred[33] = red[1] / cnt; }
      __syncthreads();
      const float mean0 = (float)sred[32], mean1 = (float)sred[33];
      float m2[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wn * NT + j) * 32 + r;
        if (col < p.N) {
          const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int sm = ((wm * MT + i) >> 1) & 1;
            const float mean = sm ? mean1 : mean0;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = (acc[i][j][e] + bv) - mean; m2[sm] += d * d; }
          }
        }
      }
      const double md0 = sred[32], md1 = sred[33];
      double red2[2] = {(double)m2[0], (double)m2[1]};
      lg_block_sum_d<2>(red2, sred);
      if (tid == 0) {
        const int part = cls * p.ntn + tile_n;  // one tile per image
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
          if (img0 + sm < p.B) {
            double* o = p.spart + ((long long)(img0 + sm) * p.nparts + part) * 3;
            const double meand = sm ? md1 : md0, df = (double)(sm ? mean1 : mean0) - meand;
            o[0] = cnt; o[1] = meand; o[2] = red2[sm] - cnt * df * df;
          }
        }
      }
      if constexpr (RES) __syncthreads();
    } else
    if (p.spart && !(p.dbg & 32)) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wn * NT + j) * 32 + r;
        if (col < p.N) {
          const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e] + bv;
        }
      }
      __syncthreads();  // the scratch (s_out, or s_hoff under RES) is dead from here on for its first use
      const int ncol = min(BN, p.N - n0);
      const double cnt = 128.0 * (double)ncol;
      double red[1] = {(double)s};
      lg_block_sum_d<1>(red, sred);
      if (tid == 0) sred[16] = red[0] / cnt;
      __syncthreads();
      const float mean = (float)sred[16];
      float m2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wn * NT + j) * 32 + r;
        if (col < p.N) {
          const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = (acc[i][j][e] + bv) - mean; m2 += d * d; }
        }
      }
      const double meand = sred[16];
      double red2[1] = {(double)m2};
      lg_block_sum_d<1>(red2, sred);
      if (tid == 0) {
        const int tin = tile_m - img0 * p.tpi;  // tile within the image
        const int part = (cls * p.tpi + tin) * p.ntn + tile_n;
        double* o = p.spart + ((long long)img0 * p.nparts + part) * 3;
        // M2 was taken about the float-rounded mean: shift it to the exact block mean (M2' = M2 - cnt*(mean_f - mean)^2)
        const double df = (double)mean - meand;
        o[0] = cnt; o[1] = meand; o[2] = red2[0] - cnt * df * df;
      }
      if constexpr (RES) __syncthreads();  // scratch reused by the next class
    }
  }
}

constexpr int LDS_BUDGET = 80 * 1024;  // two blocks per CU (160 KiB)

template <typename T, int MODE, int KCH, bool DBUF, bool SRC16, bool RES, int WAVES_M, int WAVES_N, int MT, int NT, bool W3 = false>
int launch(HaloParams p, hipStream_t st) {
  constexpr int BN = WAVES_N * NT * 32;
  if (W3 && MODE == MODE_DOWN) {  // de-interleaved halo pixel rows (even columns, then odd columns)
    p.HWH = (p.HW + 1) / 2; p.HWP = 2 * p.HWH;
    p.HROWS = p.HH * p.HWP; p.nrows = p.NI * p.HROWS;
  }
  const int ROWB = RES ? p.Cs * DT<T>::ESZ + 16 : ((W3 && MODE == MODE_DOWN) ? KCH * 32 : KCH * 32 + 16);
  size_t lds = (((RES ? 4 : 1) * 128 + p.nrows) * 4 + 15) / 16 * 16 + (size_t)p.nrows * ROWB * (DBUF ? 2 : 1);
  if (lds > (size_t)(RES ? p.res_budget : LDS_BUDGET)) return LG_ERR_UNSUPPORTED;
  const bool res_epi = RES && WAVES_N > 1 && p.N % 4 == 0 && (p.cfg & 128);  // transposed epilogue for RES too: 32 rows x (BN + 4) floats of extra LDS
  if (res_epi) lds = (lds + 15) / 16 * 16 + (size_t)32 * (BN + 4) * 4;
  if (DBUF) {  // the interleaved prefetch must fit its register window: ceil(NU/(ntaps-1)) <= 4 with the fewest taps
    constexpr int KC = KCH * 32 / DT<T>::ESZ, RPP = 256 / (SRC16 ? KC / 8 : KC / 4);
    const int NU = (p.nrows + RPP - 1) / RPP, min_taps = MODE == MODE_UP ? 4 : 25;
    if ((NU + min_taps - 2) / (min_taps - 1) > 4) return LG_ERR_UNSUPPORTED;
  }
  p.ntn = p.Npad / BN;
  {  // LDS-transposed epilogue: needs N % 4 == 0 and epi_rows x (BN + 4) floats inside the halo region
    const size_t halo_bytes = (size_t)p.nrows * ROWB;
    p.epi_rows = 0;
    if (WAVES_N > 1 && !DBUF && !RES && p.N % 4 == 0 && !(p.cfg & 64)) {
      if (halo_bytes >= (size_t)64 * (BN + 4) * 4) p.epi_rows = 64;
      else if (halo_bytes >= (size_t)32 * (BN + 4) * 4) p.epi_rows = 32;
    }
    if (res_epi) p.epi_rows = 32;
  }
  p.nparts = (MODE == MODE_UP ? 4 : 1) * p.tpi * p.ntn;
  if (p.nparts_host) *p.nparts_host = p.ksplit > 1 ? 0 : p.nparts;
  const int ntm = p.NI == 1 ? p.B * p.tpi : lg_cdiv(p.B, p.NI);
  dim3 grid(ntm * p.ntn, (MODE == MODE_UP && !RES) ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
  p.ntm = ntm;
  {
    static int mf = -1;
    if (mf < 0) { const char* e = getenv("LG_HALO_MFAST"); mf = e ? atoi(e) : 0; }   // 0 never, 1 where the weights exceed 2 MB, 2 always
    const size_t wbytes = (size_t)(MODE == MODE_UP ? 25 : 25) * p.Cs * p.Npad * DT<T>::ESZ;
    p.mfast = (p.mfast || mf == 2 || (mf == 1 && wbytes > (2u << 20))) && p.ntn > 1 ? 1 : 0;
  }
  {
    static int co = -1;
    if (co < 0) co = lg_env_flag("LG_NO_UP_CLSORDER") ? 0 : 1;
    p.clsorder = co && MODE == MODE_UP && !RES && (int)grid.x * 4 <= 2 * lg_device_cus();   // one round only: over several rounds the lightest class should come last (convT3 forward at B = 64, 2048 blocks: 269 us in the order 9, 6, 6, 4 taps, 281 with the 4-tap class third)
  }
  auto kern = conv_halo_kernel<T, MODE, KCH, DBUF, SRC16, RES, WAVES_M, WAVES_N, MT, NT, W3>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RES ? 160 * 1024 : LDS_BUDGET);
    attr_set = true;
  }
  if (p.dry) return LG_OK;
  if (p.ksplit > 1 && hipMemsetAsync(p.out, 0, (size_t)p.B * p.Ho * p.Wo * p.N * sizeof(float), st) != hipSuccess) {
    lg_set_error("lg_conv_halo: hipMemsetAsync of the split-contraction output failed");
    return LG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  lg_note_kernel(MODE == MODE_DOWN ? (sizeof(T) == 2 ? "conv_halo_kernel<bf16,DOWN>" : "conv_halo_kernel<f32,DOWN>")
                 : MODE == MODE_UP ? (RES ? (sizeof(T) == 2 ? "conv_halo_kernel<UP,resident>" : "conv_halo_kernel<f32,UP,resident>") : (sizeof(T) == 2 ? "conv_halo_kernel<bf16,UP,K-sliced>" : "conv_halo_kernel<f32,UP,K-sliced>"))
                 : "conv_halo_kernel<S1T>");
  return LG_OK;
}

template <typename T, int MODE, int KCH, bool DBUF, bool SRC16, bool RES = false>
int dispatch_bn2(const HaloParams& p, hipStream_t st) {
  // exact-f32 path, 8 x 8 maps (two samples per row tile): at B = 64 that is 32 row tiles x 3 column tiles of 128 = 96 blocks for 256
  // CUs (measured: conv4 forward takes the same 535 us at B = 64 and B = 128): 64-column tiles double the blocks.  An fp32 MFMA is 64
  // cycles for two operand registers, so the narrower tile costs little operand traffic per matrix cycle.  (The choice depends on the
  // map, not on the batch: a launch and its 32-image chunks must take the same tiling — the moment partials are compared bit for bit.)
  bool narrow = false;
  if constexpr (sizeof(T) == 4 && !DBUF && !RES) {
    static int off = -1;
    if (off < 0) off = lg_env_flag("LG_NO_F32_NARROW") ? 1 : 0;
    narrow = !off && p.NI > 1 && p.Npad % 64 == 0;
    // Round 5, the 8 x 8 maps at the C2 batches (scripts/probe/f32_ablate.sh, f32_tiling_ab.sh, f32_mfast_ab.sh).  These grids are
    // ONE round of at most 512 blocks, every block in the same phase at the same time, so what counts is how evenly that round covers the
    // 1024 SIMDs.  Measured per launch (us), conv4 forward (DOWN, 25 x 256 deep, N = 384) | convT1 forward (UP, 384 -> 256):
    //     B = 64 : 64-column tiles 349 (192 blocks) , split 32-column 308-327, split 64-column 285, + row-tile-fastest 269 | 64-col 220, 128-col 274
    //     B = 128: 64-column 482 (384 blocks), 128-column 507, split 32-col 482, split 64-col 520                          | 64-col 520, 128-col 380
    // -> DOWN: up to 256 blocks of 64 columns are split in two over the channel chunks (blockIdx.z; both halves add into the zeroed output —
    //    two summands commute, the result is deterministic; the moments come from the separate pass over the 6 MB map);
    //    UP: 128-column tiles once they fill the 512 slots (4 classes x row tiles x N / 128 >= 512), 64-column tiles below that.
    // The batch enters these two choices.  A tile's width does not change the conv result (an output element's contraction order is the
    // same in every tiling), only the grouping of the fused moment records (their last bits); the SPLIT changes the result itself by the
    // order of one addition per element.  Every OTHER choice here depends on the map only (tests compare a launch with its 32-image chunks
    // bit for bit; at 32 against 64 images both sit on the same side of both thresholds, at 2B conv4 forward does not and the test bounds
    // the difference instead: tests/test_launch_shapes_gpu.py, test_f32_8x8_level_tilings_that_depend_on_the_batch).
    const int ntm2 = lg_cdiv(p.B, p.NI > 0 ? p.NI : 1);
    if constexpr (MODE == MODE_DOWN) {
      static int ks = -1;
      if (ks < 0) ks = lg_env_flag("LG_NO_F32_KSPLIT") ? 0 : 1;
      constexpr int KC = KCH * 32 / 4;
      if (ks && narrow && p.act == 0 && !p.out16 && (p.Cs / KC) % 2 == 0 && (p.Cs / KC) >= 4 && ntm2 * (p.Npad / 64) <= 256) {
        HaloParams q = p;
        q.ksplit = 2; q.spart = nullptr; q.mfast = 1;
        return launch<T, MODE, KCH, DBUF, SRC16, RES, 2, 2, 2, 1>(q, st);
      }
    }
    if constexpr (MODE == MODE_UP) {
      static int wide = -1;
      if (wide < 0) wide = lg_env_flag("LG_NO_F32_UPWIDE") ? 0 : 1;
      if (wide && narrow && p.Npad % 128 == 0 && 4 * ntm2 * (p.Npad / 128) >= 512) narrow = false;
    }
  }
  if constexpr (!DBUF && !RES) if (!narrow) {  // tall wave tiles (128 rows x 64/32 cols): half the weight-fragment traffic per MFMA
    if (p.Npad % 256 == 0 && (p.cfg & 1)) return launch<T, MODE, KCH, DBUF, SRC16, RES, 1, 4, 4, 2>(p, st);
    if (p.Npad % 128 == 0 && (p.cfg & 2) && (!(MODE == MODE_DOWN && p.NI > 1) || (halo_w3_ok<T, MODE, KCH, SRC16, RES, 1, 4, 1>() && (p.cfg & 8)))) {
      if constexpr (halo_w3_ok<T, MODE, KCH, SRC16, RES, 1, 4, 1>()) {
        // three resident blocks per CU: measured better on the whole step than a rounds-of-the-grid heuristic (LG_CFG bit 2 = off)
        if (!(p.cfg & 4) && (MODE == MODE_DOWN || (p.cfg & 32))) return launch<T, MODE, KCH, DBUF, SRC16, RES, 1, 4, 4, 1, true>(p, st);
      }
      return launch<T, MODE, KCH, DBUF, SRC16, RES, 1, 4, 4, 1>(p, st);
    }
  }
  if (p.Npad % 128 == 0 && !narrow) return launch<T, MODE, KCH, DBUF, SRC16, RES, 2, 2, 2, 2>(p, st);
  if (p.Npad % 64 == 0) {   // (32-column tiles for the f32 8 x 8 level: conv4 forward 330 -> 305 us at B = 64, but the C2 step 9.08 -> 9.19 ms)
    if constexpr (!DBUF && halo_w3_ok<T, MODE, KCH, SRC16, RES, 2, 2, 1>()) {
      if (!(p.cfg & 16)) return launch<T, MODE, KCH, DBUF, SRC16, RES, 2, 2, 2, 1, true>(p, st);
    }
    return launch<T, MODE, KCH, DBUF, SRC16, RES, 2, 2, 2, 1>(p, st);
  }
  return launch<T, MODE, KCH, DBUF, SRC16, RES, 4, 1, 1, 1>(p, st);
}
template <typename T, int MODE, int KCH>
int dispatch_bn(const HaloParams& p, hipStream_t st) {
  int rc = LG_ERR_UNSUPPORTED;
  if constexpr (sizeof(T) == 2) {
    if (p.src16) {  // bf16 mirror of the source available
      if constexpr (MODE == MODE_UP && KCH == 4) {  // all channels resident, the four classes in one block
        if (p.res_budget > 0) {
          const int rc = dispatch_bn2<T, MODE, KCH, false, true, true>(p, st);
          if (rc != LG_ERR_UNSUPPORTED) return rc;
        }
      }
      return dispatch_bn2<T, MODE, KCH, false, true>(p, st);
    }
  }
  if constexpr (sizeof(T) == 4 && MODE == MODE_UP && KCH == 4) {
    // Round 5: the resident form for the exact-f32 N = 32 level (convT4 forward: 64 channels x 4 B = 49 KB of halo).  K-sliced, that layer is
    // 8192 blocks of 32 x 32-pixel-by-column wave tiles, each a prologue (tables, halo, first fragments) for 64 .. 288 MFMAs per wave; resident,
    // one block stages the tile once and runs the four classes' 800 MFMAs per wave behind it.  LG_NO_F32_RES=1 = before.
    static int res32 = -1;
    if (res32 < 0) res32 = lg_env_flag("LG_NO_F32_RES") ? 0 : 1;
    if (res32 && p.res_budget > 0 && p.Npad == 32 && p.NI == 1 && p.Cs % 4 == 0 && 256 % (p.Cs / 4) == 0) {
      rc = launch<T, MODE, KCH, false, false, true, 4, 1, 1, 1>(p, st);
      if (rc != LG_ERR_UNSUPPORTED) return rc;
    }
  }
  if ((p.dbg & 64) && p.Cs / (KCH * 32 / DT<T>::ESZ) > 1) rc = dispatch_bn2<T, MODE, KCH, true, false>(p, st);  // double-buffered halo: measured slower, opt-in
  if (rc == LG_ERR_UNSUPPORTED) rc = dispatch_bn2<T, MODE, KCH, false, false>(p, st);
  return rc;
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
// spart/spart_bytes/nparts_out (optional): if the tiling puts ONE sample per block, the kernel also writes per-block
// InstanceNorm moments and *nparts_out = partial records per sample (0 = not produced, run the separate stats pass).
extern "C" int lg_conv_halo_try(int mode, int dtype, const float* src, const void* src16, const void* wpack,
                                const float* bias, float* out, void* out16, int B, int Hm, int Wm, int Cs, int N, int act, void* spart,
                                size_t spart_bytes, int* nparts_out, void* stream) {
  if (nparts_out) *nparts_out = 0;
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
  p.src = src; p.wp = (const char*)wpack; p.bias = bias; p.out = out; p.out16 = (__bf16*)out16;
  p.dry = (!out && !out16) ? 1 : 0;
  p.src16 = dtype == LG_DT_BF16 ? (const __bf16*)src16 : nullptr;
  p.B = B; p.Cs = Cs; p.Hm = Hm; p.Wm = Wm; p.N = N; p.Npad = lg_npad(N); p.act = act;
  p.Hs = ss * Hm; p.Ws = ss * Wm;
  p.Ho = mode == MODE_UP ? 2 * Hm : Hm; p.Wo = mode == MODE_UP ? 2 * Wm : Wm;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("LG_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
    static int resb = -1;  // LG_RES_KB: LDS budget (KiB) of the resident-halo UP variant; 0 switches it off (A/B)
    if (resb < 0) { const char* e = getenv("LG_RES_KB"); resb = (e ? atoi(e) : 52) * 1024; }
    p.res_budget = resb;
    static int cfg = -1;
    if (cfg < 0) { const char* e = getenv("LG_CFG"); cfg = e ? atoi(e) : 170; }  // measured: bits 1 (128x32 wave tiles), 3 (also for small maps with W3), 5 (W3 for UP), 7 (RES: transposed epilogue) on
    p.cfg = cfg;
  }
  int nparts = 0;  // set by launch<> to the partial records per sample of the tiling it chose
  if (spart && nparts_out && (p.NI == 1 || (p.NI == 2 && p.TH * p.TW == 64)) && act == 0) {
    const int worst = (mode == MODE_UP ? 4 : 1) * p.tpi * (p.Npad / 32);  // narrowest column tile
    if ((size_t)B * worst * 3 * sizeof(double) <= spart_bytes) { p.spart = (double*)spart; p.nparts_host = &nparts; }
  }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (mode == MODE_DOWN) rc = dispatch<MODE_DOWN>(p, dtype, st);
  else if (mode == MODE_UP) rc = dispatch<MODE_UP>(p, dtype, st);
  else rc = dispatch<MODE_S1T>(p, dtype, st);
  if (rc != LG_OK) return rc;
  if (p.dry) return LG_OK;
  LG_CHECK_LAUNCH("lg_conv_halo");
  if (nparts_out) *nparts_out = nparts;
  return LG_OK;
}

// Bytes of the moment-partials buffer (`spart`) a fused-statistics conv of this shape may need: B samples x the partial
// records of the finest tiling (one per 128-pixel tile, 32-column slice and — UP — parity class) x {count, mean, M2}
// doubles.  mode 0 = conv ("down": Hm x Wm is the OUTPUT map), 1 = transposed conv ("up": Hm x Wm is the INPUT map).
extern "C" size_t lg_conv_stats_workspace_bytes(int mode, int B, int Hm, int Wm, int N) {
  if (B <= 0 || Hm <= 0 || Wm <= 0 || N <= 0) return 0;
  const long long tiles = ((long long)Hm * Wm + 127) / 128;
  const long long worst = (mode == MODE_UP ? 4 : 1) * tiles * (lg_npad(N) / 32);
  return (size_t)B * (size_t)worst * 3 * sizeof(double) + 256;
}

// 1 if the halo kernel covers this shape (then the bf16 mirror alone is enough as its source), else 0
extern "C" int lg_conv_halo_supported(int mode, int dtype, int B, int Hm, int Wm, int Cs, int N) {
  static const int dummy = 0;
  return lg_conv_halo_try(mode, dtype, nullptr, dtype == LG_DT_BF16 ? &dummy : nullptr, &dummy, nullptr, nullptr, nullptr, B,
                          Hm, Wm, Cs, N, 0, nullptr, 0, nullptr, nullptr) == LG_OK ? 1 : 0;
}
