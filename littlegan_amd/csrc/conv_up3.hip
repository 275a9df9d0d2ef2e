// Software-pipelined stride-2 "up" contraction for gfx950 (bf16 operands, fp32 accumulation), small-N layers:
//   out[n][2y+py][2x+px][co] = bias[co] + sum_{taps (ky,kx) of parity class (py,px)} sum_ci src[n][y+dy][x+dx][ci] W[ky][kx][co][ci]
// = Conv2DTranspose(f, 5, 2, same) forward (model.py:39-40) and the data gradient of Conv2D(f, 5, 2, same) (model.py:15),
// for the layers with few output channels and a source that fits LDS whole:
//   CS = 128, N = 64 : convT3 forward, conv2 data gradient        CS = 64, N = 32 : convT4 forward
// (the generator's 32..64-channel levels: the bulk of the transposed-conv stack's time in conv_halo.hip, where their
//  blocks of 128 output-class pixels re-fetch the whole weight tensor per 32 / 64 pixel rows and spend most of their life
//  in prologue and epilogue).
//
// One workgroup of 8 waves per CU walks a list of 8 x 16 source-pixel tiles (NT tiles at a time):
//   * the source halo of a tile (10 x 18 pixels, ALL channels) is resident in LDS; the next tile's halo is in flight
//     global -> registers while this one computes and is written behind the last MFMA: two barriers per tile, none inside;
//   * the FOUR parity classes of a tile run CONCURRENTLY on different waves — a wave owns (column half | tile, class) and
//     all 128 pixels (4 MFMA row tiles), so every weight fragment is fetched by exactly one wave and used for 4 MFMAs;
//     the classes have 9 / 6 / 6 / 4 taps, and waves w and w + 4 share a SIMD: class (3 | 0) and (1 | 2) pairs balance it;
//   * weight fragments (fragment order of pack.hip) stream global -> registers through an 8-deep ring; they do not
//     depend on the tile, so the stream never drains; every LDS address of a tap is the lane's base + a compile-time offset;
//   * the product is formed transposed (channels on accumulator rows, pixels on lanes) and leaves the registers through a
//     WHOLE-OUTPUT-TILE staging area in LDS (16 x 32 output pixels x N channels): the four classes interleave there and
//     the tile reaches HBM as contiguous rows of 32 pixels x N channels, 16 B per lane;
//   * the InstanceNormalization moments of the tile come out of the same registers (one pass about a shift);
//   * (round 3, forward forms) the staged tile's rows do not get a phase of their own: they leave INSIDE the next tile's MFMA
//     stream, one ds_read_b128 + one buffer store per second fragment (see DEFERRED ROW SWEEP below).  Measured A/B on one box
//     (LG_U3_NO_DEFER=1 variant): convT3 forward 113.4 -> 111.3 us, conv2 data gradient 124 -> 120.5 us, convT4 forward within noise
//     (172-178 both ways): the row phase was the small part of the non-MFMA share (staging + the two barriers stay).
//   * (round 4, measured and removed) convT4 forward WITHOUT the staging area: double-buffered halo, one barrier per step, every wave
//     storing its accumulators as 8-byte buffer stores (two lanes = 16 contiguous bytes of a pixel's 64-byte row): correct, and 192 us
//     against 137 — sixteen scattered 16-byte-segment stores per wave and step are what the staging area exists to avoid.  Also
//     without effect on that layer: the halo requested / committed by the three waves that leave their class loop early (153.5 vs
//     153.6 us), a start delay of the second workgroup of a CU (141.6 / 143.2 / 143.0 / 145.1 us for 0 / 5 / 9 / 13 k cycles).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "lg_common.h"

#ifndef LG_U3_NO_DEFER
#define LG_U3_NO_DEFER 0
#endif

namespace {

template <int... I, class Fn>
__device__ __forceinline__ void lg_static_for_(std::integer_sequence<int, I...>, Fn&& fn) { (fn(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void lg_static_for(Fn&& fn) { lg_static_for_(std::make_integer_sequence<int, N>{}, fn); }

constexpr int TH = 8, TW = 16, HHT = TH + 2, HWT = TW + 2, NPX = HHT * HWT;  // 10 x 18 = 180 halo pixels
constexpr int RING = 8;
#ifndef LG_U3_SCHED
#define LG_U3_SCHED 1   // 0: fragment step in three pinned groups; 1: interleaved by sched_group_barrier (MFMA, LDS read, ...)
#endif

// NTT = tiles per workgroup step.  Waves = NTT * WN * 4 classes: 8 (one workgroup per CU) or, for N = 32 with ONE tile per step,
// 4 (TWO independent workgroups per CU, 58 KB of LDS each: one's staging / row-store / barrier phases fall into the other's
// MFMA phase — r3 stamps of the 8-wave form at N = 32: 49 % of a step is not MFMA time and nothing overlaps it).
template <int CS, int N, int NTT = 2 / (N / 32)> struct Cfg {
  static constexpr int WN = N / 32;               // column waves per tile (2 | 1)
  static constexpr int NT = NTT;                  // tiles per workgroup step
  static constexpr int NWAVES = NT * WN * 4, THREADS = 64 * NWAVES;
  // 4-wave form: a wave's class ROTATES from step to step (order 3, 1, 0, 2): the waves of a workgroup sit on different SIMDs
  // and the classes have 9 / 6 / 6 / 4 taps — fixed roles would leave one SIMD with 2.25 x the matrix work of another
  static constexpr bool ROT = NWAVES == 4;
  static constexpr int PITCH = CS * 2 + 16;       // halo pixel pitch (bytes): consecutive pixels shift by one 16-B slot
  static constexpr int HB = NPX * PITCH;          // halo bytes per tile
  static constexpr int OPX = 4 * TH * TW;         // 512 output pixels per tile
  static constexpr int CROW = N * 2;              // output pixel row in the staging area (bytes): 128 | 64
  static constexpr int CB = OPX * CROW;           // staging bytes per tile
  static constexpr int KB = CS / 16;              // k-steps
  static constexpr int C_OFF = NT * HB;
  static constexpr int SRED_OFF = C_OFF + NT * CB;
  static constexpr int SBIAS_OFF = SRED_OFF + THREADS * 8;    // per-thread {sum d, sum d^2} floats of the current step
  static constexpr int LDS = SBIAS_OFF + N * 4;
  static constexpr int PIECES = NT * NPX * (CS * 2 / 16);       // 16-B halo pieces per step
  static constexpr int PPT = (PIECES + THREADS - 1) / THREADS;
};
constexpr int rot_class(int i) { return (i & 3) == 0 ? 3 : (i & 3) == 1 ? 1 : (i & 3) == 2 ? 0 : 2; }

struct U3Params {
  const __bf16* src;   // [B][Hs][Ws][CS]
  const char* wp;      // "up" pack [25][N/32][CS/16][64][16 B]
  const float* bias;   // [N] or null
  __bf16* out;         // [B][2Hs][2Ws][N]
  double* spart;       // [B][tpi][3] or null
  int B, Hs, Ws, tpi_x, tpi, nitems;
  LgNormFuse nf;       // FUSE instantiation: norm-backward sums of the produced gradient (lg_common.h)
  unsigned long long* stamps;  // diagnostic build (LG_U3_STAMPS): [block][8 waves][32]
  int rotoff;          // 4-wave form: class-rotation offset of the SECOND workgroup of a CU (see ROT below)
};

__device__ __forceinline__ int pix32(int r) {
  const int q = r >> 2, lo = r & 3;
  const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;
  const int rank = odd ? ((q == 1) ? 0 : (q == 2) ? 1 : (q == 4) ? 2 : 3) : ((q == 0) ? 0 : (q == 3) ? 1 : (q == 5) ? 2 : 3);
  return odd * 16 + rank * 4 + lo;
}

// class = 2*py + px; taps a < nky(py), b < nkx(px):  ky = py ? 2a : 2a+1, dy = (py + 1 - ky) / 2  (same in x)
constexpr int nk(int p) { return p ? 3 : 2; }
constexpr int ntaps_of(int cls) { return nk(cls >> 1) * nk(cls & 1); }
constexpr int tap_k(int p, int a) { return p ? 2 * a : 2 * a + 1; }
constexpr int tap_d(int p, int a) { return (p + 1 - tap_k(p, a)) / 2; }

template <int CS, int N, bool STATS, bool FUSE = false, int NTT = 2 / (N / 32)>
__global__ __launch_bounds__((Cfg<CS, N, NTT>::THREADS), 2) void conv_up3_kernel(const U3Params p) {
  static_assert(!(STATS && FUSE), "forward moments and backward sums are never needed together");
  using C = Cfg<CS, N, NTT>;
  constexpr int NTH = C::THREADS, NWV = C::NWAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x2* sstat = reinterpret_cast<f32x2*>(smem + C::SRED_OFF);
  float* sbias = reinterpret_cast<float*>(smem + C::SBIAS_OFF);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave roles.  N = 64: (column half, class slot); N = 32: (tile, class slot).  Waves w and w+4 share a SIMD: slots
  // pair the 9-tap class with the 4-tap one and the two 6-tap ones.
  const int wn = C::WN == 2 ? (wid & 1) : 0;
  const int tsel = C::WN == 2 ? 0 : (wid >> 2);                      // which of the NT tiles
  const int slot = C::WN == 2 ? (wid >> 1) : (wid & 3);
  const int cls_fixed = C::WN == 2 ? (slot == 0 ? 3 : slot == 1 ? 1 : slot == 2 ? 0 : 2)
                                   : (tsel == 0 ? (slot == 0 ? 3 : slot == 1 ? 1 : slot == 2 ? 2 : 0) : (slot == 0 ? 0 : slot == 1 ? 2 : slot == 2 ? 1 : 3));
  const int G = gridDim.x;
  // ROT: class of step s = rot_class(wid + s + rsh).  The two workgroups of a CU put one wave each on every SIMD and run the same
  // program from the same start: in phase, a SIMD holds two 9-tap waves while another holds two 4-tap ones (18 : 8 taps).  The
  // second workgroup of a CU (the upper half of the grid) can start its rotation rsh places on (LG_U3_ROTOFF; 2: its wave w runs the class
  // that COMPLEMENTS the first workgroup's — (3, 0), (1, 2), (0, 3), (2, 1): 13 / 12 taps per SIMD and step).  Measured in round 5
  // (scripts/probe/u3_rotoff_ab.sh, convT4 forward at B = 256): 145.3 / 143.1 us in phase, 141.4 (offset 1), 142.7 / 143.8 (offset 2) —
  // the matrix pipe's balance is not what bounds this layer; default 0.
  const int rsh = (C::ROT && (int)blockIdx.x >= (G + 1) / 2) ? p.rotoff : 0;
  int cls = C::ROT ? rot_class(wid + rsh) : cls_fixed;
  const int lb = lg_xcd_remap(blockIdx.x, G);
  const int nsteps_all = (p.nitems + C::NT - 1) / C::NT;       // steps (NT tiles each) over the whole problem
  const int nmine = (nsteps_all - lb + G - 1) / G;

  if (tid < N) sbias[tid] = p.bias ? p.bias[tid] : 0.f;

  // ---- halo pieces of this thread (same positions every step) -----------------------------------------------------------
  constexpr int PPR = CS * 2 / 16;  // 16-B pieces per halo pixel
  int pl[C::PPT], pyx[C::PPT];
#pragma unroll
  for (int u = 0; u < C::PPT; ++u) {
    const int q = tid + u * NTH;
    pl[u] = -1; pyx[u] = 0;
    if (q < C::PIECES) {
      const int t = q / (NPX * PPR), rem = q - t * (NPX * PPR);
      const int px = rem / PPR, pc = rem - px * PPR;
      const int hy = px / HWT, hx = px - hy * HWT;
      pl[u] = t * C::HB + px * C::PITCH + pc * 16;
      pyx[u] = (t << 20) | (pc << 12) | (hy << 6) | hx;
    }
  }
  auto tile_of = [&](int step, int t, int& n, int& y0, int& x0) {  // false: no such tile (odd tail)
    const int item = (lb + step * G) * C::NT + t;
    if (item >= p.nitems) { n = 0; y0 = 0; x0 = 0; return false; }
    n = item / p.tpi;
    const int tt = item - n * p.tpi;
    y0 = (tt / p.tpi_x) * TH; x0 = (tt % p.tpi_x) * TW;
    return true;
  };
  // Halo pieces come through RAW BUFFER loads from a descriptor over the step's sample(s): a piece outside the image (SAME
  // padding), of a tile that does not exist (odd tail) or beyond the thread's share gets an out-of-range offset and reads as
  // zeros — NO branch around any load, so hipcc keeps an exact count of the loads in flight (with `if (inside) v = load` it
  // loses the count at the join and drains the weight ring with s_waitcnt vmcnt(0) at every commit; see conv_down3.hip).
  constexpr unsigned OOB = 0x80000000u;   // >= num_records for every supported shape (checked on the host)
  const int sample_elems = p.Hs * p.Ws * CS;
  auto issue = [&](int step, u32x4 (&v)[C::PPT]) {
    int n[C::NT], y0[C::NT], x0[C::NT];
    bool ok[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) ok[t] = tile_of(step, t, n[t], y0[t], x0[t]);
    // the NT tiles of a step are consecutive items: the second one lies in the same sample or in the next
    // (readfirstlane: hipcc forms this select on the VALU, can then no longer prove the descriptor wave-uniform and wraps EVERY halo
    //  load in a waterfall loop — four v_readfirstlane, two compares, exec juggling, one load at a time; found in the .s in round 4,
    //  it had been there since round 2: 54 -> 8 v_readfirstlane per kernel.  cdna_hip_programming.md T20.)
    const int nrec = __builtin_amdgcn_readfirstlane((p.B - n[0] >= 2 ? 2 : 1) * sample_elems * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.src + (long long)n[0] * sample_elems), 0, nrec, 0x00027000);
#pragma unroll
    for (int u = 0; u < C::PPT; ++u) {
      const bool t1 = C::NT == 2 && (pyx[u] >> 20) != 0;  // selects, not runtime-indexed arrays (those go to scratch)
      const int ty = t1 ? y0[C::NT - 1] : y0[0], tx = t1 ? x0[C::NT - 1] : x0[0], dn = t1 ? n[C::NT - 1] - n[0] : 0;
      const bool tok = t1 ? ok[C::NT - 1] : ok[0];
      const int sy = ty - 1 + ((pyx[u] >> 6) & 63), sx = tx - 1 + (pyx[u] & 63);
      const bool in = pl[u] >= 0 && tok && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      const unsigned off = in ? (unsigned)((((dn * p.Hs + sy) * p.Ws + sx) * CS + ((pyx[u] >> 12) & 255) * 8) * 2) : OOB;
      v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
  auto commit = [&](const u32x4 (&v)[C::PPT]) {
#pragma unroll
    for (int u = 0; u < C::PPT; ++u)
      if (pl[u] >= 0) *reinterpret_cast<u32x4*>(smem + pl[u]) = v[u];
  };

  // ---- per-lane A bases (pixel m = i*32 + pix32(r) at halo position (ly, lx); taps add (dy+1, dx+1)) -----------------------
  int abase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = i * 32 + pix32(r);
    abase[i] = tsel * C::HB + ((m >> 4) * HWT + (m & 15)) * C::PITCH + h * 16;
  }
  const unsigned lane16 = lane * 16;
  const char* wwave = p.wp + (long long)wn * C::KB * 1024;      // this wave's column tile
  constexpr unsigned WTAP = (unsigned)(N / 32) * C::KB * 1024u;  // bytes from one tap's fragments to the next tap's

  u32x4 bf[RING];
  u32x4 hv[C::PPT];
  f32x16 acc[4];

  // fragment f of class CLS: tap = f / KB, k-step = f % KB
  // (an opaque zero added per tile keeps the 72 fragment addresses out of loop-invariant VGPRs; the POINTER itself must
  //  not pass through the asm: it would come back as a generic pointer and every fragment load would become a flat_load,
  //  which counts on vmcnt AND lgkmcnt and forces "s_waitcnt vmcnt(0) lgkmcnt(0)" before each MFMA group)
  unsigned long long wzero = 0;
  auto frag_ptr = [&](auto cls_c, int f) {
    constexpr int CLS = decltype(cls_c)::value;
    constexpr int PY = CLS >> 1, PX = CLS & 1, NKX = nk(PX);
    const int t = f / C::KB, kb = f - t * C::KB;
    const int a = t / NKX, b = t - a * NKX;
    const int widx = tap_k(PY, a) * 5 + tap_k(PX, b);
    return (wwave + (wzero + ((unsigned long long)widx * WTAP + kb * 1024))) + lane16;  // scalar base + constant, 32-bit lane offset
  };

  // ---- prologue ----------------------------------------------------------------------------------------------------------
  issue(0, hv);
  auto prime = [&](auto cls_c) {
#pragma unroll
    for (int f = 0; f < RING; ++f) bf[f] = *reinterpret_cast<const u32x4*>(frag_ptr(cls_c, f));
  };
  if (cls == 3) prime(std::integral_constant<int, 3>{});
  else if (cls == 2) prime(std::integral_constant<int, 2>{});
  else if (cls == 1) prime(std::integral_constant<int, 1>{});
  else prime(std::integral_constant<int, 0>{});
  commit(hv);
  __syncthreads();

  // ---- DEFERRED ROW SWEEP (forward forms, !FUSE): the staged output tile of step s leaves for HBM INSIDE the MFMA stream of step
  // s + 1 — per thread 8 pieces, one ds_read_b128 and one buffer store each, spread over the first fragments of the class loop —
  // instead of in a phase of its own between two barriers (r3 stamps: staging / row stores / barriers are 36 % (N = 64) to 49 %
  // (N = 32) of a step and nothing overlapped them).  Order per step: class loop (+ rows of the previous tile) | next halo requested
  // | BARRIER (halo and staging area free) | stage | halo commit | BARRIER | moments of this tile.  The stores go through a buffer
  // descriptor over the whole output: a tile that does not exist (step 0's "previous" tile, the odd tail) gets an out-of-range
  // offset and its stores are dropped — no branch in the loop.
  constexpr bool DEFER = !FUSE && !LG_U3_NO_DEFER;   // (LG_U3_NO_DEFER=1: the round-2 order, rows in a phase of their own — A/B builds)
  constexpr int PPO_ = C::CROW / 16, TOT_ = C::NT * C::OPX * PPO_, NPC = TOT_ / NTH;   // pieces per thread and step (8)
  // Piece q8 of thread tid is output pixel o = tid / PPO + q8 * (NTH / PPO) of tile (q8 * (NTH / PPO)) / OPX, 16-byte column tid % PPO:
  // ONE LDS address and ONE global offset per thread (the swizzle term does not depend on q8), everything else is an immediate or
  // a scalar — the tile's byte offset and the piece's row offset ride in the store's scalar offset.
  constexpr int OPQ = NTH / PPO_;   // pixels between a thread's consecutive pieces
  static_assert(C::OPX % OPQ == 0 && OPQ % 32 == 0 && (OPQ / 2) % PPO_ == 0, "piece map");
  unsigned pbyte[C::NT];   // previous step's tiles: byte offset of their first output pixel
  unsigned pnrec[C::NT];   // ... and the descriptor size: the output's, or 0 (no such tile: step 0, the odd tail) — stores dropped
#pragma unroll
  for (int t = 0; t < C::NT; ++t) { pbyte[t] = 0; pnrec[t] = 0; }
  u32x4 rv;
  const unsigned out_bytes = (unsigned)((long long)p.B * 4 * p.Hs * p.Ws * N * 2);   // < 0xffffff00 (host check)
  const int po = tid / PPO_, pj = tid - po * PPO_;
  const int plds = C::C_OFF + po * C::CROW + (((pj ^ (po >> 1)) & (PPO_ - 1)) << 4);
  const unsigned pgl = (unsigned)((((po >> 5) * (2 * p.Ws) + (po & 31)) * N + pj * 8) * 2);
  const unsigned prow = (unsigned)(2 * p.Ws * N * 2);   // bytes per output row
  auto piece_lds = [&](int q8) {   // q8 is a compile-time constant at every call
    const int t = (q8 * OPQ) / C::OPX, oq = (q8 * OPQ) % C::OPX;
    return *reinterpret_cast<const u32x4*>(smem + plds + t * C::CB + oq * C::CROW);
  };
  auto piece_store = [&](int q8, const u32x4 v) {
    const int t = (q8 * OPQ) / C::OPX, oq = (q8 * OPQ) % C::OPX;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)pnrec[t], 0x00027000);
    __builtin_amdgcn_raw_buffer_store_b128(v, ors, pgl, (int)(pbyte[t] + (unsigned)(oq >> 5) * prow), 0);
  };

  // one class of one tile: F = taps * KB fragments, ring position OFF at entry (F % RING != 0 only for CS = 64, class 3)
  auto run_class = [&](auto cls_c, auto off_c, auto next_c) {
    constexpr int CLS = decltype(cls_c)::value, OFF = decltype(off_c)::value;
    constexpr int PY = CLS >> 1, PX = CLS & 1, NKX = nk(PX), F = ntaps_of(CLS) * C::KB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    wzero = 0;
    asm volatile("" : "+s"(wzero));  // opaque to LICM (a scalar pair; the adds below are SALU)
    bf16x8 a[2][4];
    auto a_off = [&](int f) {  // compile-time after unrolling
      const int t = f / C::KB, kb = f - t * C::KB;
      const int ta = t / NKX, tb = t - ta * NKX;
      return ((tap_d(PY, ta) + 1) * HWT + tap_d(PX, tb) + 1) * C::PITCH + kb * 32;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(smem + abase[i] + a_off(0));
    __builtin_amdgcn_sched_barrier(0);
    // (a compile-time loop: with the deferred row sweep's f-dependent pieces inside, "#pragma unroll" left the 72-fragment loop
    //  rolled — ring slots through s_set_gpr_idx)
    lg_static_for<F>([&](auto f_c) {
      constexpr int f = decltype(f_c)::value;
      if constexpr (f + 1 < F) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[(f + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(smem + abase[i] + a_off(f + 1 < F ? f + 1 : 0));
      }
      if constexpr (LG_U3_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      const int sl = (f + OFF) % RING;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[sl]), a[f & 1][i], acc[i], 0, 0, 0);
      if constexpr (LG_U3_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      // the stream wraps: the same class every step — or (ROT) runs on into the first fragments of the wave's NEXT class
#ifndef LG_U3_NO_WLOAD   // ablation (round 5, scripts/probe/u3_nowload.sh): the ring is primed once and never refilled — results wrong, timing only
      if constexpr (f + RING < F) bf[sl] = *reinterpret_cast<const u32x4*>(frag_ptr(cls_c, f + RING));
      else bf[sl] = *reinterpret_cast<const u32x4*>(frag_ptr(next_c, f + RING - F));
#endif
      // deferred row sweep of the previous tile: piece q is read from the staging area at fragment 1 + q STEP and stored one STEP later
      constexpr int STEP = F >= 2 * NPC + 2 ? 2 : 1;
      static_assert(!DEFER || F >= NPC * STEP + 2, "class loop too short for the deferred row sweep");
      constexpr bool RD = DEFER && f >= 1 && (f - 1) % STEP == 0 && (f - 1) / STEP < NPC;
      constexpr bool ST = DEFER && f >= 1 + STEP && (f - 1 - STEP) % STEP == 0 && (f - 1 - STEP) / STEP < NPC;
      // The pieces sit between two sched_barrier(0) at the fragment boundary.  Handed to the group pipeline below as groups of their
      // own (VMEM write, DS read), the piece read could be picked by ANY earlier "DS read" group: the whole class loop came out scrambled
      // (ring waits collapsed to vmcnt(0..3), MFMAs of different fragments interleaved, 12 min of compile time).
      if constexpr (ST || RD) __builtin_amdgcn_sched_barrier(0);
      if constexpr (ST) piece_store((f - 1 - STEP) / STEP, rv);   // one register: piece q leaves, piece q + 1 is read behind it
      if constexpr (RD) rv = piece_lds((f - 1) / STEP);
      if constexpr (ST || RD) __builtin_amdgcn_sched_barrier(0);
      if constexpr (LG_U3_SCHED == 0) {
        __builtin_amdgcn_sched_barrier(0);
      } else {  // one MFMA, one LDS read in its shadow, ..., the ring refill behind the last MFMA (see conv_down3.hip)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    });
  };

#ifdef LG_U3_STAMPS
  int nst = 0;
#define U3_STAMP() do { if (p.stamps && nst < 32) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) p.stamps[((long long)blockIdx.x * NWV + wid) * 32 + nst] = t_; ++nst; } } while (0)
#else
#define U3_STAMP() do {} while (0)
#endif
  int roff = 0;  // ring position at the start of this wave's class (toggles 0 / 4 when F % RING == 4)
  for (int s = 0; s < nmine; ++s) {
    const bool more = s + 1 < nmine;
    int n, y0, x0;
    U3_STAMP();  // item start
    const bool live = tile_of(s, tsel, n, y0, x0);  // (N = 32: the second tile of the last step may not exist: computed, not stored)

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using IH = std::integral_constant<int, (RING / 2) % RING>;
    if constexpr (C::ROT) {
      // rotation 3 -> 1 -> 0 -> 2 -> 3 ...; the ring position at a class's entry is 0 or RING / 2 (only the 9-tap class of the
      // 64-channel form has a fragment count that is not a multiple of RING): both instantiations of every class
      if (cls == 3) { if (roff == 0) run_class(I3{}, I0{}, I1{}); else run_class(I3{}, IH{}, I1{}); }
      else if (cls == 1) { if (roff == 0) run_class(I1{}, I0{}, I0{}); else run_class(I1{}, IH{}, I0{}); }
      else if (cls == 0) { if (roff == 0) run_class(I0{}, I0{}, I2{}); else run_class(I0{}, IH{}, I2{}); }
      else { if (roff == 0) run_class(I2{}, I0{}, I3{}); else run_class(I2{}, IH{}, I3{}); }
      roff = (roff + ntaps_of(cls) * C::KB) % RING;
    } else if (cls == 3) {
      if (roff == 0) run_class(I3{}, I0{}, I3{});
      else run_class(I3{}, IH{}, I3{});
      roff = (roff + ntaps_of(3) * C::KB) % RING;
    } else if (cls == 2) run_class(I2{}, I0{}, I2{});
    else if (cls == 1) run_class(I1{}, I0{}, I1{});
    else run_class(I0{}, I0{}, I0{});

    U3_STAMP();  // class computed
    // The next step's halo is requested HERE, behind the MFMA stream, and lands under the staging pass and the barrier below.
    // (vmcnt retires in order: requested at the top of the step, these HBM loads sat in front of the weight ring's refills and
    //  every ring wait of the first fragments also waited for them.)  Last step: the current halo again — valid addresses, result
    // unused, no branch around the loads.
    issue(more ? s + 1 : s, hv);
    if constexpr (DEFER) __syncthreads();  // every halo read of this step and every row read of the previous tile is done
    // ---- this wave's class into the output-tile staging area: acc[i][e] = channel (e&3) + 8*(e>>2) + 4*h of pixel m ----------
    {
      char* Cst = smem + C::C_OFF + tsel * C::CB;
      const int py = cls >> 1, px = cls & 1;
      const float shift = STATS ? sbias[0] : 0.f;
      const f32x2 shift2 = {shift, shift};
      f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
      f32x4 bq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(sbias + wn * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i * 32 + pix32(r);
        const int o = (2 * (m >> 4) + py) * (2 * TW) + 2 * (m & 15) + px;  // output pixel inside the 16 x 32 tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const f32x2 v = f32x2{acc[i][4 * g + 2 * jj], acc[i][4 * g + 2 * jj + 1]} + f32x2{bq[g][2 * jj], bq[g][2 * jj + 1]};
            if constexpr (STATS) {
              const f32x2 d = v - shift2;
              s1v += d;
              s2v = __builtin_elementwise_fma(d, d, s2v);
            }
            w[2 * jj] = (__bf16)v[0]; w[2 * jj + 1] = (__bf16)v[1];
          }
          // piece (wn*4 + g) of the pixel's row, XOR-swizzled with the pixel pair index (neighbouring lanes are 2 pixels apart)
          constexpr int PM = C::CROW / 16 - 1;
          *reinterpret_cast<bf16x4*>(Cst + o * C::CROW + ((((wn * 4 + g) ^ (o >> 1)) & PM) << 4) + 8 * h) = w;
        }
      }
      if constexpr (STATS) sstat[tid] = f32x2{s1v[0] + s1v[1], s2v[0] + s2v[1]};  // reduced by ONE wave per tile behind the barrier
    }
    U3_STAMP();  // staged
    int tns[C::NT], ty0s[C::NT], tx0s[C::NT];
    bool tlive[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) tlive[t] = tile_of(s, t, tns[t], ty0s[t], tx0s[t]);
    constexpr int PPO = C::CROW / 16;                 // pieces per output pixel (8 | 4)
    constexpr int TOT = C::NT * C::OPX * PPO;         // 4096 pieces per step either way
    // FUSE: the z pieces this thread will meet in the row sweep are requested NOW — the accumulators are dead (staged), and
    // the loads land behind the barrier and the halo commit instead of in front of every use
    u32x4 zq[FUSE ? TOT / NTH : 1];
    (void)zq;
    if constexpr (FUSE) {
#pragma unroll
      for (int q8 = 0; q8 < TOT / NTH; ++q8) {
        const int q = tid + q8 * NTH;
        const int t = q / (C::OPX * PPO), rem = q - t * (C::OPX * PPO);
        const int o = rem / PPO, j = rem - o * PPO;
        const bool t1 = C::NT == 2 && t != 0;
        const int tn_ = t1 ? tns[C::NT - 1] : tns[0], ty0 = t1 ? ty0s[C::NT - 1] : ty0s[0], tx0 = t1 ? tx0s[C::NT - 1] : tx0s[0];
        const bool tl = t1 ? tlive[C::NT - 1] : tlive[0];
        const long long goff = ((long long)(tn_ * 2 * p.Hs + 2 * ty0 + (o >> 5)) * (2 * p.Ws) + 2 * tx0 + (o & 31)) * N + j * 8;
        zq[q8] = tl ? *reinterpret_cast<const u32x4*>(p.nf.z + goff) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    if constexpr (!DEFER) __syncthreads();  // every class of the tile(s) is staged, every halo read is done
    U3_STAMP();  // barrier 1 passed

    if (more) commit(hv);
    if constexpr (DEFER) {
      __syncthreads();  // tile staged, next halo in place; its rows leave inside the next class loop (or below, after the last step)
#pragma unroll
      for (int t = 0; t < C::NT; ++t) {
        // (readfirstlane: both ride in scalar operands of the row stores — descriptor size and scalar offset; see issue())
        pbyte[t] = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long long)(tns[t] * 2 * p.Hs + 2 * ty0s[t]) * (2 * p.Ws) + 2 * tx0s[t]) * N) * 2));
        pnrec[t] = (unsigned)__builtin_amdgcn_readfirstlane((int)(tlive[t] ? out_bytes : 0u));
      }
    } else
    // ---- whole output rows out: 16 rows x (32 pixels x N channels) contiguous, 16 B per lane ------------------------------
    {
      float nf1 = 0.f, nf2 = 0.f;
      (void)nf1; (void)nf2;
#pragma unroll
      for (int q8 = 0; q8 < TOT / NTH; ++q8) {
        const int q = tid + q8 * NTH;
        const int t = q / (C::OPX * PPO), rem = q - t * (C::OPX * PPO);
        const int o = rem / PPO, j = rem - o * PPO;
        const bool t1 = C::NT == 2 && t != 0;
        const int tn_ = t1 ? tns[C::NT - 1] : tns[0], ty0 = t1 ? ty0s[C::NT - 1] : ty0s[0], tx0 = t1 ? tx0s[C::NT - 1] : tx0s[0];
        const bool tl = t1 ? tlive[C::NT - 1] : tlive[0];
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + C::C_OFF + t * C::CB + o * C::CROW + (((j ^ (o >> 1)) & (PPO - 1)) << 4));
        const long long goff = ((long long)(tn_ * 2 * p.Hs + 2 * ty0 + (o >> 5)) * (2 * p.Ws) + 2 * tx0 + (o & 31)) * N + j * 8;
        if (tl) *reinterpret_cast<u32x4*>(p.out + goff) = v;
        if constexpr (FUSE) {
          if (tl) {
            const lg_const_f32p sp = lg_as_const(p.nf.stats + (long long)tn_ * 8);   // scalar loads (lg_common.h)
            lg_nf_accum(v, zq[q8], sp[0], sp[4], sp[2], sp[3], p.nf.alpha, nf1, nf2);
          }
        }
      }
      if constexpr (FUSE) sstat[tid] = f32x2{nf1, nf2};
    }
    if constexpr (FUSE) {
      __syncthreads();
      const int t = wid / (NWV / C::NT);
      if (wid == t * (NWV / C::NT)) {
        double S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int w = 0; w < NWV / C::NT; ++w) {
          const f32x2 v = sstat[(t * (NWV / C::NT) + w) * 64 + lane];
          S1 += (double)v[0]; S2 += (double)v[1];
        }
        S1 = lg_wave_sum_d(S1); S2 = lg_wave_sum_d(S2);
        const bool t1 = C::NT == 2 && t != 0;
        const bool tl = t1 ? tlive[C::NT - 1] : tlive[0];
        const int sy0 = t1 ? ty0s[C::NT - 1] : ty0s[0], sx0 = t1 ? tx0s[C::NT - 1] : tx0s[0], sn = t1 ? tns[C::NT - 1] : tns[0];
        if (lane == 0 && tl) {
          double* o = p.nf.part + ((long long)sn * p.tpi + (sy0 / TH) * p.tpi_x + sx0 / TW) * 2;
          o[0] = S1; o[1] = S2;
        }
      }
    }
    if constexpr (STATS) {
      // the first wave of each tile adds the per-thread partials of that tile's waves (fp64 from here on) while the other
      // waves are still writing rows: one cross-lane reduction per tile instead of one per wave
      const int t = wid / (NWV / C::NT);
      if (wid == t * (NWV / C::NT)) {
        double S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int w = 0; w < NWV / C::NT; ++w) {
          const f32x2 v = sstat[(t * (NWV / C::NT) + w) * 64 + lane];
          S1 += (double)v[0]; S2 += (double)v[1];
        }
        S1 = lg_wave_sum_d(S1); S2 = lg_wave_sum_d(S2);
        const bool t1 = C::NT == 2 && t != 0;
        const bool tl = t1 ? tlive[C::NT - 1] : tlive[0];
        const int sy0 = t1 ? ty0s[C::NT - 1] : ty0s[0], sx0 = t1 ? tx0s[C::NT - 1] : tx0s[0], sn = t1 ? tns[C::NT - 1] : tns[0];
        if (lane == 0 && tl) {
          constexpr double cnt = (double)(C::OPX * N);
          const double md = S1 / cnt;
          const int tin = (sy0 / TH) * p.tpi_x + sx0 / TW;
          double* o = p.spart + ((long long)sn * p.tpi + tin) * 3;
          o[0] = cnt; o[1] = (double)sbias[0] + md; o[2] = S2 - cnt * md * md;
        }
      }
    }
    (void)live;
    U3_STAMP();  // rows out
    if constexpr (C::ROT) cls = rot_class(wid + rsh + s + 1);
    if constexpr (!DEFER) __syncthreads();  // next halo complete, staging area free again
  }
  if constexpr (DEFER) {  // the last tile's rows
#pragma unroll
    for (int q8 = 0; q8 < NPC; ++q8) piece_store(q8, piece_lds(q8));
  }
}

template <int CS, int N, int NTT = 2 / (N / 32)>
int launch_up3(U3Params p, bool stats, hipStream_t st, bool fuse = false) {
  using C = Cfg<CS, N, NTT>;
  static bool attr_set = false;
  const int nblk = (C::NWAVES == 8 ? 1 : 2) * lg_grid_cus();  // one 8-wave workgroup per CU, or two independent 4-wave ones
  if (!attr_set) {
    attr_set = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_up3_kernel<CS, N, true, false, NTT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_up3_kernel<CS, N, false, false, NTT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_up3_kernel<CS, N, false, true, NTT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
  }
  const int nsteps = (p.nitems + C::NT - 1) / C::NT;
  const int grid = nsteps < nblk ? nsteps : nblk;
  if (fuse) hipLaunchKernelGGL((conv_up3_kernel<CS, N, false, true, NTT>), dim3(grid), dim3(C::THREADS), C::LDS, st, p);
  else if (stats) hipLaunchKernelGGL((conv_up3_kernel<CS, N, true, false, NTT>), dim3(grid), dim3(C::THREADS), C::LDS, st, p);
  else hipLaunchKernelGGL((conv_up3_kernel<CS, N, false, false, NTT>), dim3(grid), dim3(C::THREADS), C::LDS, st, p);
  return LG_OK;
}

}  // namespace

extern "C" int lg_conv_up3_supported(int B, int Hm, int Wm, int Cs, int N) {
  return (!lg_env_flag("LG_NO_UP3") && B > 0 && Hm % TH == 0 && Wm % TW == 0 && ((Cs == 128 && N == 64) || (Cs == 64 && N == 32))) ? 1 : 0;
}

// LG_OK: launched.  LG_ERR_UNSUPPORTED: the caller falls back to conv_halo.hip.  *nparts_out = records per sample (tiles).
extern "C" int lg_conv_up3_nf_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                                  int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                  size_t nf_bytes, void* stream);
extern "C" int lg_conv_up3_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                               int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream) {
  return lg_conv_up3_nf_try(src16, wpack_up, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, nullptr, 0, stream);
}
// nf (optional; data-gradient use): also writes the norm-backward sums of the produced gradient, *nparts_out = records
// per sample in nf->part ([B][nparts][2] doubles, nf_bytes available)
extern "C" int lg_conv_up3_nf_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                                  int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                  size_t nf_bytes, void* stream) {
  if (nparts_out) *nparts_out = 0;
  if (!src16 || !wpack_up || !out16 || !lg_conv_up3_supported(B, Hm, Wm, Cs, N)) return LG_ERR_UNSUPPORTED;
  if ((long long)Hm * Wm * Cs * 2 * 2 >= (1ll << 31)) return LG_ERR_UNSUPPORTED;  // buffer descriptor: two samples below the OOB offset
  if ((long long)B * 4 * Hm * Wm * N * 2 >= 0xffffff00ll) return LG_ERR_UNSUPPORTED;   // the row stores: one descriptor over the whole output
  U3Params p{};
  p.src = (const __bf16*)src16; p.wp = (const char*)wpack_up; p.bias = bias; p.out = (__bf16*)out16;
  p.B = B; p.Hs = Hm; p.Ws = Wm; p.tpi_x = Wm / TW; p.tpi = p.tpi_x * (Hm / TH);
  const long long nitems = (long long)B * p.tpi;
  if (nitems <= 0 || nitems >= (1ll << 30)) return LG_ERR_UNSUPPORTED;
  p.nitems = (int)nitems;
#ifdef LG_U3_STAMPS
  { const char* e = getenv("LG_U3_STAMPBUF"); p.stamps = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  // N = 32: two independent 4-wave workgroups per CU, one tile per step (LG_U3_T4_8W=1: the round-2 form, one 8-wave workgroup
  // with two tiles per step)
  static int t4_8w = -1;
  if (t4_8w < 0) t4_8w = lg_env_flag("LG_U3_T4_8W") ? 1 : 0;
  { static int ro = -1; if (ro < 0) { const char* e = getenv("LG_U3_ROTOFF"); ro = e ? (atoi(e) & 3) : 0; } p.rotoff = ro; }   // A/B (round 5, measured: 141 - 145 us for offsets 0 / 1 / 2 — no effect, default 0)
  // the norm-backward sums are produced by the one-tile-per-step forms only: with two tiles per step a thread's row sweep covers
  // both tiles, i.e. possibly two samples, and the per-thread sums would mix them (no layer of the step asks for that form)
  const bool fuse = (Cs == 128 || !t4_8w) && nf && nf->z && nf->stats && nf->part && nparts_out && (size_t)B * p.tpi * 2 * sizeof(double) <= nf_bytes;
  const bool stats = !fuse && spart && nparts_out && (size_t)B * p.tpi * 3 * sizeof(double) <= spart_bytes;
  // moments asked for (the caller may then write z as bf16 ONLY) but the workspace cannot hold this tiling's records: decline, so
  // that the dispatch chain tries the next kernel instead of launching without them (lg_conv_fwd_stats_fused's promise)
  if (!nf && spart && nparts_out && !stats) return LG_ERR_UNSUPPORTED;
  p.spart = stats ? (double*)spart : nullptr;
  if (fuse) p.nf = *nf;
  hipStream_t st = (hipStream_t)stream;
  if (Cs == 128) launch_up3<128, 64>(p, stats, st, fuse);
  else if (t4_8w) launch_up3<64, 32>(p, stats, st, fuse);
  else launch_up3<64, 32, 1>(p, stats, st, fuse);
  LG_CHECK_LAUNCH("lg_conv_up3");
  lg_note_kernel(Cs == 128 ? "conv_up3_kernel<128,64>" : t4_8w ? "conv_up3_kernel<64,32>" : "conv_up3_kernel<64,32,4w>");
  if (stats || fuse) *nparts_out = p.tpi;
  return LG_OK;
}
