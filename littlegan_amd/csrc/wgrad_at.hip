// All-taps weight gradient of the 5x5 stride-2 layers from the bf16 mirrors (8x8 maps and larger):
//   dW[ky,kx,cb,cs] = sum_{n,y,x} big[n, 2y+ky-1, 2x+kx-1, cb] * small[n, y, x, cs]       (wgrad_igemm.hip:1-5)
// The per-tap kernel of wgrad_igemm.hip stages a [pixels x (32..128 + 64..128) channels] operand pair per tap: at the
// 32/64-channel levels that is 3 KB of staging for two MFMAs, and `big` is pulled through L2 25 times.  Here one block
// (8 waves, one per CU) owns ALL 25 taps of a 32 (cb) x 64 (cs) slice of dW — 50 accumulator tiles of 32x32 — and walks
// a list of items (sample, band of R small rows, strip of SW small columns):
//   * LDS holds the (2R+3) x (2SW+3) halo of `big` (its 32 channels) and the R x SW pixels of `small` (64 channels),
//     double buffered, staged global -> registers -> LDS behind the MFMAs of the previous item;
//   * `big` pixels are stored de-interleaved by x parity, `small` split into its two 32-channel halves: every fragment
//     is then 16 consecutive 64-B rows, which is what ds_read_b64_tr_b16 reads without bank conflicts
//     (wgrad_igemm.hip:32-36), and a tap is just a byte offset (compile-time per k step, one add per wave for the tap);
//   * a k step is 16 small pixels of one row; an item has 8 of them.  Wave w owns taps 3w..3w+2 (both cs halves: 6 tiles,
//     A fragment read once, used twice); tap 24 rotates — wave w runs it on k step w of every item and the 8 partial
//     tiles are summed through LDS at the end — so every wave issues exactly 50 MFMAs per item.
//   * output: slab[split][tap][cb][cs] (fp32), reduced in fixed order by slab_reduce4_kernel (deterministic, no atomics).
#include <stdlib.h>
#include "lg_common.h"

#ifndef LG_WGAT_SCHED
#define LG_WGAT_SCHED 0   // 0: k step in pinned groups (sched_barrier); 1: interleaved by sched_group_barrier (MFMA, two LDS reads, ...)
#endif
#ifndef LG_WGAT_DBG
#define LG_WGAT_DBG 0   // timing ablations (results wrong): 1 no MFMA, 2 no fragment reads in the k loop, 4 no staging after the first item
#endif

namespace {

struct WgAtParams {
  const __bf16* big;    // [B, 2Hm, 2Wm, Cb]
  const __bf16* small;  // [B, Hm, Wm, Cs]
  float* slab;          // [nsplit][25][Cb][Cs]
  int B, Hm, Wm, Cb, Cs;
  int nuj, nunits;      // units = (Cb/32) x (Cs/64)
  int items_total, items_per;
  int nsplit, interleave;   // interleave (round 5, LG_WGAT_INTERLEAVE): split s walks items s, s + nsplit, ... — the chip sweeps the maps front to back together
};

// SW == 8 is the 8x8-map configuration: an item is a PAIR of samples (2 x 64 small pixels = 8 k steps), each with its own
// halo image in LDS; a k step is two small rows of one sample (the transposed read takes one address per row, so the two
// rows need not be adjacent in LDS — only the lane's row offset changes).
template <int SW, int R>
struct AtCfg {
  static constexpr bool PAIR = SW == 8;
  static_assert(PAIR ? R == 8 : R * SW / 16 == 8, "8 k steps per item");
  static constexpr int NS = PAIR ? 2 : 1;         // samples per item
  static constexpr int NA = SW + 2;               // entries per x-parity array (odd: SW+2 used, even: SW+1)
  static constexpr int EVEN_OFF = NA * 64;        // odd-x array first
  static constexpr int ROWP = 2 * NA * 64;        // bytes per big row
  static constexpr int NBR = 2 * R + 3, NPX = 2 * SW + 3;
  static constexpr int BIGS = NBR * ROWP;         // one sample's halo image
  static constexpr int BIG = NS * BIGS, SMALL = NS * R * 2 * SW * 64, BUF = BIG + SMALL;
  static constexpr int NBPS = NBR * NPX * 4;      // 16-B pieces of one halo image
  static constexpr int NBP = NS * NBPS, NBL = (NBP + 511) / 512;  // ... per item, per thread
  static constexpr int NSP = NS * R * SW * 8, NSL = NSP / 512;
  static_assert(BUF < 65536, "k-step offsets must fit the ds offset field");
  static_assert(2 * BUF <= 160 * 1024 && 2 * BUF >= 8 * 2 * 4096, "LDS budget; the tap-24 reduction reuses it");
};

__device__ __forceinline__ f32x16 at_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (LG_WGAT_DBG & 1) { c[0] += (float)a[0] * (float)b[0]; return c; }  // keeps the operands alive
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 rd_tr(const char* p) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 256));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int SW, int R>
__global__ __launch_bounds__(512) void wgrad_at_kernel(const WgAtParams p) {
  using C = AtCfg<SW, R>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = lg_xcd_remap(blockIdx.x, gridDim.x);
  const int unit = bx % p.nunits, split = bx / p.nunits;   // units of one split run together: they share the pixels
  const int i0 = (unit / p.nuj) * 32, j0 = (unit % p.nuj) * 64;
  const int it0 = p.interleave ? 0 : split * p.items_per;
  const int it1 = p.interleave ? (p.items_total - split + p.nsplit - 1) / p.nsplit : min(it0 + p.items_per, p.items_total);   // local item indices it0 .. it1 - 1
  auto item_of = [&](int j) { return p.interleave ? split + j * p.nsplit : j; };
  const int nxs = p.Wm / SW, nyb = p.Hm / R;
  const int Hb = 2 * p.Hm, Wb = 2 * p.Wm;

  u32x4 rbig[C::NBL], rsm[C::NSL];
  auto load_item = [&](int it) {
    int n, yb, xs;
    if constexpr (C::PAIR) { n = 2 * it; yb = 0; xs = 0; }
    else { xs = it % nxs; const int t2 = it / nxs; yb = t2 % nyb; n = t2 / nyb; }
    const int gy0 = 2 * yb * R - 1, gx0 = 2 * xs * SW - 1;
    const __bf16* bbase = p.big + (long long)n * Hb * Wb * p.Cb + i0;
    const __bf16* sbase = p.small + ((long long)(n * p.Hm + yb * R) * p.Wm + xs * SW) * p.Cs + j0;
#pragma unroll
    for (int k = 0; k < C::NBL; ++k) {
      const int q = tid + k * 512, si = C::PAIR ? q / C::NBPS : 0, qs = q - si * C::NBPS;
      const int piece = qs & 3, pp = qs >> 2;
      const int row = pp / C::NPX, px = pp - row * C::NPX;
      const int gy = gy0 + row, gx = gx0 + px;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < C::NBP && (unsigned)gy < (unsigned)Hb && (unsigned)gx < (unsigned)Wb)
        v = *reinterpret_cast<const u32x4*>(bbase + ((long long)si * Hb * Wb + (long long)(gy * Wb + gx)) * p.Cb + piece * 8);
      rbig[k] = v;
    }
#pragma unroll
    for (int k = 0; k < C::NSL; ++k) {
      const int q = tid + k * 512, c8 = q & 7, pp = q >> 3;
      if constexpr (C::PAIR) {  // pp = sample * 64 + pixel: the two samples are adjacent in memory
        rsm[k] = *reinterpret_cast<const u32x4*>(sbase + (long long)pp * p.Cs + c8 * 8);
      } else {
        const int yy = pp / SW, px = pp % SW;
        rsm[k] = *reinterpret_cast<const u32x4*>(sbase + (long long)(yy * p.Wm + px) * p.Cs + c8 * 8);
      }
    }
  };
  auto store_item = [&](char* buf) {
#pragma unroll
    for (int k = 0; k < C::NBL; ++k) {
      const int q = tid + k * 512, si = C::PAIR ? q / C::NBPS : 0, qs = q - si * C::NBPS;
      const int piece = qs & 3, pp = qs >> 2;
      const int row = pp / C::NPX, px = pp - row * C::NPX;
      // px 0 <-> big x = 2*X0 - 1 (odd, entry 0); px 1 <-> 2*X0 (even, entry 0); ...
      if (q < C::NBP)
        *reinterpret_cast<u32x4*>(buf + si * C::BIGS + row * C::ROWP + ((px & 1) ? C::EVEN_OFF : 0) + (px >> 1) * 64 + piece * 16) = rbig[k];
    }
#pragma unroll
    for (int k = 0; k < C::NSL; ++k) {
      const int q = tid + k * 512, c8 = q & 7, pp = q >> 3;
      if constexpr (C::PAIR) {  // [sample][half][64 pixels]: a k step = 16 consecutive pixels (two rows of 8)
        const int si = pp >> 6, pix = pp & 63;
        *reinterpret_cast<u32x4*>(buf + C::BIG + ((si * 2 + (c8 >> 2)) * 64 + pix) * 64 + (c8 & 3) * 16) = rsm[k];
      } else {
        const int yy = pp / SW, px = pp % SW;
        *reinterpret_cast<u32x4*>(buf + C::BIG + ((yy * 2 + (c8 >> 2)) * SW + px) * 64 + (c8 & 3) * 16) = rsm[k];
      }
    }
  };

  // fragment addressing (wgrad_igemm.hip:195-201): 16-lane group g = (channel half, k half), lane 4q+p -> row q, cols 4p..
  const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int lanepart = (8 * (g >> 1) + lq) * 64 + (16 * (g & 1) + 4 * lp) * 2;
  // PAIR: the k half (g >> 1) is the second small row of the step: two big rows further down, same entries
  const int lanepartA = C::PAIR ? (g >> 1) * 2 * C::ROWP + lq * 64 + (16 * (g & 1) + 4 * lp) * 2 : lanepart;
  int ab[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = 3 * wid + i, ky = t / 5, kx = t - ky * 5;
    // tap (ky,kx) at small pixel (yy, x): big row 2yy+ky of the halo; x = 2x+kx-1 -> odd array for even kx, entry x + (kx>>1)
    ab[i] = lanepartA + ky * C::ROWP + ((kx & 1) ? C::EVEN_OFF : 0) + (kx >> 1) * 64;
  }
  const int a24 = lanepartA + 4 * C::ROWP + 2 * 64;
  const int bb = lanepart + C::BIG;

  f32x16 acc[3][2], acc24[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i][0][e] = 0.f; acc[i][1][e] = 0.f; }
    acc24[0][e] = 0.f; acc24[1][e] = 0.f;
  }

  if (it0 < it1) {
    load_item(item_of(it0));
    store_item(smem);
  }
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    const int cur = (it - it0) & 1;
    if (it + 1 < it1 && !(LG_WGAT_DBG & 4)) load_item(item_of(it + 1));
    const char* sb = smem + cur * C::BUF;
    // k-step offsets of the two operands (compile-time once the loop is unrolled).  PAIR: k step ks = rows 2(ks&3),
    // 2(ks&3)+1 of sample ks >> 2; otherwise the 16 pixels of chunk xc of row yy.
    auto koffA = [](int ks) {
      constexpr int CH = SW / 16 > 0 ? SW / 16 : 1;
      return C::PAIR ? (ks >> 2) * C::BIGS + 4 * (ks & 3) * C::ROWP : 2 * (ks / CH) * C::ROWP + (ks % CH) * 16 * 64;
    };
    auto koffB = [](int ks, int h) {
      constexpr int CH = SW / 16 > 0 ? SW / 16 : 1;
      return C::PAIR ? (((ks >> 2) * 2 + h) * 64 + 16 * (ks & 3)) * 64 : (((ks / CH) * 2 + h) * SW + (ks % CH) * 16) * 64;
    };
    // Order PINNED (sched_barrier; left alone hipcc puts every fragment read right in front of its MFMA and waits for it):
    // a step starts with a0, b0, b1 already in registers (requested during the previous step), requests a1, a2, runs the
    // two MFMAs of a0 under that latency, requests the next step's a0, b0, b1, then runs the other four.
    bf16x8 b[2], a0;
#pragma unroll
    for (int h = 0; h < 2; ++h) b[h] = rd_tr(sb + bb + koffB(0, h));
    a0 = rd_tr(sb + ab[0] + koffA(0));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if constexpr (LG_WGAT_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      const bf16x8 a1 = (LG_WGAT_DBG & 2) ? a0 : rd_tr(sb + ab[1] + koffA(ks)), a2 = (LG_WGAT_DBG & 2) ? a0 : rd_tr(sb + ab[2] + koffA(ks));
      if constexpr (LG_WGAT_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = at_mfma(a0, b[0], acc[0][0]);
      acc[0][1] = at_mfma(a0, b[1], acc[0][1]);
      if constexpr (LG_WGAT_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      bf16x8 bn[2] = {b[0], b[1]}, a0n = a0;
      if (ks + 1 < 8 && !(LG_WGAT_DBG & 2)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) bn[h] = rd_tr(sb + bb + koffB(ks + 1, h));
        a0n = rd_tr(sb + ab[0] + koffA(ks + 1));
      }
      if constexpr (LG_WGAT_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      acc[1][0] = at_mfma(a1, b[0], acc[1][0]);
      acc[1][1] = at_mfma(a1, b[1], acc[1][1]);
      acc[2][0] = at_mfma(a2, b[0], acc[2][0]);
      acc[2][1] = at_mfma(a2, b[1], acc[2][1]);
      if constexpr (LG_WGAT_SCHED != 0 && (LG_WGAT_DBG & 3) == 0) {
        // the ten transposed reads of the step (a1, a2; then b0', b1', a0' of the next one) go two by two into the shadows of
        // the MFMAs: an MFMA holds the vector issue port for 8 of its 32 cycles
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (wid == ks) {
        const bf16x8 a3 = rd_tr(sb + a24 + koffA(ks));
#pragma unroll
        for (int h = 0; h < 2; ++h) acc24[h] = at_mfma(a3, b[h], acc24[h]);
      }
      b[0] = bn[0]; b[1] = bn[1]; a0 = a0n;
    }
    if (it + 1 < it1 && !(LG_WGAT_DBG & 4)) store_item(smem + (cur ^ 1) * C::BUF);
    __syncthreads();
  }

  // slab[split][t][cb][cs]: accumulator register e of lane (r, h) = row (e&3) + 8*(e>>2) + 4h, column r
  const int r = lane & 31, hh = lane >> 5;
  float* out = p.slab + (long long)split * 25 * p.Cb * p.Cs;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float* o = out + (long long)(3 * wid + i) * p.Cb * p.Cs + (long long)i0 * p.Cs + j0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[(long long)((e & 3) + 8 * (e >> 2) + 4 * hh) * p.Cs + h * 32 + r] = acc[i][h][e];
  }
  // tap 24: 8 partial tile pairs -> LDS [wave][h][e][lane], summed in wave order
  float* sred = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 16; ++e) sred[((wid * 2 + h) * 16 + e) * 64 + lane] = acc24[h][e];
  __syncthreads();
  {
    const int h = tid >> 8, e4 = (tid >> 6) & 3;
    float* o = out + (long long)24 * p.Cb * p.Cs + (long long)i0 * p.Cs + j0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += sred[((w * 2 + h) * 16 + 4 * e4 + q) * 64 + lane];
      o[(long long)(q + 8 * e4 + 4 * hh) * p.Cs + h * 32 + r] = s;
    }
  }
}

extern "C" int lg_device_cus(void);
// The split-K plan — and with it the fp32 summation order of dW — is a function of the DEVICE, not of the CU reservation (lg_set_reserved_cus):
// weight gradients stay bit-reproducible across reservation settings and across hipGraphs captured before a reservation changed.  With CUs
// reserved the grid is a few blocks larger than the free CUs: a short tail, not a different result.
inline int at_cus() { return lg_device_cus(); }

// 0: not applicable, else the strip width (32 or 16)
inline int at_shape(int Hm, int Wm, int cb, int cs) {
  if (cb % 32 || cs % 64 || cb < 32 || cs < 64) return 0;
  if ((cb / 32) * (cs / 64) > 64) return 0;   // deep levels: few pixels per unit, the per-tap kernel's big tiles do better
  if (Wm % 16 == 0 && Hm % 8 == 0) return 16;   // 16 x 8 strips re-read less halo than 32 x 4 (1.30x vs 1.44x of `big`)
  if (Wm % 32 == 0 && Hm % 4 == 0) return 32;
  if (Wm == 8 && Hm == 8) return 8;  // sample pairs (B even, checked by the caller)
  return 0;
}

inline void at_plan(int B, int Hm, int Wm, int cb, int cs, int sw, int* nsplit, int* items_total, int* items_per) {
  const int R = sw == 32 ? 4 : 8;
  const int nunits = (cb / 32) * (cs / 64);
  *items_total = sw == 8 ? B / 2 : B * (Hm / R) * (Wm / sw);
  int ns = at_cus() / nunits;
  if (ns < 1) ns = 1;
  if (ns > *items_total) ns = *items_total;
  *items_per = lg_cdiv(*items_total, ns);
  *nsplit = lg_cdiv(*items_total, *items_per);
}

}  // namespace

extern "C" size_t lg_wgrad_at_workspace_bytes(int B, int Hm, int Wm, int cb, int cs) {
  const int sw = at_shape(Hm, Wm, cb, cs);
  if (!sw || (sw == 8 && (B & 1))) return 0;
  int ns, tot, per;
  at_plan(B, Hm, Wm, cb, cs, sw, &ns, &tot, &per);
  return (size_t)ns * 25 * cb * cs * sizeof(float);
}

// writes slab[nsplit][25][cb][cs] into `workspace` and *nsplit_out; the caller reduces the slabs (lg_conv_wgrad_m16)
extern "C" int lg_wgrad_at_try(const void* big16, const void* small16, void* workspace, size_t ws_bytes, int B, int Hm, int Wm,
                               int cb, int cs, int* nsplit_out, void* stream) {
  static int off = -1;
  if (off < 0) off = lg_env_flag("LG_NO_WGAT") ? 1 : 0;
  const int sw = at_shape(Hm, Wm, cb, cs);
  if (off || !sw || (sw == 8 && (B & 1)) || !big16 || !small16 || !nsplit_out) return LG_ERR_UNSUPPORTED;
  WgAtParams p{};
  p.big = (const __bf16*)big16; p.small = (const __bf16*)small16; p.slab = (float*)workspace;
  p.B = B; p.Hm = Hm; p.Wm = Wm; p.Cb = cb; p.Cs = cs;
  p.nuj = cs / 64; p.nunits = (cb / 32) * (cs / 64);
  int ns;
  at_plan(B, Hm, Wm, cb, cs, sw, &ns, &p.items_total, &p.items_per);
  p.nsplit = ns;
  { static int il = -1; if (il < 0) il = lg_env_flag("LG_WGAT_INTERLEAVE") ? 1 : 0; p.interleave = il; }
  LG_CHECK_ARG(ws_bytes >= (size_t)ns * 25 * cb * cs * sizeof(float), "lg_wgrad_at: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_at_kernel<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AtCfg<32, 4>::BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_at_kernel<16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AtCfg<16, 8>::BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_at_kernel<8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AtCfg<8, 8>::BUF);
    attr = true;
  }
  constexpr int LDS32 = 2 * AtCfg<32, 4>::BUF, LDS16 = 2 * AtCfg<16, 8>::BUF, LDS8 = 2 * AtCfg<8, 8>::BUF;
  if (sw == 32) hipLaunchKernelGGL((wgrad_at_kernel<32, 4>), dim3(p.nunits * ns), dim3(512), LDS32, st, p);
  else if (sw == 16) hipLaunchKernelGGL((wgrad_at_kernel<16, 8>), dim3(p.nunits * ns), dim3(512), LDS16, st, p);
  else hipLaunchKernelGGL((wgrad_at_kernel<8, 8>), dim3(p.nunits * ns), dim3(512), LDS8, st, p);
  LG_CHECK_LAUNCH("lg_wgrad_at");
  lg_note_kernel(sw == 32 ? "wgrad_at_kernel<32,4>" : sw == 16 ? "wgrad_at_kernel<16,8>" : "wgrad_at_kernel<8,8>");
  *nsplit_out = ns;
  return LG_OK;
}
