// Final generator layer on the bf16 activation path, as a row-sliding product with NO block barrier and no P round trip:
//   y[n,y,x,co] = tanh(b[co] + sum_{ky,kx,c} h[n, y+2-ky, x+2-kx, c] * W[ky][kx][co][c])     /root/reference/model.py:86-87,104
// One wave owns a strip of 16 output columns and walks DOWN the input rows, holding FIVE output rows in flight
// (5 accumulator tiles of 16 pixels x 16 columns, columns 0..2 = co).  Input row yy is staged once (20 pixels x 32
// channels), its five kx-shifted A fragments are read with ds_read_b128, and 25 v_mfma_f32_16x16x32_bf16 add its
// contribution to the five output rows y = yy - 2 + ky it touches (B fragment (ky,kx) = W[ky][kx][co][c], resident in
// registers for the whole strip).  After that the oldest output row is complete: its 48 values go through 192 B of
// wave-private LDS to become one contiguous 192-B store after bias + tanh, and its accumulator restarts as the newest row.
// 13 of the 16 MFMA columns are padding — deliberately: the matrix pipe is otherwise idle in this HBM-bound layer and the
// alternative (15 real columns (ky,co) + a per-lane select / cross-lane shift-sum) costs 3x the VALU issue slots, which
// is what bounded the first version of this kernel.  Per row segment: 1.25 KB from L2/HBM (fetched once, 16 B per lane,
// coalesced, two rows ahead), 6.25 KB of LDS traffic, 25 MFMAs (400 issue cycles), ~50 other instructions.
// (Round 3, measured and dropped: the 25 B fragments in LDS instead of 100 VGPRs — 152 VGPRs, three blocks per CU instead of two,
//  but 30 instead of 5 ds_read_b128 per 25 MFMAs make the LDS array the bound: 97 -> 128 us at B = 256.)
// NORM: the input is the raw bf16 conv output z of the last decoder level and h = bf16(leaky(a*((z-mu)-mu_lo)+beta)) is
// formed while staging (same arithmetic, same rounding as apply16_kernel in norm.hip) — the stand-alone apply pass and the
// h16 tensor of the 128x128x32 map disappear.
#include <stdlib.h>
#include <type_traits>
#include "lg_common.h"

namespace {

struct RowsNormIn { const float* stats; float alpha; };
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 cvt8w(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  bf16x8 v;
  v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
  v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
  return v;
}

// XJ (1 | 2): x positions per accumulator row.  XJ = 1 is the form described at the top.  XJ = 2 (W % 32 == 0) puts TWO neighbouring
// output pixels of every co on the MFMA columns (column n = 3 j + co, j = 0 | 1): a wave then owns a strip of 32 columns, MFMA row m
// is the pixel PAIR (2m, 2m + 1), and the kx shifts of both pixels fold into SIX A fragments t = 0..5 (staged pixel 2m + t) with
// B_t[c][3 j + co] = W[ky][j + 4 - t][co][c] (zero where j + 4 - t is no tap): 30 MFMAs per 32 output pixels instead of 50, six
// ds_read_b128 instead of ten, half the per-row bookkeeping — the round-2 form ran 25 MFMAs of 16 issue cycles per 16 pixels, as
// long as the HBM time of the same pixels, and the two did not overlap (98 us at B = 256 against 45 us of bytes).
// LDS layout of a staged row, XJ = 2: pixel PAIRS at a pitch of 2 * 64 + 16 bytes — lane r of a fragment read sits r pairs on, so the
// 16 lanes r = 0..15 of one piece g fall into 16 different 4-bank sets (pitch / 4 mod 64 = 36: an odd multiple of 4).  The hardware's
// ds_read_b128 groups are not the four g, though (MI355X_MICROARCH, LDS: {0-3, 12-15, 20-27}, ...: each r once, half with piece g, half
// with g + 1), and the counters show 36 % of this kernel's LDS cycles as conflicts.  A layout that is conflict-free for those groups
// (four piece planes an odd number of 16-B slots apart) was built and measured: no gain on the raw-z form (LDS is 10 % of its wave cycles:
// 261 vs 249 - 257 us at 2B) and the plain form fell off the 256-VGPR edge into spills (78 -> 121 us).  Dropped.  XJ = 1 keeps the
// round-2 layout (64 B per pixel, 16-B pieces XOR-swizzled by the pixel quad): the padded form measured 130 us against 100 there.
template <int C, bool NORM, int XJ>
__global__ __launch_bounds__(256, XJ == 2 ? 2 : 1) void s1t_fwd_rows_kernel(const __bf16* __restrict__ x16, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                           int W, int RB, const RowsNormIn ni) {
  static_assert(C == 32, "one K = 32 MFMA step per fragment; the B fragments of 4 registers stay resident");
  static_assert(XJ == 1 || XJ == 2, "x positions per accumulator row");
  constexpr int SW = 16 * XJ, SP = SW + 4;          // output columns of a wave's strip; staged pixels of a row
  constexpr int NT = XJ + 4;                        // A fragments (pixel shifts) per row
  constexpr int NPC = C / 8, NP = SP * NPC, NL = (NP + 63) / 64;
  constexpr int GP = XJ == 1 ? C * 2 : XJ * C * 2 + 16;   // pitch of a group of XJ staged pixels
  constexpr int ROWB = (SP / XJ) * GP;              // staged row
  auto poff = [](int p, int j) {                    // byte offset of 16-B piece j of staged pixel p
    if constexpr (XJ == 1) return p * C * 2 + (((j ^ (p >> 2)) & (NPC - 1)) << 4);
    else return (p / XJ) * GP + (p % XJ) * C * 2 + j * 16;
  };
  constexpr int NO = 48 * XJ, VW = XJ, NLN = NO / VW;   // output floats of a row segment; floats per storing lane; storing lanes (48)
  constexpr unsigned OOB = 0x40000000u;            // buffer offsets >= this are out of range for any image here (< 1 GiB)
  // per wave: the staged row, then 64 x 16 B of dump slots (lanes with nothing to stage write there: the steady-state loop
  // has NO branch — a branch makes the compiler wait for ALL outstanding loads at the join, i.e. one HBM latency per row)
  __shared__ __attribute__((aligned(16))) char srow[4][2 * ROWB + 64 * 16];  // two row buffers: row s+1 is staged under the MFMAs of row s
  __shared__ __attribute__((aligned(16))) float sout[4][NO + 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nsx = (W + 4 * SW - 1) / (4 * SW), nry = (H + RB - 1) / RB;
  int bb = blockIdx.x;
  const int sx = bb % nsx; bb /= nsx;
  const int ry = bb % nry, n = bb / nry;
  const int x0 = sx * 4 * SW + wid * SW;
  if (x0 >= W) return;  // no block barrier anywhere below
  const int ya = ry * RB, yb = min(ya + RB, H), cnt = yb - ya + 4;
  char* my = srow[wid];
  float* mo = sout[wid];

  // B fragment (ky, t): column nn = 3 j + co (3 XJ .. 15 = padding), k = channel: lane l holds B[k = 8(l>>4) + i][col l&15];
  // its tap is kx = j + 4 - t
  bf16x8 bw[5][NT];
  {
    const int j = r / 3, co = r - 3 * j;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int kx = j + 4 - t;
        bw[ky][t] = (r < 3 * XJ && kx >= 0 && kx < 5) ? cvt8w(w + (long long)((ky * 5 + (kx < 0 ? 0 : kx > 4 ? 4 : kx)) * 3 + co) * C + g * 8)
                                                      : __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
      }
  }
  float bl[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) bl[k] = bias[(lane * VW + k) % 3];
  float mu = 0.f, mul = 0.f, na = 1.f, nb = 0.f;
  if constexpr (NORM) {
    const lg_const_f32p sp = lg_as_const(ni.stats + (long long)n * 8);   // scalar loads (lg_common.h)
    mu = sp[0]; na = sp[2]; nb = sp[3]; mul = sp[4];
  }
  // staging geometry: piece q = lane + 64k of the SP-pixel row -> pixel p, 16-B piece j.  The row is fetched with raw
  // buffer loads over this image (num_records = its bytes): pieces left / right of the image and rows above / below it
  // get an out-of-range offset and come back as zeros.
  unsigned goff[NL];
  int loff[NL], lbuf[NL];  // lbuf: distance to the same slot of the second row buffer (0 for a dump slot)
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int q = lane + 64 * k, p = q / NPC, j = q - p * NPC, xx = x0 - 2 + p;
    goff[k] = (q < NP && (unsigned)xx < (unsigned)W) ? (unsigned)(xx * C + j * 8) * 2u : OOB;
    loff[k] = q < NP ? poff(p, j) : 2 * ROWB + lane * 16;
    lbuf[k] = q < NP ? ROWB : 0;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x16 + (long long)n * H * W * C), 0, H * W * C * 2, 0x00027000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long long)n * H * W * 3, 0, H * W * 3 * 4, 0x00027000);
  const int rowbytes = W * C * 2;
  u32x4 rg[2][NL];
  // branch-free scalar select (a ?: here becomes a scalar branch, and a branch is a scheduling barrier for the MFMA stream)
  auto sel = [](bool c, unsigned a, unsigned b) { const unsigned m = 0u - (unsigned)c; return (a & m) | (b & ~m); };
  auto fetch = [&](int s, u32x4 (&dst)[NL]) {
    const int yy = ya - 2 + s;
    const unsigned ro = sel((unsigned)yy < (unsigned)H && s < cnt, (unsigned)(yy * rowbytes), OOB);
#pragma unroll
    for (int k = 0; k < NL; ++k) dst[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[k] + ro, 0, 0);
  };
  // A fragment of shift t: lane (r, g) needs staged pixel XJ r + t (image column x0 - 2 + XJ r + t), channels 8g .. 8g+7 (piece g):
  // XJ = 2: its pair is r + t / 2 — the lane's base plus a constant
  int aoffs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) aoffs[t] = poff(XJ * r + t, g);
  auto aoff = [&](int t) { return aoffs[t]; };
  // output staging: lane (column nn < 3 XJ, g) drops accumulator row m = 4g + e (pixels XJ m + j) at float m * 3 XJ + nn;
  // other lanes into their dump slot
  const int wofs = r < 3 * XJ ? (4 * g * 3 * XJ + r) : NO + lane;
  const int wstep = r < 3 * XJ ? 3 * XJ : 0;
  const unsigned yofs = lane < NLN ? (unsigned)((x0 * 3 + lane * VW) * 4) : OOB;

  // the five output rows in flight are five NAMED tiles handed to a step by reference in rotated order (an accumulator
  // ARRAY indexed by the step would become a runtime-indexed private array, i.e. scratch memory)
  f32x4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0, T3 = T0, T4 = T0;

  // row s: registers -> (norm) -> wave-private LDS buffer s & 1, then its A fragments
  bf16x8 af[2][NT];
  auto stage = [&](auto p_c, int s) {
    constexpr int P = decltype(p_c)::value;
    const int yy = ya - 2 + s;
    const bool rok = (unsigned)yy < (unsigned)H && s < cnt;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      u32x4 v = rg[P][k];
      if constexpr (NORM) {
        const u32x4 t = lg_norm8(v, mu, mul, na, nb, ni.alpha);
        const bool ok = rok && goff[k] != OOB;  // padding stays zero (it is h that is zero-padded, not z)
        v = u32x4{ok ? t[0] : 0u, ok ? t[1] : 0u, ok ? t[2] : 0u, ok ? t[3] : 0u};
      }
      *reinterpret_cast<u32x4*>(my + loff[k] + lbuf[k] * P) = v;
    }
    fetch(s + 2, rg[P]);
#pragma unroll
    for (int t = 0; t < NT; ++t) af[P][t] = *reinterpret_cast<const bf16x8*>(my + P * ROWB + aoff(t));
  };
  // input row s (yy = ya - 2 + s) adds to the output rows yy - 2 + ky: O0 (ky = 0) is completed by it, O4 (ky = 4) starts.
  // Row s + 1 is staged and its fragments requested BEFORE the MFMAs of row s, so the LDS round trip hides under them.
  auto step = [&](auto p_c, int s, f32x4& O0, f32x4& O1, f32x4& O2, f32x4& O3, f32x4& O4) {
    constexpr int P = decltype(p_c)::value;
    stage(std::integral_constant<int, P ^ 1>{}, s + 1);
    O4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // five independent accumulation chains
      O0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[P][t], bw[0][t], O0, 0, 0, 0);
      O1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[P][t], bw[1][t], O1, 0, 0, 0);
      O2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[P][t], bw[2][t], O2, 0, 0, 0);
      O3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[P][t], bw[3][t], O3, 0, 0, 0);
      O4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[P][t], bw[4][t], O4, 0, 0, 0);
    }
    // output row ya + s - 4 is complete (when 4 <= s < cnt; otherwise the store offset is out of range and dropped):
    // C layout row (= pixel group m) 4g + e, column r (= 3 j + co for r < 3 XJ)
#pragma unroll
    for (int e = 0; e < 4; ++e) mo[wofs + e * wstep] = O0[e];
    {
      // tanh(v) = 1 - 2 / (1 + e^{2v}): |error| < 2e-7 absolute over the whole range (e^{2v} -> inf / 0 saturate to +-1)
      const unsigned yo = sel(s >= 4 && s < cnt, (unsigned)((ya + s - 4) * W * 12), OOB);
      if constexpr (VW == 1) {
        const float v = mo[lane] + bl[0];
        const float ex = __builtin_amdgcn_exp2f(v * 2.885390082f);
        const float o = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + ex);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), yrsrc, yofs + yo, 0, 0);
      } else {
        const f32x2 v2 = *reinterpret_cast<const f32x2*>(mo + 2 * lane);   // (lanes >= 48 read dump slots; their store is out of range)
        u32x2 o2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float ex = __builtin_amdgcn_exp2f((v2[k] + bl[k]) * 2.885390082f);
          o2[k] = __builtin_bit_cast(unsigned, 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + ex));
        }
        __builtin_amdgcn_raw_buffer_store_b64(o2, yrsrc, yofs + yo, 0, 0);
      }
    }
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  fetch(0, rg[0]);
  fetch(1, rg[1]);
  stage(P0{}, 0);
  for (int s0 = 0; s0 < cnt; s0 += 10) {  // whole groups of ten; steps past cnt fetch nothing and store nothing
    step(P0{}, s0 + 0, T0, T1, T2, T3, T4);
    step(P1{}, s0 + 1, T1, T2, T3, T4, T0);
    step(P0{}, s0 + 2, T2, T3, T4, T0, T1);
    step(P1{}, s0 + 3, T3, T4, T0, T1, T2);
    step(P0{}, s0 + 4, T4, T0, T1, T2, T3);
    step(P1{}, s0 + 5, T0, T1, T2, T3, T4);
    step(P0{}, s0 + 6, T1, T2, T3, T4, T0);
    step(P1{}, s0 + 7, T2, T3, T4, T0, T1);
    step(P0{}, s0 + 8, T3, T4, T0, T1, T2);
    step(P1{}, s0 + 9, T4, T0, T1, T2, T3);
  }
}

}  // namespace

extern "C" int lg_n3_rows_supported(int H, int W, int C) { return (W % 16 == 0 && H >= 1 && C == 32) ? 1 : 0; }

// x16: the bf16 input h [B,H,W,C]; or, with stats != null, the raw bf16 conv output z of the level below, normalised +
// LeakyReLU(alpha)'d on the fly from its statistics records [B][8] (norm.hip)
extern "C" int lg_n3_s1t_fwd_rows_try(const void* x16, const float* stats, float alpha, const float* w, const float* bias, float* y,
                                      int B, int H, int W, int C, void* stream) {
  static int off = -1;
  if (off < 0) off = lg_env_flag("LG_NO_ROWS") ? 1 : 0;
  if (off || !lg_n3_rows_supported(H, W, C) || !x16 || !w || !bias || !y) return LG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int RB = H <= 64 ? H : 64;
  const RowsNormIn ni{stats, alpha};
  const __bf16* x = (const __bf16*)x16;
  static int no2 = -1;
  if (no2 < 0) no2 = lg_env_flag("LG_ROWS_XJ1") ? 1 : 0;   // A/B switch: the one-pixel-per-row form everywhere
  const bool xj2 = W % 32 == 0 && !no2;
  const int sw = xj2 ? 32 : 16;
  const dim3 grid(B * ((W + 4 * sw - 1) / (4 * sw)) * ((H + RB - 1) / RB));
  if (xj2) {
    if (stats) hipLaunchKernelGGL((s1t_fwd_rows_kernel<32, true, 2>), grid, dim3(256), 0, st, x, w, bias, y, B, H, W, RB, ni);
    else hipLaunchKernelGGL((s1t_fwd_rows_kernel<32, false, 2>), grid, dim3(256), 0, st, x, w, bias, y, B, H, W, RB, ni);
  } else {
    if (stats) hipLaunchKernelGGL((s1t_fwd_rows_kernel<32, true, 1>), grid, dim3(256), 0, st, x, w, bias, y, B, H, W, RB, ni);
    else hipLaunchKernelGGL((s1t_fwd_rows_kernel<32, false, 1>), grid, dim3(256), 0, st, x, w, bias, y, B, H, W, RB, ni);
  }
  LG_CHECK_LAUNCH("lg_n3_s1t_fwd_rows");
  lg_note_kernel(stats ? "s1t_fwd_rows_kernel<32,NORM>" : "s1t_fwd_rows_kernel<32>");
  return LG_OK;
}
