// All-taps weight gradient of the 5x5 stride-2 layers on the EXACT-f32 path (v_mfma_f32_32x32x2_f32), the f32 twin of wgrad_at.hip:
//   dW[ky,kx,cb,cs] = sum_{n,y,x} big[n, 2y+ky-1, 2x+kx-1, cb] * small[n, y, x, cs]       (wgrad_igemm.hip:1-5)
// The per-tap kernel of wgrad_igemm.hip stages an operand pair per tap and pulls `big` through L2 25 times; it runs at 57 % of the
// fp32 matrix peak (profiles/r3_bench_c2.json).  Here, as in wgrad_at.hip, one block (8 waves, one per CU) owns ALL 25 taps of a
// 32 (cb) x 64 (cs) slice of dW — 50 accumulator tiles of 32x32 — and walks a list of items (sample, band of R small rows, strip of
// SW small columns = 64 small pixels):
//   * LDS holds the (2R+3) x (2SW+3) halo of `big` (its 32 channels, fp32) and the 64 pixels of `small` (64 channels), double
//     buffered, staged global -> registers -> LDS behind the MFMAs of the previous item;
//   * `big` pixels are stored de-interleaved by x parity, so a tap is a byte offset and the two pixels of a k step are neighbouring
//     128-B entries.  No transposed read is needed at 4 bytes per element: lane (r, h) of a fragment reads channel r of pixel h
//     (A) / column r of pixel h (B) — 32 consecutive floats per half wave, conflict-free ds_read_b32;
//   * a k step is TWO small pixels (K = 2), an item has 32 of them.  Wave w owns taps 3w..3w+2 (both cs halves: 6 tiles, A fragment
//     read once, used twice); tap 24 rotates — wave w runs it on the k steps with ks % 8 == w and the 8 partial tiles are summed
//     through LDS at the end — so every wave issues exactly 200 MFMAs per item (64 cycles each: the loop is MFMA-bound by
//     construction, 5 LDS reads per 384 matrix cycles);
//   * output: slab[split][tap][cb][cs] (fp32), reduced in fixed order by slab_reduce4_kernel (deterministic, no atomics).
#include <stdlib.h>
#include "lg_common.h"

namespace {

struct Wg32Params {
  const float* big;    // [B, 2Hm, 2Wm, Cb]
  const float* small;  // [B, Hm, Wm, Cs]
  float* slab;         // [nsplit][25][Cb][Cs]
  int B, Hm, Wm, Cb, Cs;
  int nuj, nunits;     // units = (Cb/32) x (Cs/64)
  int items_total, items_per;
};

template <int SW, int R>
struct At32Cfg {
  static_assert(R * SW == 64, "64 small pixels = 32 k steps per item");
  static constexpr int NA = SW + 2;               // 128-B entries per x-parity array
  static constexpr int EVEN_OFF = NA * 128;       // odd-x array first (halo column 0 is big x = 2 X0 - 1)
  static constexpr int ROWP = 2 * NA * 128;       // bytes per big row
  static constexpr int NBR = 2 * R + 3, NPX = 2 * SW + 3;
  static constexpr int BIG = NBR * ROWP, SMALL = R * SW * 256, BUF = BIG + SMALL;
  static constexpr int NBP = NBR * NPX * 8, NBL = (NBP + 511) / 512;   // 16-B pieces of the halo image, per thread
  static constexpr int NSP = R * SW * 16, NSL = NSP / 512;
  static_assert(BUF < 65536 + 4096, "k-step offsets stay near the ds offset field");
  static_assert(2 * BUF <= 160 * 1024 && 2 * BUF >= 8 * 2 * 4096, "LDS budget; the tap-24 reduction reuses it");
};

template <int SW, int R>
__global__ __launch_bounds__(512) void wgrad_at32_kernel(const Wg32Params p) {
  using C = At32Cfg<SW, R>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = lg_xcd_remap(blockIdx.x, gridDim.x);
  const int unit = bx % p.nunits, split = bx / p.nunits;   // units of one split run together: they share the pixels
  const int i0 = (unit / p.nuj) * 32, j0 = (unit % p.nuj) * 64;
  const int it0 = split * p.items_per, it1 = min(it0 + p.items_per, p.items_total);
  const int nxs = p.Wm / SW, nyb = p.Hm / R;
  const int Hb = 2 * p.Hm, Wb = 2 * p.Wm;

  u32x4 rbig[C::NBL], rsm[C::NSL];
  auto load_item = [&](int it) __attribute__((always_inline)) {
    const int xs = it % nxs, t2 = it / nxs, yb = t2 % nyb, n = t2 / nyb;
    const int gy0 = 2 * yb * R - 1, gx0 = 2 * xs * SW - 1;
    const float* bbase = p.big + (long long)n * Hb * Wb * p.Cb + i0;
    const float* sbase = p.small + ((long long)(n * p.Hm + yb * R) * p.Wm + xs * SW) * p.Cs + j0;
#pragma unroll
    for (int k = 0; k < C::NBL; ++k) {
      const int q = tid + k * 512, piece = q & 7, pp = q >> 3;
      const int row = pp / C::NPX, px = pp - row * C::NPX;
      const int gy = gy0 + row, gx = gx0 + px;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < C::NBP && (unsigned)gy < (unsigned)Hb && (unsigned)gx < (unsigned)Wb)
        v = *reinterpret_cast<const u32x4*>(bbase + (long long)(gy * Wb + gx) * p.Cb + piece * 4);
      rbig[k] = v;
    }
#pragma unroll
    for (int k = 0; k < C::NSL; ++k) {
      const int q = tid + k * 512, c4 = q & 15, pp = q >> 4;
      const int yy = pp / SW, px = pp % SW;
      rsm[k] = *reinterpret_cast<const u32x4*>(sbase + (long long)(yy * p.Wm + px) * p.Cs + c4 * 4);
    }
  };
  auto store_item = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < C::NBL; ++k) {
      const int q = tid + k * 512, piece = q & 7, pp = q >> 3;
      const int row = pp / C::NPX, px = pp - row * C::NPX;
      // px 0 <-> big x = 2*X0 - 1 (odd, entry 0); px 1 <-> 2*X0 (even, entry 0); ...
      if (q < C::NBP) *reinterpret_cast<u32x4*>(buf + row * C::ROWP + ((px & 1) ? C::EVEN_OFF : 0) + (px >> 1) * 128 + piece * 16) = rbig[k];
    }
#pragma unroll
    for (int k = 0; k < C::NSL; ++k) {
      const int q = tid + k * 512, c4 = q & 15, pp = q >> 4;   // pp = yy * SW + px: the item's 64 small pixels in row-major order
      *reinterpret_cast<u32x4*>(buf + C::BIG + pp * 256 + c4 * 16) = rsm[k];
    }
  };

  // fragment addressing: lane (r, h) = channel / column r of the k step's pixel h
  const int r = lane & 31, hh = lane >> 5;
  const int lpA = hh * 128 + r * 4, lpB = C::BIG + hh * 256 + r * 4;
  int ab[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = 3 * wid + i, ky = t / 5, kx = t - ky * 5;
    // tap (ky,kx) at small pixel (yy, x): big row 2yy+ky of the halo; halo column 2x+kx -> parity kx & 1, entry x + (kx>>1)
    ab[i] = lpA + ky * C::ROWP + ((kx & 1) ? C::EVEN_OFF : 0) + (kx >> 1) * 128;
  }
  const int a24 = lpA + 4 * C::ROWP + 2 * 128;

  f32x16 acc[3][2], acc24[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i][0][e] = 0.f; acc[i][1][e] = 0.f; }
    acc24[0][e] = 0.f; acc24[1][e] = 0.f;
  }

  if (it0 < it1) {
    load_item(it0);
    store_item(smem);
  }
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    const int cur = (it - it0) & 1;
    if (it + 1 < it1) load_item(it + 1);
    const char* sb = smem + cur * C::BUF;
    // k step ks = small pixels (yy, x), (yy, x + 1): compile-time offsets once the loop is unrolled
    auto koffA = [](int ks) { constexpr int PR = SW / 2; return 2 * (ks / PR) * C::ROWP + 2 * (ks % PR) * 128; };
    auto koffB = [](int ks, int h) { constexpr int PR = SW / 2; return ((ks / PR) * SW + 2 * (ks % PR)) * 256 + h * 128; };
    auto rdf = [&](int off) { return *reinterpret_cast<const float*>(sb + off); };
    float a[3], b[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = rdf(ab[i] + koffA(0));
#pragma unroll
    for (int h = 0; h < 2; ++h) b[h] = rdf(lpB + koffB(0, h));
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      float an[3] = {a[0], a[1], a[2]}, bn[2] = {b[0], b[1]};
      if (ks + 1 < 32) {   // the next step's five operands are requested under this step's MFMAs
#pragma unroll
        for (int i = 0; i < 3; ++i) an[i] = rdf(ab[i] + koffA(ks + 1));
#pragma unroll
        for (int h = 0; h < 2; ++h) bn[h] = rdf(lpB + koffB(ks + 1, h));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[h], acc[i][h], 0, 0, 0);
      if (wid == (ks & 7)) {
        const float a3 = rdf(a24 + koffA(ks));
#pragma unroll
        for (int h = 0; h < 2; ++h) acc24[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b[h], acc24[h], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = an[i];
      b[0] = bn[0]; b[1] = bn[1];
    }
    if (it + 1 < it1) store_item(smem + (cur ^ 1) * C::BUF);
    __syncthreads();
  }

  // slab[split][t][cb][cs]: accumulator register e of lane (r, hh) = row (e&3) + 8*(e>>2) + 4hh, column r
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
inline int at32_cus() { return lg_device_cus(); }   // the split plan (= summation order) depends on the device alone, see wgrad_at.hip

// 0: not applicable, else the strip width (16: 16 x 4 strips; 8: whole 8-column maps, 8 rows)
inline int at32_shape(int Hm, int Wm, int cb, int cs) {
  if (cb % 32 || cs % 64 || cb < 32 || cs < 64) return 0;
  if ((cb / 32) * (cs / 64) > 64) return 0;
  if (Wm % 16 == 0 && Hm % 4 == 0) return 16;
  if (Wm == 8 && Hm % 8 == 0) return 8;
  return 0;
}

inline void at32_plan(int B, int Hm, int Wm, int cb, int cs, int sw, int* nsplit, int* items_total, int* items_per) {
  const int R = 64 / sw;
  const int nunits = (cb / 32) * (cs / 64);
  *items_total = B * (Hm / R) * (Wm / sw);
  int ns = at32_cus() / nunits;
  if (ns < 1) ns = 1;
  if (ns > *items_total) ns = *items_total;
  *items_per = lg_cdiv(*items_total, ns);
  *nsplit = lg_cdiv(*items_total, *items_per);
}

}  // namespace

extern "C" size_t lg_wgrad_at32_workspace_bytes(int B, int Hm, int Wm, int cb, int cs) {
  const int sw = at32_shape(Hm, Wm, cb, cs);
  if (!sw) return 0;
  int ns, tot, per;
  at32_plan(B, Hm, Wm, cb, cs, sw, &ns, &tot, &per);
  return (size_t)ns * 25 * cb * cs * sizeof(float);
}

// writes slab[nsplit][25][cb][cs] into `workspace` and *nsplit_out; the caller reduces the slabs (lg_conv_wgrad_m16)
extern "C" int lg_wgrad_at32_try(const float* big, const float* small, void* workspace, size_t ws_bytes, int B, int Hm, int Wm,
                                 int cb, int cs, int* nsplit_out, void* stream) {
  static int off = -1;
  if (off < 0) off = lg_env_flag("LG_NO_WGAT32") ? 1 : 0;
  const int sw = at32_shape(Hm, Wm, cb, cs);
  if (off || !sw || !big || !small || !nsplit_out) return LG_ERR_UNSUPPORTED;
  if ((long long)B * 4 * Hm * Wm * cb >= (1ll << 31) || (long long)B * Hm * Wm * cs >= (1ll << 31)) return LG_ERR_UNSUPPORTED;
  Wg32Params p{};
  p.big = big; p.small = small; p.slab = (float*)workspace;
  p.B = B; p.Hm = Hm; p.Wm = Wm; p.Cb = cb; p.Cs = cs;
  p.nuj = cs / 64; p.nunits = (cb / 32) * (cs / 64);
  int ns;
  at32_plan(B, Hm, Wm, cb, cs, sw, &ns, &p.items_total, &p.items_per);
  LG_CHECK_ARG(ws_bytes >= (size_t)ns * 25 * cb * cs * sizeof(float), "lg_wgrad_at32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_at32_kernel<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * At32Cfg<16, 4>::BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_at32_kernel<8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * At32Cfg<8, 8>::BUF);
    attr = true;
  }
  constexpr int LDS16 = 2 * At32Cfg<16, 4>::BUF, LDS8 = 2 * At32Cfg<8, 8>::BUF;
  if (sw == 16) hipLaunchKernelGGL((wgrad_at32_kernel<16, 4>), dim3(p.nunits * ns), dim3(512), LDS16, st, p);
  else hipLaunchKernelGGL((wgrad_at32_kernel<8, 8>), dim3(p.nunits * ns), dim3(512), LDS8, st, p);
  LG_CHECK_LAUNCH("lg_wgrad_at32");
  lg_note_kernel(sw == 16 ? "wgrad_at32_kernel<16,4>" : "wgrad_at32_kernel<8,8>");
  *nsplit_out = ns;
  return LG_OK;
}
