// Software-pipelined stride-2 "up" contraction for gfx950 (bf16 operands, fp32 accumulation), WIDE layers (N % 128 == 0):
//   out[n][2y+py][2x+px][co] = bias[co] + sum_{taps (ky,kx) of parity class (py,px)} sum_ci src[n][y+dy][x+dx][ci] W[ky][kx][co][ci]
// = Conv2DTranspose(f, 5, 2, same) forward (model.py:39-40) and the data gradient of Conv2D(f, 5, 2, same) (model.py:15)
// for convT2 forward / conv3 data gradient (16 x 16 x 256 -> 32 x 32 x 128), which conv_halo.hip ran at 39 % MFMA busy.
//
// Same machine as conv_down3.hip (persistent blocks, 2 per CU, 4 waves of 32 channels x 128 pixels; source halo in LDS in
// channel slices, double buffered global -> registers -> LDS behind the MFMAs; weight fragments through a register ring with
// scalar addressing; transposed product; epilogue through the idle halo buffer with fused moments / norm-backward sums), with
// what the up form changes:
//   * an item is (8 x 16 source tile, PARITY CLASS, 128-column tile): the four classes of a tile are four items — a class
//     is a stride-1 contraction over the 10 x 18 halo with 4 / 6 / 6 / 9 taps, and with N >= 128 an output pixel is a
//     256-B run by itself, so the classes need no common staging area (conv_up3.hip needs one for N = 32 / 64);
//   * a BLOCK is bound to one class for its whole life (the grid is split 9 : 6 : 6 : 4 between the classes, the tap counts):
//     the slice body then exists ONCE inside each persistent loop.  (Switching the class per item put four to eight
//     instantiations of the body inside one loop: 256 VGPRs + ~100 spills, among them the per-lane LDS bases, and every
//     reload in the tap loop was a scratch_load + s_waitcnt vmcnt(0) that also drained the weight ring — 162 us against
//     conv_halo's 137; one instantiation per loop needs ~200 VGPRs.)
//   * a slice is 64 channels (4 k-steps per tap): 16 .. 36 fragments = 64 .. 144 MFMAs per wave between two barriers;
//   * the ring holds 8 fragments (12 for the 9-tap class, whose 36 fragments per slice are not a multiple of 8), so the ring
//     position of a fragment is a compile-time register.
#include <stdlib.h>
#include <type_traits>
#include "lg_common.h"

namespace {

constexpr int TH = 8, TW = 16, HHT = TH + 2, HWT = TW + 2, NPX = HHT * HWT;  // 10 x 18 = 180 halo pixels
constexpr int KC = 64, KS = KC / 16;          // channels per slice, k-steps per tap and slice
constexpr int PITCH = KC * 2 + 16;            // 144 B per halo pixel: 16 consecutive pixels hit 16 distinct 16-B slots
constexpr int HB = 32768;                     // buffer size: the halo slice (25920 B) or the 128 x 256-B output tile
// PAIR (8 x 8 source maps, even batch): a tile = TWO samples side by side (class-pixel columns 0..7 = sample n, 8..15 = sample n + 1),
// each with its own 10 x 10 halo.  Both halos use a 10-slot row pitch (tap offsets stay compile-time constants) and sample 1's
// image starts 104 slots behind sample 0's: a ds_read_b128 lane group is one tile row = 8 pixels of each sample, and with a
// 144-B pitch the 16 slots of 8 + 8 consecutive pixels are distinct exactly when the two runs are 8 (mod 16) slots apart
// (100 slots of sample 0 + 4 spare; side by side in one 20-slot row, pixels p and p + 16 would share their banks).
constexpr int PHW = 10, PNPX = 100, PS1 = 104;
template <bool PAIR> struct U4L {
  static constexpr int HWP = PAIR ? PHW : HWT;                       // halo row pitch in pixel slots
  static constexpr int NPIECE = (PAIR ? 2 * PNPX : NPX) * (KC / 8);  // 16-B pieces per slice (1600 | 1440)
  static constexpr int PPT = (NPIECE + 255) / 256;                   // 7 | 6
};
static_assert((PS1 + PNPX) * PITCH <= HB, "pair halo fits the buffer");
constexpr int SRED_OFF = 2 * HB, SBIAS_OFF = 2 * HB + 256, LDS_BYTES = 2 * HB + 256 + 512;

struct U4Params {
  const __bf16* src;   // [B][Hs][Ws][Cs]
  const char* wp;      // "up" pack [25][N/32][Cs/16][64][16 B]
  const float* bias;   // [N] or null
  __bf16* out;         // [B][2Hs][2Ws][N]
  double* spart;       // [B][nparts][3] or null
  int B, Hs, Ws, Cs, N, N32, KB;
  int tpi_x, tpi, ntn, nper, nparts;   // nper = items per class
  int gend[4];         // block ranges: class rank rk (taps 9, 6, 6, 4) owns blocks [gend[rk-1], gend[rk])
  int xcdmajor;        // class-pair mode: XCD-major block ranks inside a type (LG_U4_XCD; see the kernel)
  int mfast, ntm;      // mfast: item -> (row tile fastest, column tile); ntm = row tiles (LG_U4_MFAST, with xcdmajor: an XCD's blocks share ONE column tile's weights)
  int pairmode;        // 1 (default): blocks [0, gend[0]) run classes 3 then 0, blocks [gend[0], gend[1]) classes 1 then 2 (see the kernel)
  LgNormFuse nf;
};

__device__ __forceinline__ int pix32(int r) {  // MFMA column -> tile pixel inside its 32-pixel group (conv_halo.hip)
  const int q = r >> 2, lo = r & 3;
  const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;
  const int rank = odd ? ((q == 1) ? 0 : (q == 2) ? 1 : (q == 4) ? 2 : 3) : ((q == 0) ? 0 : (q == 3) ? 1 : (q == 5) ? 2 : 3);
  return odd * 16 + rank * 4 + lo;
}

// class = 2*py + px; taps a < nk(py), b < nk(px):  ky = py ? 2a : 2a+1, dy = (py + 1 - ky) / 2  (same in x)   (conv_up3.hip)
constexpr int nk(int p) { return p ? 3 : 2; }
constexpr int ntaps_of(int cls) { return nk(cls >> 1) * nk(cls & 1); }
constexpr int tap_k(int p, int a) { return p ? 2 * a : 2 * a + 1; }
constexpr int tap_d(int p, int a) { return (p + 1 - tap_k(p, a)) / 2; }
constexpr int tap_widx(int cls, int t) { return tap_k(cls >> 1, t / nk(cls & 1)) * 5 + tap_k(cls & 1, t % nk(cls & 1)); }
constexpr int tap_aoff(int cls, int t, int hwp) {  // LDS byte offset of the tap's source pixel relative to the lane's base
  return ((tap_d(cls >> 1, t / nk(cls & 1)) + 1) * hwp + tap_d(cls & 1, t % nk(cls & 1)) + 1) * PITCH;
}
template <int CLS, bool STATS, bool FUSE, bool PAIR>
__device__ __forceinline__ void up4_run(const U4Params& p, char* smem, int lb, int G) {
  constexpr int HWP = U4L<PAIR>::HWP, NPIECE = U4L<PAIR>::NPIECE, PPT = U4L<PAIR>::PPT;
  constexpr int F = ntaps_of(CLS) * KS;            // fragments per slice (16 | 24 | 36)
  constexpr int RING = F % 8 == 0 ? 8 : 12;
  double* sred = reinterpret_cast<double*>(smem + SRED_OFF);
  float* sbias = reinterpret_cast<float*>(smem + SBIAS_OFF);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nmine = (p.nper - lb + G - 1) / G;   // items lb, lb + G, ... of this class
  const int nchunk = p.Cs / KC;
  const int total = nmine * nchunk;

  // ---- the halo pieces this thread stages in every slice --------------------------------------------------------------
  int pl[PPT], pyx[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int q = tid + u * 256;
    pl[u] = -1; pyx[u] = 0;
    if (q < NPIECE) {
      const int px = q >> 3, pc = q & 7;
      if constexpr (PAIR) {   // pieces 0..799: sample n, 800..1599: sample n + 1 (bit 7 of the x byte)
        const int sl = px >= PNPX, pp = px - PNPX * sl, hy = pp / PHW, hx = pp - hy * PHW;
        pl[u] = (pp + PS1 * sl) * PITCH + pc * 16;
        pyx[u] = (pc << 16) | (hy << 8) | (sl << 7) | hx;
      } else {
        const int hy = px / HWT, hx = px - hy * HWT;
        pl[u] = px * PITCH + pc * 16;
        pyx[u] = (pc << 16) | (hy << 8) | hx;
      }
    }
  }
  int abase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = i * 32 + pix32(r);
    const int col = m & 15;
    abase[i] = ((m >> 4) * HWP + (PAIR ? (col & 7) + PS1 * (col >> 3) : col)) * PITCH + h * 16;
  }

  unsigned long long wzero = 0;  // opaque zero, re-read per slice (see wfrag below)
  struct Item { int n, y0, x0, tn; };
  auto decode = [&](int k) {
    const int rest = lb + k * G;
    Item it;
    it.tn = p.mfast ? rest / p.ntm : rest % p.ntn;
    const int tm = p.mfast ? rest - it.tn * p.ntm : rest / p.ntn;
    if constexpr (PAIR) { it.n = 2 * tm; it.y0 = 0; it.x0 = 0; return it; }
    it.n = tm / p.tpi;
    const int tt = tm - it.n * p.tpi;
    it.y0 = (tt / p.tpi_x) * TH; it.x0 = (tt % p.tpi_x) * TW;
    return it;
  };
  // RAW BUFFER loads from a descriptor over the item's sample: a piece outside the image (SAME padding) or beyond the thread's
  // share gets an out-of-range offset and reads as zeros — no branch around any load, so hipcc keeps an exact count of the loads
  // in flight (conv_down3.hip: with `if (inside) v = load` it fell back to short vmcnt waits that exposed the halo's HBM latency)
  constexpr unsigned OOB = 0x80000000u;   // >= num_records for every supported shape (checked on the host)
  const int sample_elems = p.Hs * p.Ws * p.Cs;
  auto issue = [&](const Item& it, int c0, u32x4 (&v)[PPT]) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.src + (long long)it.n * sample_elems), 0,
                                                                        (PAIR ? 2 : 1) * sample_elems * 2, 0x00027000);
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int pq = pyx[u] + (int)wzero;  // (re-derived per slice: hoisted out of the slice loop these offsets spill)
      const int sy = it.y0 - 1 + ((pq >> 8) & 255), sx = it.x0 - 1 + (pq & (PAIR ? 127 : 255));
      const int sl = PAIR ? (pq >> 7) & 1 : 0;
      const bool in = pl[u] >= 0 && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      const unsigned off = in ? (unsigned)((((sl * p.Hs + sy) * p.Ws + sx) * p.Cs + c0 + (pq >> 16) * 8) * 2) : OOB;
      v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
  auto commit = [&](char* buf, const u32x4 (&v)[PPT]) {
#pragma unroll
    for (int u = 0; u < PPT; ++u)
      if (pl[u] >= 0) *reinterpret_cast<u32x4*>(buf + pl[u]) = v[u];
  };
  // weight fragment (tap widx, this wave's 32 columns of column tile tn, k-step kb): uniform base + 32-bit lane offset
  const unsigned lane16 = lane * 16;
  const unsigned long long wstride = (unsigned long long)p.N32 * p.KB * 1024ull;  // bytes between consecutive pack taps
  auto wbase = [&](int tn, int kb) { return p.wp + ((long long)(tn * 4 + wid) * p.KB + kb) * 1024; };
  // (an opaque zero re-read per slice keeps the fragment addresses of the eight slice bodies out of loop-invariant VGPR
  //  pairs — hoisted, they spill; the pointer itself must not pass through the asm or the loads become flat_load)
  auto wfrag = [&](const char* base, int widx, int ks) {
    return *reinterpret_cast<const u32x4*>((base + ((wzero + (unsigned long long)widx) * wstride + ks * 1024)) + lane16);
  };

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 bf[RING];
  u32x4 hv[PPT];

  // ---- prologue ---------------------------------------------------------------------------------------------------------
  Item cur = decode(0);
  issue(cur, 0, hv);
  {
    const char* w0 = wbase(cur.tn, 0);
#pragma unroll
    for (int f = 0; f < RING; ++f) bf[f] = wfrag(w0, tap_widx(CLS, f / KS), f % KS);
  }
  commit(smem, hv);
  auto load_bias = [&](int tn) {
    if (tid < 128) sbias[tid] = p.bias ? p.bias[tn * 128 + tid] : 0.f;
  };
  load_bias(cur.tn);
  __syncthreads();

  int k = 0, c = 0;  // item index in this block's list, slice index
  Item nxt = cur;
  int c2 = 0;
  auto slice = [&](int s) {
    constexpr int OFF = 0;  // F % RING == 0: fragment f always sits in ring slot f % RING
    const char* hbuf = smem + (s & 1) * HB;
    const char* wcur = wbase(cur.tn, c * KS);
    const char* wnxt = wbase(nxt.tn, c2 * KS);
    // order pinned as in conv_down3.hip: A fragments of step f+1 requested, 4 MFMAs of step f, ring slot refilled
    bf16x8 a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(hbuf + abase[i] + tap_aoff(CLS, 0, HWP));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < F; ++f) {
      if (f + 1 < F) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a[(f + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(hbuf + abase[i] + tap_aoff(CLS, (f + 1 < F ? f + 1 : 0) / KS, HWP) + ((f + 1) % KS) * 32);
      }
      const int slot = (f + OFF) % RING;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[slot]), a[f & 1][i], acc[i], 0, 0, 0);
      if (f + RING < F) bf[slot] = wfrag(wcur, tap_widx(CLS, (f + RING < F ? f + RING : 0) / KS), (f + RING) % KS);
      else bf[slot] = wfrag(wnxt, tap_widx(CLS, (f + RING >= F ? f + RING - F : 0) / KS), (f + RING - F) % KS);
      // one MFMA, one LDS read in its shadow, ..., the ring refill behind the last MFMA (conv_down3.hip)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    commit(smem + ((s + 1) & 1) * HB, hv);
    __syncthreads();  // slice s consumed by every wave, slice s+1 complete
  };
  auto epilogue = [&](int s, bool more) {
      // ---- epilogue of item `cur`: acc[i][e] = channel (e&3) + 8*(e>>2) + 4*h of class pixel i*32 + pix32(r) -------------
      char* C = smem + (s & 1) * HB;  // the buffer slice s ran out of is idle until the commit of step s+1
      const float shift = STATS ? sbias[0] : 0.f;
      const f32x2 shift2 = {shift, shift};
      f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
      f32x4 bq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(sbias + wid * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + pix32(r);
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
          *reinterpret_cast<bf16x4*>(C + row * 256 + (((wid * 4 + g) ^ (row & 15)) << 4) + 8 * h) = w;
        }
      }
      if constexpr (STATS) {
        const double l1 = (double)s1v[0] + (double)s1v[1], l2 = (double)s2v[0] + (double)s2v[1];
        if constexpr (PAIR) {  // a lane's 64 values all belong to ONE sample (its tile column never changes): masked sums
          const bool sb = (pix32(r) & 8) != 0;
          const double a1 = lg_wave_sum_d(sb ? 0.0 : l1), a2 = lg_wave_sum_d(sb ? 0.0 : l2);
          const double b1 = lg_wave_sum_d(sb ? l1 : 0.0), b2 = lg_wave_sum_d(sb ? l2 : 0.0);
          if (lane == 0) {
            double* q = sred + (k & 1) * 16;
            q[wid] = a1; q[4 + wid] = a2; q[8 + wid] = b1; q[12 + wid] = b2;
          }
        } else {
          const double w1 = lg_wave_sum_d(l1), w2 = lg_wave_sum_d(l2);
          if (lane == 0) { sred[(k & 1) * 16 + wid] = w1; sred[(k & 1) * 16 + 4 + wid] = w2; }
        }
      }
      // element offset of class pixel `row` (= 16 my + mx): output pixel (2 (y0 + my) + py, 2 (x0 + mx) + px); PAIR: columns
      // 8..15 are sample n + 1
      constexpr int py = CLS >> 1, pxc = CLS & 1;
      const long long obase = ((long long)(cur.n * 2 * p.Hs + 2 * cur.y0 + py) * (2 * p.Ws) + 2 * cur.x0 + pxc) * p.N + cur.tn * 128;
      const long long sample_out = (long long)4 * p.Hs * p.Ws * p.N;
      auto pix_off = [&](int row) -> long long {
        if constexpr (PAIR) return obase + ((row >> 3) & 1) * sample_out + ((long long)(2 * (row >> 4)) * (2 * p.Ws) + 2 * (row & 7)) * p.N;
        else return obase + ((long long)(2 * (row >> 4)) * (2 * p.Ws) + 2 * (row & 15)) * p.N;
      };
      u32x4 zq[FUSE ? 8 : 1];
      (void)zq;
      if constexpr (FUSE) {  // requested now (the accumulators are dead): the loads land behind the barrier
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
          const int piece = tid + q8 * 256 + (int)wzero, row = piece >> 4, j = piece & 15;
          zq[q8] = *reinterpret_cast<const u32x4*>(p.nf.z + pix_off(row) + j * 8);
        }
      }
      __syncthreads();  // tile complete in LDS (and the wave sums)
      float nf1 = 0.f, nf2 = 0.f;
      (void)nf1; (void)nf2;
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) {
        const int piece = tid + q8 * 256 + (int)wzero, row = piece >> 4, j = piece & 15;
        const u32x4 v = *reinterpret_cast<const u32x4*>(C + row * 256 + ((j ^ (row & 15)) << 4));
        *reinterpret_cast<u32x4*>(p.out + pix_off(row) + j * 8) = v;
        if constexpr (FUSE) {
          // PAIR: the tile column of a thread's pieces is fixed ((tid >> 4) & 15): waves 0, 1 sweep sample n, waves 2, 3 n + 1
          const lg_const_f32p sp = lg_as_const(p.nf.stats + (long long)(cur.n + (PAIR ? wid >> 1 : 0)) * 8);   // scalar loads (lg_common.h)
          lg_nf_accum(v, zq[q8], lg_uniform(sp[0]), lg_uniform(sp[4]), lg_uniform(sp[2]), lg_uniform(sp[3]), p.nf.alpha, nf1, nf2);   // (cur.n, wid: wave-uniform)
        }
      }
      if constexpr (FUSE) {
        const double w1 = lg_wave_sum_d((double)nf1), w2 = lg_wave_sum_d((double)nf2);
        if (lane == 0) { sred[(k & 1) * 16 + wid] = w1; sred[(k & 1) * 16 + 4 + wid] = w2; }
      }
      const int tin = PAIR ? 0 : (cur.y0 / TH) * p.tpi_x + cur.x0 / TW;
      const long long prec = (long long)cur.n * p.nparts + (tin * 4 + CLS) * p.ntn + cur.tn;   // PAIR: sample n + 1 is p.nparts further
      if constexpr (STATS) {
        if (tid == 0) {
          const double* q = sred + (k & 1) * 16;
          if constexpr (PAIR) {
            constexpr double cnt = 64.0 * 128.0;
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
              const double S1 = (q[8 * sm] + q[8 * sm + 1]) + (q[8 * sm + 2] + q[8 * sm + 3]);
              const double S2 = (q[8 * sm + 4] + q[8 * sm + 5]) + (q[8 * sm + 6] + q[8 * sm + 7]);
              const double md = S1 / cnt;
              double* o = p.spart + (prec + (long long)sm * p.nparts) * 3;
              o[0] = cnt; o[1] = (double)shift + md; o[2] = S2 - cnt * md * md;
            }
          } else {
            constexpr double cnt = 128.0 * 128.0;
            const double S1 = (q[0] + q[1]) + (q[2] + q[3]), S2 = (q[4] + q[5]) + (q[6] + q[7]);
            const double md = S1 / cnt;
            double* o = p.spart + prec * 3;
            o[0] = cnt; o[1] = (double)shift + md; o[2] = S2 - cnt * md * md;
          }
        }
      }
      __syncthreads();  // C fully read before the buffer is staged again (sred alternates between two sets of slots)
      if constexpr (FUSE) {
        if (tid == 0) {
          const double* q = sred + (k & 1) * 16;
          if constexpr (PAIR) {
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
              double* o = p.nf.part + (prec + (long long)sm * p.nparts) * 2;
              o[0] = q[2 * sm] + q[2 * sm + 1]; o[1] = q[4 + 2 * sm] + q[4 + 2 * sm + 1];
            }
          } else {
            double* o = p.nf.part + prec * 2;
            o[0] = (q[0] + q[1]) + (q[2] + q[3]); o[1] = (q[4] + q[5]) + (q[6] + q[7]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
      if (more && nxt.tn != cur.tn) load_bias(nxt.tn);
      ++k;
  };
  for (int s = 0; s < total; ++s) {
    const bool more = s + 1 < total;
    const bool last_c = c + 1 == nchunk;
    nxt = cur;
    c2 = c + 1;
    if (last_c) { c2 = 0; if (more) nxt = decode(k + 1); }
    if (!more) c2 = c;  // final step: re-request the current slice (valid addresses, results unused) -> no branches below
    wzero = 0;
    asm volatile("" : "+s"(wzero));
    issue(nxt, c2 * KC, hv);
    slice(s);
    if (last_c) epilogue(s, more);
    cur = nxt;
    c = c2;
  }
}

template <bool STATS, bool FUSE, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv_up4_kernel(const U4Params p) {
  static_assert(!(STATS && FUSE), "forward moments and backward sums are never needed together");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;  // (uniform) class of this block: rank 0..3 = classes 3, 1, 2, 0
  // CLASS PAIRS (round 4): with one class per block the four classes' item counts are equal but their lengths are 9 : 6 : 6 : 4 taps,
  // and whole items cannot be dealt 9 : 6 : 6 : 4 over 512 slots — at B = 256 the 8 x 8 level has 256 items per class: the 9-tap
  // blocks take 2 items (18 tap units) or 1, the 6-tap ones 3 or 2, the 4-tap ones 4 or 3, and the launch lasts 18 units against a
  // mean of 12.5 (16 x 16 level: 30 against 25).  A block bound to the class PAIR (3, 0) or (1, 2) runs two persistent loops back to
  // back — still ONE slice body per loop, so no register growth — and a pair is 13 or 12 tap units: 256 + 256 blocks take exactly
  // one item of each of their classes.  Measured: see DESIGN 10.
  if (p.pairmode) {
    // xcdmajor (round 5, LG_U4_XCD): a block's rank inside its type is XCD-major (lg_xcd_remap) — consecutive items, i.e. the column tiles of
    // one source tile and neighbouring tiles, then run on ONE XCD and share its L2 (with the plain rank they are dealt round-robin over
    // the eight L2s).  Valid while the hardware deals blocks to XCDs by blockIdx & 7 and gend[0] is a multiple of 8 (checked on the host).
    const int ga = p.gend[0], gb = p.gend[1] - p.gend[0];
    if (b < ga) {
      const int lb = p.xcdmajor ? lg_xcd_remap(b, ga) : b;
      up4_run<3, STATS, FUSE, PAIR>(p, smem, lb, ga);
      __syncthreads();
      up4_run<0, STATS, FUSE, PAIR>(p, smem, lb, ga);
    } else {
      const int lb = p.xcdmajor ? lg_xcd_remap(b - ga, gb) : b - ga;
      up4_run<1, STATS, FUSE, PAIR>(p, smem, lb, gb);
      __syncthreads();
      up4_run<2, STATS, FUSE, PAIR>(p, smem, lb, gb);
    }
    return;
  }
  if (b < p.gend[0]) up4_run<3, STATS, FUSE, PAIR>(p, smem, b, p.gend[0]);
  else if (b < p.gend[1]) up4_run<1, STATS, FUSE, PAIR>(p, smem, b - p.gend[0], p.gend[1] - p.gend[0]);
  else if (b < p.gend[2]) up4_run<2, STATS, FUSE, PAIR>(p, smem, b - p.gend[1], p.gend[2] - p.gend[1]);
  else up4_run<0, STATS, FUSE, PAIR>(p, smem, b - p.gend[2], p.gend[3] - p.gend[2]);
}

}  // namespace

extern "C" int lg_conv_up4_supported(int B, int Hm, int Wm, int Cs, int N) {
  const bool pair = Hm == 8 && Wm == 8 && B % 2 == 0 && !lg_env_flag("LG_NO_UP4_PAIR");   // 8 x 8 maps: a tile = two samples
  return (!lg_env_flag("LG_NO_UP4") && B > 0 && ((Hm % TH == 0 && Wm % TW == 0) || pair) && Cs % KC == 0 && N % 128 == 0 &&
          (long long)Hm * Wm * Cs * 2 * 2 < (1ll << 31)) ? 1 : 0;   // (two samples below the out-of-range offset of the halo loads)
}

// LG_OK: launched.  LG_ERR_UNSUPPORTED: the caller falls back to conv_halo.hip.  Hm, Wm: the SOURCE (small) map.
// nf (optional; data-gradient use): also the norm-backward sums of the produced gradient ([B][*nparts_out][2] doubles)
extern "C" int lg_conv_up4_nf_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                                  int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                  size_t nf_bytes, void* stream) {
  if (nparts_out) *nparts_out = 0;
  static int off = -1;
  if (off < 0) off = lg_env_flag("LG_NO_UP4") ? 1 : 0;
  if (off || !src16 || !wpack_up || !out16 || !lg_conv_up4_supported(B, Hm, Wm, Cs, N)) return LG_ERR_UNSUPPORTED;
  U4Params p{};
  p.src = (const __bf16*)src16; p.wp = (const char*)wpack_up; p.bias = bias; p.out = (__bf16*)out16;
  p.B = B; p.Hs = Hm; p.Ws = Wm; p.Cs = Cs; p.N = N; p.N32 = N / 32; p.KB = Cs / 16;
  static int nopair = -1;
  if (nopair < 0) nopair = lg_env_flag("LG_NO_UP4_PAIR") ? 1 : 0;
  const bool pair = Hm == 8 && Wm == 8 && B % 2 == 0 && !nopair;
  p.tpi_x = pair ? 1 : Wm / TW; p.tpi = pair ? 1 : p.tpi_x * (Hm / TH); p.ntn = N / 128;
  const long long nper = (long long)(pair ? B / 2 : B) * p.tpi * p.ntn;
  if (nper <= 0 || 4 * nper >= (1ll << 30)) return LG_ERR_UNSUPPORTED;
  p.nper = (int)nper; p.nparts = p.tpi * 4 * p.ntn;
  const bool fuse = nf && nf->z && nf->stats && nf->part && nparts_out && (size_t)B * p.nparts * 2 * sizeof(double) <= nf_bytes;
  const bool stats = !fuse && spart && nparts_out && (size_t)B * p.nparts * 3 * sizeof(double) <= spart_bytes;
  // moments asked for (the caller may then write z as bf16 ONLY) but the workspace cannot hold this tiling's records: decline, so
  // that the dispatch chain tries the next kernel instead of launching without them (lg_conv_fwd_stats_fused's promise)
  if (!nf && spart && nparts_out && !stats) return LG_ERR_UNSUPPORTED;
  p.spart = stats ? (double*)spart : nullptr;
  if (fuse) p.nf = *nf;
  static bool attr_set = false;
  const int nblk = 2 * lg_grid_cus();
  if (!attr_set) {
    attr_set = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_up4_kernel<true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_up4_kernel<false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_up4_kernel<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_up4_kernel<true, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_up4_kernel<false, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_up4_kernel<false, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  // the grid is split between the classes in proportion to their tap counts (9 : 6 : 6 : 4), at most one block per item
  int grid = 0;
  static int classpair = -1;
  if (classpair < 0) classpair = lg_env_flag("LG_U4_NO_CLASSPAIR") ? 0 : 1;
  p.pairmode = classpair;
  if (classpair) {
    // blocks of type A run classes (3, 0): 13 tap units per item pair; type B classes (1, 2): 12.  GA + GB <= nblk, each <= nper:
    // the split with the smallest makespan max(ceil(nper / GA) * 13, ceil(nper / GB) * 12); ties go to the split with fewer idle slots.
    long long best = -1;
    int ga_best = 1, gb_best = 1;
    for (int ga = 1; ga < nblk && ga <= p.nper; ++ga) {
      int gb = nblk - ga;
      if (gb > p.nper) gb = p.nper;
      if (gb < 1) break;
      const long long ca = (long long)((p.nper + ga - 1) / ga) * 13, cb = (long long)((p.nper + gb - 1) / gb) * 12;
      const long long cost = (ca > cb ? ca : cb) * 4096 - (ga + gb);   // makespan first, then as many blocks as fit
      if (best < 0 || cost < best) { best = cost; ga_best = ga; gb_best = gb; }
    }
    p.gend[0] = ga_best; p.gend[1] = ga_best + gb_best; p.gend[2] = p.gend[3] = p.gend[1];
    { static int xm = -1; if (xm < 0) xm = lg_env_flag("LG_U4_XCD") ? 1 : 0; p.xcdmajor = (xm && ga_best % 8 == 0) ? 1 : 0; }
    {
      static int mf = -1;
      if (mf < 0) mf = lg_env_flag("LG_U4_MFAST") ? 1 : 0;
      p.ntm = p.nper / p.ntn;
      p.mfast = (mf && p.ntn > 1 && ga_best % 8 == 0 && gb_best % 8 == 0) ? 1 : 0;
      if (p.mfast) p.xcdmajor = 1;
    }
    grid = ga_best + gb_best;
  } else
  {
    const int taps[4] = {9, 6, 6, 4};
    int left = nblk;
    for (int rk = 0; rk < 4; ++rk) {
      int g = rk == 3 ? left : (nblk * taps[rk] + 12) / 25;
      if (g > left - (3 - rk)) g = left - (3 - rk);
      if (g < 1) g = 1;
      if (g > p.nper) g = p.nper;
      left -= g; grid += g;
      p.gend[rk] = grid;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (pair) {
    if (fuse) hipLaunchKernelGGL((conv_up4_kernel<false, true, true>), dim3(grid), dim3(256), LDS_BYTES, st, p);
    else if (stats) hipLaunchKernelGGL((conv_up4_kernel<true, false, true>), dim3(grid), dim3(256), LDS_BYTES, st, p);
    else hipLaunchKernelGGL((conv_up4_kernel<false, false, true>), dim3(grid), dim3(256), LDS_BYTES, st, p);
  } else if (fuse) hipLaunchKernelGGL((conv_up4_kernel<false, true>), dim3(grid), dim3(256), LDS_BYTES, st, p);
  else if (stats) hipLaunchKernelGGL((conv_up4_kernel<true, false>), dim3(grid), dim3(256), LDS_BYTES, st, p);
  else hipLaunchKernelGGL((conv_up4_kernel<false, false>), dim3(grid), dim3(256), LDS_BYTES, st, p);
  LG_CHECK_LAUNCH("lg_conv_up4");
  lg_note_kernel(pair ? "conv_up4_kernel<PAIR>" : "conv_up4_kernel");
  if (stats || fuse) *nparts_out = p.nparts;
  return LG_OK;
}
extern "C" int lg_conv_up4_try(const void* src16, const void* wpack_up, const float* bias, void* out16, int B, int Hm, int Wm,
                               int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream) {
  return lg_conv_up4_nf_try(src16, wpack_up, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, nullptr, 0, stream);
}
