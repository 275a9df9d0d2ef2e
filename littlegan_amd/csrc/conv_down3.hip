// Software-pipelined stride-2 "down" contraction for gfx950, bf16 operands / fp32 accumulation:
//   out[n][y][x][co] = bias[co] + sum_{ky,kx,ci} src[n][2y+ky-1][2x+kx-1][ci] * W[ky][kx][ci][co]        (TF SAME, k5 s2)
// = Conv2D forward (model.py:15) and the data gradient of Conv2DTranspose (model.py:39-40), for the layers whose
// small map is at least 8 x 16 pixels and whose N is a multiple of 128 (conv2 / conv3 forward, convT3 / convT2 dgrad).
//
// Why a second kernel next to conv_halo.hip: there a block's phases (halo staging, weight-fragment fetch, MFMAs,
// epilogue) run one after the other and only other resident blocks fill the gaps — measured on conv2 forward (B = 256):
// 160 us of which the MFMAs need 43.  Here the phases of ONE block overlap:
//   * persistent blocks (2 per CU) walk a list of (pixel tile, column tile) items;
//   * the source halo of a tile lives in LDS in 16-channel slices, DOUBLE-buffered: while the 25 taps of slice c run out
//     of one buffer, the 16-B pieces of slice c+1 (or of the next item's first slice) are in flight global -> registers
//     and are written to the other buffer behind the taps; one barrier per slice;
//   * the weight fragments (1 KiB per tap, fragment order of pack.hip, straight global -> registers) are fetched one
//     whole slice ahead: the 25 fragments of slice c+1 are requested while slice c computes (25 KiB in flight per wave);
//   * halo rows are 32 B of data in a 48-B pitch and a pixel row is stored de-interleaved (even columns, then odd ones),
//     so every A-fragment address of a tap is the lane's base plus a COMPILE-TIME offset (no address arithmetic in
//     the tap loop) and ds_read_b128 is conflict-free (16 consecutive rows x 48 B hit 16 distinct 16-B slots);
//   * the product is formed transposed (channels on the accumulator rows, pixels on the lanes): a lane then holds 4
//     consecutive channels of its pixel per register group, the tile leaves the registers through the idle halo buffer
//     with 8-byte LDS writes and reaches HBM as whole 256-B pixel rows, 16 B per lane;
//   * the InstanceNormalization moments of the tile ({count, mean, M2}, instance.py:114-115) come out of the same
//     registers, so no pass re-reads the output.
#include <stdlib.h>
#include <type_traits>
#include "lg_common.h"

namespace {

constexpr int TH = 8, TW = 16;                 // output tile (pixels)
constexpr int HH = 2 * TH + 3;                 // halo rows  (19)
constexpr int ROWB = 48;                       // LDS pitch of a halo pixel (32 B of channels + 16 B)
constexpr int KC = 16;                         // channels per slice = one MFMA k-step
constexpr int NTAP = 25;
// Halo image of a tile.  A pixel row is stored de-interleaved: HWH even-column slots, then HWH odd-column slots.
//   default: one sample, 8 x 16 output pixels -> 35 halo columns, 18 + 18 slots.
//   PAIR (8 x 8 maps): the tile is TWO samples side by side (columns 0..7 = sample n, 8..15 = sample n + 1), each with its
//   own 19-column halo.  A ds_read_b128 is served in four 16-lane groups and pix32 makes such a group one tile row = 8
//   pixels of each sample, 48 B apart: conflict-free only if sample 1's image starts 128 B (mod 256) behind sample 0's.
//   Slots of a halo row (48 B each): sample 0 even 0..9, odd 10..18, (19..23 unused), sample 1 even 24..33, odd 34..42:
//   24 slots = 1152 B = 128 (mod 256), the odd arrays sit HWH = 10 slots behind the even ones in both samples, 43 per row.
//   (Measured with the two images 10 slots apart: SQ_LDS_BANK_CONFLICT = 49 % of the LDS cycles, LDS-active time doubled.)
template <bool PAIR>
struct D3L {
  static constexpr int HWH = PAIR ? 10 : 18, HWP = PAIR ? 43 : 2 * HWH;
  static constexpr int NROWS = HH * HWP;               // halo pixels per tile (684 | 836)
  static constexpr int HB = NROWS * ROWB;              // bytes per buffer (32832 | 40128)
  static constexpr int NPIECE = NROWS * 2;             // 16-B pieces per slice
  static constexpr int PPT = (NPIECE + 255) / 256;     // pieces per thread (6 | 7)
  static constexpr int SRED_OFF = 2 * HB;              // 32 doubles of reduction scratch behind the two buffers
  static constexpr int SBIAS_OFF = 2 * HB + 256;       // 128 floats: bias of the block's current column tile
  static constexpr int LDS_BYTES = 2 * HB + 256 + 512;
};
#ifndef LG_D3_DBG
#define LG_D3_DBG 0   // compile-time ablation bits (timing only, results wrong): 1 no MFMA, 2 no fragment loads, 4 no halo staging, 8 no epilogue
#endif
constexpr int DBG = LG_D3_DBG;
#ifndef LG_D3_RING
#define LG_D3_RING 10   // weight-fragment look-ahead in taps (5: one instantiation of the slice body; 10: two, ring offsets 0 / 5)
#endif
#ifndef LG_D3_SCHED
#define LG_D3_SCHED 1   // 0: tap body in three pinned groups (A reads | MFMAs | ring refill); 1: interleaved by sched_group_barrier
#endif
#ifndef LG_D3_HALO_AUX
#define LG_D3_HALO_AUX 0   // cache policy bits of the halo loads (2 = nt: streamed-once activations; A/B builds)
#endif
#ifndef LG_D3_ADEPTH
#define LG_D3_ADEPTH 2  // A-fragment buffers: 2 = requested one tap ahead, 3 = two taps ahead
#endif

struct D3Params {
  const __bf16* src;   // [B][Hs][Ws][Cs] bf16
  const char* wp;      // fragment-ordered pack [25][N32][KB][64][16 B]
  const float* bias;   // [N] or null
  __bf16* out;         // [B][Hm][Wm][N] bf16
  double* spart;       // [B][nparts][3] or null
  int B, Hs, Ws, Cs, Hm, Wm, N, N32, KB;
  int tpi_x, tpi, ntn, nitems, nparts;
  LgNormFuse nf;       // FUSE instantiation: norm-backward sums of the produced gradient (lg_common.h)
  const float* nstats; // NORM = 1: src is the RAW output z of the layer below, these are its statistics records [B][8]
  float nalpha;        //   ... and the operand is bf16(leaky(InstanceNorm(z))), formed while the halo is staged
  const __bf16* gsrc;  // NORM = 2 (BWDNORM): src is the raw output z of THIS level, gsrc the gradient g w.r.t. its normalised + activated
  const float* bcoef;  //   map, bcoef the per-sample records of lg_instnorm_bwd_coef: the operand dz is formed while the halo is staged
  unsigned long long* clk;  // clock census (runtime.hip: lg_set_clock_census) or null
  int lists;           // item lists (LG_D3_LISTS): 0 rounds with a balanced partial round (default), 1 the round-2 lists, 2 one contiguous range per XCD
  int slice_major;     // experiment (round 5, LG_D3_SLICE_MAJOR): the source is [B][Cs/16][Hs][Ws][16] (channel-slice-major) instead of NHWC
  int lds_order;       // 1: halo pieces dealt to the threads in LDS order (the round-2 map; LG_D3_LDS_ORDER, A/B), 0: in memory order
  int stagger;         // start delay of the odd-slot block in ~1024-cycle units
  unsigned long long* stamps;  // diagnostic build only (LG_D3_STAMPS): [block][64] s_memtime stamps of wave 0
  int stamp_lite;              // only the block's first / last stamp (the per-phase stamps cost ~11 % and change the clock)
};

__device__ __forceinline__ int pix32(int r) {  // MFMA row -> tile pixel inside its 32-pixel group (see conv_halo.hip)
  const int q = r >> 2, lo = r & 3;
  const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;
  const int rank = odd ? ((q == 1) ? 0 : (q == 2) ? 1 : (q == 4) ? 2 : 3) : ((q == 0) ? 0 : (q == 3) ? 1 : (q == 5) ? 2 : 3);
  return odd * 16 + rank * 4 + lo;
}

template <bool PAIR>
constexpr int toff_bytes(int t) {  // tap t = ky*5+kx: LDS byte offset of its source pixel relative to the lane's base
  return ((t / 5) * D3L<PAIR>::HWP + ((t % 5) >> 1) + ((t % 5) & 1) * D3L<PAIR>::HWH) * ROWB;
}

// NW = output channels per tile: 128 (each of the 4 waves owns 32 channels x all 128 pixels) or 64 (2 x 2 waves: 32 channels
// x 64 pixels each — half the MFMAs per weight fragment, for the layers whose N is only a multiple of 64)
// NORM: the source is the raw bf16 conv output z of the layer below; InstanceNormalization + LeakyReLU (instance.py:105-128,
// model.py:22-24) are applied to each halo piece between its buffer load and its LDS store — the same arithmetic and rounding as
// apply16_kernel (norm.hip), so the operand image is bit-identical to the one a stand-alone apply pass would have written, and
// that pass with its tensor disappears.  Pieces outside the image stay zero (TF SAME pads the NORMALISED map).  Forward passes
// nothing differentiates the weights of (the discriminator run on the Adjuster's output, eager_trainer.py:158-160): there the
// normalised maps have no other reader.  Measured at 2B = 512 (scripts/probe/zn_layers.py), apply pass + conv -> normalising conv:
// conv2 (64 -> 128 channels, 64 x 64 map) 101 + 266 -> 272 us; conv3 (128 -> 256, 32 x 32) 46 + 197 -> 227 us.  The sample-PAIR
// tiling of the 8 x 8 level was built too and LOSES (conv4: 27 + 189 -> 228 us: 16 slices per item, each re-normalising a halo the
// three column tiles share) — it has no normalising form.
// NORM = 2 (BWDNORM, round 4): the data gradient of a transposed conv whose incoming gradient has not been taken through the
// InstanceNorm + LeakyReLU backward yet.  The source is the pair (z, g) of the level; dz = a (g' - m1 - c m2') — bwd_apply16_kernel's
// arithmetic and rounding, lg_bwdnorm8 — is formed per 16-B halo piece between the two buffer loads and the LDS store, so the
// norm-backward apply pass and its dz tensor disappear (6 B per element of HBM traffic less; the piece loads double).  For tapes that
// ask the level for no weight gradient: the Adjuster's decoder chain at 2B (eager_trainer.py:158-163), partition steps of the
// Generator.  Bit-identical to bwd_apply16 + conv (tests/test_launch_shapes_gpu.py).
// Measured in the C3 step (round 4, same box, A/B by LG_NO_BWDNORM): the N = 64 level (convT4's data gradient at 2B = 512: apply pass
// 303 us + conv 321 us -> 552 us) gains 72 us per launch; the N = 128 level (convT3: 150 + 222 -> 395 us) LOSES 23 — 17 VALU
// operations per element, 1.3 x the elements (halo), land on a kernel that was matrix-bound, not HBM-bound — and is not instantiated.
// Two facts from the way: (1) the staging arithmetic must be pinned behind the tap loop (see commit below); (2) the per-item
// coefficient record is read through the constant address space (lg_as_const, lg_common.h: as a plain load hipcc made it a VECTOR
// load inside the item loop, `s_waitcnt vmcnt(0)` at every item boundary) AND every field is made wave-uniform explicitly
// (lg_uniform), so that the staging arithmetic takes scalar operands.  The vector-load build (-DLG_D3_COEF_PLAIN at commit 43a08f2) was
// NOT deterministic with two blocks per CU: hipcc kept the record as per-lane copies paired in 64-bit registers and formed g' - m1 as
// a packed fp32 subtraction selecting the pair's HIGH register; read back through one-hot weights (tests/diagnostics/bwdnorm_probe.py)
// every wrong operand element — 82 of 82 — was the LOW result of that instruction, bit-equal to the value with m1 not subtracted.
// No wait, barrier or scheduling hole is involved (full waits in front of every instruction leave it in place); round-5 analysis in
// DESIGN 11a.  tests/test_launch_shapes_gpu.py::test_persistent_kernels_are_deterministic launches this form at B = 512.
template <bool STATS, bool FUSE = false, bool PAIR = false, int NW = 128, int NORM = 0>
__global__ __launch_bounds__(256, 2) void conv_down3_kernel(const D3Params p) {
  static_assert(!(STATS && FUSE), "forward moments and backward sums are never needed together");
  static_assert(NORM != 1 || (STATS && NW == 128 && !PAIR), "the normalising form exists for the forward passes with fused moments, 8 x 16 tiles");
  static_assert(NORM != 2 || (!STATS && !PAIR && NW == 64), "the backward-normalising form exists for the 64-column data gradients on 8 x 16 tiles");
  static_assert(NW == 128 || (NW == 64 && !PAIR), "tile widths");
  constexpr int NWV = NW / 32, NI = NW / 32;          // waves along the channels; 32-pixel groups per wave (4 | 2)
  constexpr int PPR = NW / 8, NQ = 128 * PPR / 256;   // 16-B pieces per output pixel row; pieces per thread in the row sweep
  using L = D3L<PAIR>;
  constexpr int HWH = L::HWH, HWP = L::HWP, HB = L::HB, NPIECE = L::NPIECE, PPT = L::PPT, SRED_OFF = L::SRED_OFF, SBIAS_OFF = L::SBIAS_OFF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sred = reinterpret_cast<double*>(smem + SRED_OFF);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: the weight addresses stay in SGPRs
  const int wn = wid % NWV, wm = wid / NWV;                  // this wave: channels 32 wn .., pixel groups wm * NI ..
  const int G = gridDim.x;
  const int nchunk = p.Cs / KC;
  // (grid <= nitems).  Dealing the items out dynamically (a counter per XCD, first item static) was built
  // and measured in round 3: it closes the 16 us gap between the two blocks of a CU (the older wave wins the matrix-pipe
  // arbitration) to 5 us, but with 4 items of ~23 us per block the last block still ends one item late: kernel time unchanged.
  // Item lists (round 5): the XCD x (the hardware deals workgroups to the 8 XCDs round-robin: blockIdx & 7) owns a CONTIGUOUS range of
  // nitems / 8 items — its L2 then holds a contiguous run of tiles / samples — and its blocks walk that range with the XCD's block
  // count as the stride.  The round-2 lists (items lb, lb + G, ... with lb the XCD-major block rank) gave the remainder of
  // nitems / G to the LOWEST ranks = the first XCDs: at 768 items on 512 blocks (the 8 x 8 level at 2B) every block of XCDs 0-3 ran
  // two items and every block of XCDs 4-7 one — half the chip idle for the second half of the launch.
  // Two balanced forms (p.lists, LG_D3_LISTS): 2 = the contiguous range per XCD just described; 0 (default) = ROUNDS — the whole chip sweeps the map
  // front to back in rounds of G items as in round 2 (the producer of the map has just written it and its front is what the caches still hold), and only
  // the LAST, partial round is dealt out as one contiguous sub-range per XCD.  1 = the round-2 lists (A/B).
  const int nx = G < 8 ? G : 8, xcd = (int)blockIdx.x % nx, xidx = (int)blockIdx.x / nx;
  const bool ranges = p.lists == 2;
  const int nbx = ranges ? (G - xcd + nx - 1) / nx : G;                      // stride of a block's list
  const int icnt = (p.nitems - xcd + nx - 1) / nx;                           // ranges: items of this XCD (>= nbx: grid <= nitems)
  const int istart = xcd * (p.nitems / nx) + (xcd < p.nitems % nx ? xcd : p.nitems % nx);
  const int lb = ranges ? istart + xidx : lg_xcd_remap(blockIdx.x, G);       // first item; then lb + nbx, lb + 2 nbx, ...
  const int nfull = p.nitems / G, nrem = p.nitems - nfull * G;               // rounds: full rounds, items of the partial one
  const int rcnt = (nrem - xcd + nx - 1) / nx;                               // ... of which this XCD takes rcnt, from rstart on
  const int rstart = nfull * G + xcd * (nrem / nx) + (xcd < nrem % nx ? xcd : nrem % nx);
  const int nmine = ranges ? (icnt - xidx + nbx - 1) / nbx : p.lists == 1 ? (p.nitems - lb + G - 1) / G : nfull + (xidx < rcnt ? 1 : 0);
  const int total = nmine * nchunk;
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();   // clock census (see the end)
  // ---- the (up to) 6 halo pieces this thread stages in every slice: LDS offset and position inside the halo ----------
  int pl[PPT], pyx[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int q = tid + u * 256;
    pl[u] = -1; pyx[u] = 0;
    if (q < NPIECE) {
      const int row = q >> 1, hy = row / HWP, hxp = row - hy * HWP;
      pl[u] = row * ROWB + (q & 1) * 16;
      if constexpr (PAIR) {  // slots 0..19: sample 0 (even 0..9, odd 10..19), 20..23 unused, 24..42: sample 1 (even, odd)
        const int sm = hxp >= 24, j2 = hxp - 24 * sm, odd = j2 >= HWH, hx = 2 * (j2 - odd * HWH) + odd;
        pyx[u] = (hxp < 20 || hxp >= 24) && hx < 2 * 8 + 3 ? (hy << 8) | (sm << 7) | hx : (0x7fff << 8);
      } else {
        const int hx = hxp < HWH ? 2 * hxp : 2 * (hxp - HWH) + 1;
        pyx[u] = hx < 2 * TW + 3 ? (hy << 8) | hx : (0x7fff << 8);  // the 36th slot of a row is padding: never valid
      }
    }
    if constexpr (!PAIR) {
      // GLOBAL-ORDER pieces (round 4; p.lds_order = 1 restores the LDS-order map above for A/B): consecutive lanes take consecutive
      // 16-B pieces of the source in MEMORY order — (pixel x, half 0), (x, half 1), (x + 1, half 0), ... along a halo row — and write
      // them to their de-interleaved LDS slots.  In LDS order a wave's 64 pieces are 32 pixels two columns apart: for the 32-channel
      // source of the N = 64 level (64 B per pixel) every load instruction touched 32 cache lines and used a quarter of each; in
      // memory order it touches 16 and uses half (the other half is the next slice).  The padding slot of a row is never read
      // (largest slot a tap reaches: 34) and no longer written.
      if (!p.lds_order) {
        constexpr int PPROW = 2 * (2 * TW + 3);   // 70 real pieces per halo row
        pl[u] = -1; pyx[u] = (0x7fff << 8);
        if (q < HH * PPROW) {
          const int hy = q / PPROW, rem = q - hy * PPROW, hx = rem >> 1;
          const int hxp = (hx & 1) ? HWH + (hx >> 1) : (hx >> 1);
          pl[u] = (hy * HWP + hxp) * ROWB + (rem & 1) * 16;
          pyx[u] = (hy << 8) | hx;
        }
      }
    }
  }
  const int half8 = (tid & 1) * 8;  // channel offset of this thread's pieces inside the slice (256 is even: same for all)

  // ---- per-lane A bases: pixel m = i*32 + pix32(r) of the 8 x 16 tile -> halo pixel (2 ly, lx) of the even columns ----
  int abase[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = (wm * NI + i) * 32 + pix32(r);
    const int col = m & 15;
    abase[i] = (2 * (m >> 4) * HWP + (PAIR ? col + 16 * (col >> 3) : col)) * ROWB + h * 16;  // PAIR: sample 1 starts at slot 24
  }

  struct Item { int n, y0, x0, tn; };
  auto decode = [&](int k) {
    const int item = (p.lists == 0 && k == nfull) ? rstart + xidx : lb + k * nbx;
    Item it;
    it.tn = item % p.ntn;
    const int tm = item / p.ntn;
    if constexpr (PAIR) { it.n = 2 * tm; it.y0 = 0; it.x0 = 0; return it; }
    it.n = tm / p.tpi;
    const int tt = tm - it.n * p.tpi;
    it.y0 = (tt / p.tpi_x) * TH; it.x0 = (tt % p.tpi_x) * TW;
    return it;
  };
  // The halo pieces are fetched with RAW BUFFER loads from a descriptor over the item's sample(s): a piece outside the image
  // (TF SAME padding), in a padding slot or beyond the thread's share gets an out-of-range offset and comes back as zeros.
  // No branch surrounds a load.  (With `if (inside) v = load` hipcc branches around every load, can no longer count the loads
  // in flight at the join, and falls back to `s_waitcnt vmcnt(4)` in front of the first taps of every other slice: the wave then
  // sat out the HBM latency of the halo it had just requested — 1600 of a slice's 5100 cycles for a lone wave, r3 census.)
  constexpr unsigned OOB = 0x80000000u;   // >= num_records for every supported shape (checked on the host)
  const int sample_elems = p.Hs * p.Ws * p.Cs;
  // The piece offsets of an item are formed ONCE (set_item: ~90 VALU operations per thread) and every slice adds its channel offset:
  // the slice boundary — where this wave issues no MFMA — is what a lone wave per SIMD pays in full (r3 census: 60 % of the pipe).
  unsigned hoff[PPT];
  float nmu0 = 0.f, nml0 = 0.f, nna0 = 0.f, nnb0 = 0.f;   // NORM: the item's sample
  LgBwdCoef bco{};                                         // BWDNORM: the item's sample
  auto set_item = [&](const Item& it) __attribute__((always_inline)) {
    if constexpr (NORM == 1) {
      const lg_const_f32p sp = lg_as_const(p.nstats + (long long)it.n * 8);   // uniform + constant space: scalar loads
      nmu0 = lg_uniform(sp[0]); nna0 = lg_uniform(sp[2]); nnb0 = lg_uniform(sp[3]); nml0 = lg_uniform(sp[4]);   // SGPRs whatever the load (lg_common.h)
    }
    if constexpr (NORM == 2) {
#ifdef LG_D3_COEF_PLAIN   // diagnostic build: the round-4 form whose BWDNORM instantiation was not deterministic
      const float* sp = p.bcoef + (long long)it.n * 8;
#else
      const lg_const_f32p sp = lg_as_const(p.bcoef + (long long)it.n * 8);    // uniform + constant space: one 32-byte scalar load
#endif
#ifdef LG_D3_COEF_PLAIN
      bco.mu = sp[0]; bco.mul = sp[1]; bco.a = sp[2]; bco.b = sp[3]; bco.m1 = sp[4]; bco.m2 = sp[5]; bco.m1l = sp[6]; bco.m2l = sp[7];
#else   // explicitly wave-uniform: the staging arithmetic takes SCALAR operands whatever load the compiler emits (lg_common.h)
      bco.mu = lg_uniform(sp[0]); bco.mul = lg_uniform(sp[1]); bco.a = lg_uniform(sp[2]); bco.b = lg_uniform(sp[3]);
      bco.m1 = lg_uniform(sp[4]); bco.m2 = lg_uniform(sp[5]); bco.m1l = lg_uniform(sp[6]); bco.m2l = lg_uniform(sp[7]);
#endif
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int sy = 2 * it.y0 - 1 + (pyx[u] >> 8), sx = 2 * it.x0 - 1 + (pyx[u] & (PAIR ? 127 : 255));
      const int sl = PAIR ? (pyx[u] >> 7) & 1 : 0;   // PAIR: second sample of the tile
      const bool ok = pl[u] >= 0 && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      // slice-major source: a pixel of a slice is 32 contiguous bytes, a slice a plane of Hs x Ws x 32 B (the channel offset c0 of issue() is
      // rescaled there); NHWC: the pixel's Cs x 2 bytes, 32 of them used per slice
      hoff[u] = ok ? (p.slice_major ? (unsigned)(sl * p.Hs * p.Ws * p.Cs * 2 + ((sy * p.Ws + sx) * KC + half8) * 2)
                                    : (unsigned)((((sl * p.Hs + sy) * p.Ws + sx) * p.Cs + half8) * 2)) : OOB;   // (+ 2 c0 < 2^31 keeps OOB out of range)
    }
  };
  const unsigned cmul = p.slice_major ? (unsigned)(p.Hs * p.Ws * 2) : 2u;   // bytes per channel step of the slice offset: c0 * cmul
  auto issue = [&](const Item& it, int c0, u32x4 (&v)[PPT]) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(p.src + (long long)it.n * sample_elems), 0, (PAIR ? 2 : 1) * sample_elems * 2, 0x00027000);
#pragma unroll
    for (int u = 0; u < PPT; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hoff[u] + (unsigned)c0 * cmul, 0, LG_D3_HALO_AUX));
  };
  auto issue_g = [&](const Item& it, int c0, u32x4 (&v)[NORM == 2 ? PPT : 1]) __attribute__((always_inline)) {   // BWDNORM: the same pieces of g
    if constexpr (NORM == 2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<__bf16*>(p.gsrc + (long long)it.n * sample_elems), 0, sample_elems * 2, 0x00027000);
#pragma unroll
      for (int u = 0; u < PPT; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hoff[u] + (unsigned)c0 * cmul, 0, LG_D3_HALO_AUX));
    }
  };
  u32x4 gv[NORM == 2 ? PPT : 1];
  auto commit = [&](char* buf, const u32x4 (&v)[PPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PPT; ++u)
      if (pl[u] >= 0) {
        u32x4 w = v[u];
        if constexpr (NORM == 2) {   // out-of-range pieces: z = g = 0 came back, but dz(0, 0) != 0 — the padding of dz is zero
          const u32x4 dn = lg_bwdnorm8(w, gv[u], bco, p.nalpha);
          const bool inside = hoff[u] != OOB;
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = inside ? dn[k] : 0u;
        }
        if constexpr (NORM == 1) {   // (hoff is the offset this very piece was requested with: commit follows the issue of the same slice)
          const u32x4 hn = lg_norm8(w, nmu0, nml0, nna0, nnb0, p.nalpha);
          const bool inside = hoff[u] != OOB;
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = inside ? hn[k] : 0u;
        }
        if constexpr ((DBG & 16) != 0) {  // probe: the VALU cost of InstanceNorm + LeakyReLU applied while staging (norm.hip apply16 arithmetic)
          const float mu = p.nf.alpha, mul = 0.001f, na = 1.01f, nb = 0.02f, al = 0.3f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t0 = __builtin_bit_cast(float, w[k] << 16), t1 = __builtin_bit_cast(float, w[k] & 0xffff0000u);
            t0 = na * ((t0 - mu) - mul) + nb; t1 = na * ((t1 - mu) - mul) + nb;
            t0 = fmaxf(t0, al * t0); t1 = fmaxf(t1, al * t1);
            const __bf16 h0 = (__bf16)t0, h1 = (__bf16)t1;
            const unsigned o = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            w[k] = pyx[u] != (0x7fff << 8) ? o : 0u;
          }
        }
        *reinterpret_cast<u32x4*>(buf + pl[u]) = w;
      }
  };
  // weight fragment of (tap t, column tile tn, k-step kb) for this wave's 32 columns
  const unsigned lane16 = lane * 16;
  const unsigned wstride = (unsigned)p.N32 * (unsigned)p.KB * 1024u;  // bytes from one tap's fragments to the next tap's
  auto wbase = [&](int tn, int kb) {  // uniform 64-bit base (SGPRs) of (column tile, k-step); taps are wstride apart
    return p.wp + ((long long)(tn * NWV + wn) * p.KB + kb) * 1024;
  };
  auto wfrag = [&](const char* base, int t) {  // SGPR base + 32-bit lane offset: no 64-bit arithmetic in the tap loop
    return *reinterpret_cast<const u32x4*>((base + (unsigned long long)t * wstride) + lane16);  // scalar add, lane offset
  };

  f32x16 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // Weight-fragment ring: RING fragments (= taps) ahead of the MFMAs.  25 taps per slice and RING = 10 -> the ring
  // position of tap 0 alternates between 0 and 5 from one slice to the next: the slice body exists in two copies
  // (OFF = 0 / 5) so that every ring access is a compile-time register.
  constexpr int RING = LG_D3_RING;
  constexpr int AD = LG_D3_ADEPTH;
  u32x4 bf[RING];
  u32x4 hv[PPT];

  // ---- prologue: slice 0 of the first item ---------------------------------------------------------------------------
  Item cur = decode(0);
  set_item(cur);
  issue(cur, 0, hv);
  issue_g(cur, 0, gv);
#pragma unroll
  for (int t = 0; t < RING; ++t) bf[t] = wfrag(wbase(cur.tn, 0), t);
  commit(smem, hv);
  __syncthreads();

#ifdef LG_D3_STAMPS
  int nst = 0;
#define D3_STAMP() do { if (p.stamps && !p.stamp_lite && wid == 0 && nst < 60) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) p.stamps[(long long)blockIdx.x * 64 + nst] = t_; ++nst; } } while (0)
  // residency census (diagnostic build only): where and when this block started — slot 61 = HW_ID | XCC_ID << 32, 62 / 63 = the
  // chip-wide 100 MHz clock (s_memrealtime) at the block's start / end, comparable ACROSS blocks (s_memtime is per XCD)
  if (p.stamps && wid == 0 && lane == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    p.stamps[(long long)blockIdx.x * 64 + 61] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    p.stamps[(long long)blockIdx.x * 64 + 62] = __builtin_amdgcn_s_memrealtime();
    p.stamps[(long long)blockIdx.x * 64 + 59] = __builtin_amdgcn_s_memtime();
  }
#else
#define D3_STAMP() do {} while (0)
#endif
  // bias of this wave's 32 channels in accumulator order (reloaded only when the column tile changes) and the shift of
  // the one-pass moments
  float* sbias = reinterpret_cast<float*>(smem + SBIAS_OFF);
  auto load_bias = [&](int tn) {  // (a barrier lies between this and the next epilogue that reads it)
    if (tid < NW) sbias[tid] = p.bias ? p.bias[tn * NW + tid] : 0.f;
  };
  load_bias(cur.tn);
  {  // Two blocks share a CU (one wave of each per SIMD) and run the same periodic program: started together they stay
     // in lockstep and meet at the matrix pipe, then idle it together in their epilogues.  The block whose waves sit in
     // the odd wave slot starts about one epilogue late, so that one block's epilogue / barrier falls into the other's
     // tap phase (MI355X_MICROARCH, two waves per SIMD, item 9).  Speed only; bounded.
    // (the dispatcher deals one block to every CU before any CU gets its second: the late half = the upper half of the grid)
    if (p.stagger && (int)blockIdx.x >= (G + 1) / 2)
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);  // ~1024 cycles each
  }
  int k = 0, c = 0;  // item index in this block's list, slice index
  D3_STAMP();
  // (Round 3, measured and removed: the two blocks of a CU finish 16 us apart — at equal priority the older wave wins the matrix-pipe
  //  arbitration.  Dealing the items out dynamically, alternating s_setprio per item, and a per-CU progress board that gives the
  //  block that is behind priority 1 all CLOSE the gap (to 5 / 8 / 3 us) and none shortens the kernel: with forced turns both blocks
  //  end late, 103 us instead of 83 / 99 — a prioritised partner costs the other wave more than it gains, MI355X guide "two waves per
  //  SIMD" item 2.  Uneven static lists — 9/16 or 10/16 of the items to the grid's lower half — change nothing either: +-2 % per layer.)
  auto slice = [&](auto off_c, int s) {
    constexpr int OFF = decltype(off_c)::value;
    const bool more = s + 1 < total;
    const bool last_c = c + 1 == nchunk;
    Item nxt = cur;
    int c2 = c + 1;
    if (last_c) { c2 = 0; if (more) { nxt = decode(k + 1); set_item(nxt); } }
    if (!more) c2 = c;  // final step: re-request the current slice (valid addresses, results unused) -> no branches below
    if constexpr (!(DBG & 4)) { issue(nxt, c2 * KC, hv); issue_g(nxt, c2 * KC, gv); }
    const char* hbuf = smem + (s & 1) * HB;
    const char* wcur = wbase(cur.tn, c);
    const char* wnxt = wbase(nxt.tn, c2);
    // The order below is PINNED (sched_barrier): left alone, hipcc sinks every fragment load down to its use and
    // serialises each MFMA behind its own LDS read (measured in the .s: "ds_read; s_waitcnt lgkmcnt(0); v_mfma" per
    // MFMA, "global_load; s_waitcnt vmcnt(0)" per tap).  Per tap: the A fragments of tap t+1 are requested, then the 4
    // MFMAs of tap t issue, then the ring slot they freed is refilled RING taps ahead.
    bf16x8 a[AD][NI];
#pragma unroll
    for (int d = 0; d + 1 < AD; ++d)
#pragma unroll
      for (int i = 0; i < NI; ++i) a[d][i] = *reinterpret_cast<const bf16x8*>(hbuf + abase[i] + toff_bytes<PAIR>(d));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (t + AD - 1 < NTAP) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
          a[(t + AD - 1) % AD][i] = *reinterpret_cast<const bf16x8*>(hbuf + abase[i] + toff_bytes<PAIR>(t + AD - 1 < NTAP ? t + AD - 1 : 0));
      }
      if constexpr (LG_D3_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      const int slot = (t + OFF) % RING;
#pragma unroll
      for (int i = 0; i < NI; ++i)  // transposed product: rows = this wave's 32 output channels, columns = 32 pixels
        if constexpr (!(DBG & 1)) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[slot]), a[t % AD][i], acc[i], 0, 0, 0);
      if constexpr (LG_D3_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
      // the slot is free: request the fragment RING taps ahead (this slice, or the next one)
      if constexpr (!(DBG & 2)) {
        if (t + RING < NTAP) bf[slot] = wfrag(wcur, t + RING);
        else bf[slot] = wfrag(wnxt, t + RING - NTAP);
      }
      if constexpr (LG_D3_SCHED == 0) {
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // one MFMA, then one LDS read in its shadow (an MFMA holds the vector issue port for 8 of its 32 cycles), ..., the
        // ring refill behind the last MFMA: the wave never leaves the MFMA pipe idle to issue its loads
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
    }
    D3_STAMP();   // taps done
    // NORM forms: the staging arithmetic stays BEHIND the tap loop.  Left free, hipcc hoists its first unpacking shifts into the first
    // taps of the slice, and with them an `s_waitcnt vmcnt(2)` for the halo pieces requested a few instructions earlier: every slice
    // then opens by sitting out the HBM latency of its own prefetch (seen in the .s of the BWDNORM form, round 4).
    if constexpr (NORM != 0) __builtin_amdgcn_sched_barrier(0);
    commit(smem + ((s + 1) & 1) * HB, hv);
    __syncthreads();  // slice s consumed by every wave, slice s+1 complete
    D3_STAMP();   // barrier passed

    if (last_c && !(DBG & 8)) {
      // ---- epilogue of item `cur`: acc[i][e] = channel (e&3) + 8*(e>>2) + 4*h of pixel i*32 + pix32(r) ---------------
      char* C = smem + (s & 1) * HB;  // the buffer slice s ran out of is idle until the commit of step s+1
      // moments in ONE pass: sums of d = v - shift and d^2 (shift = the block's first bias value, near the tile mean for
      // zero-mean kernels), fp32 per lane over its 64 values, fp64 from there on:  M2 = S2 - S1^2 / count
      const float shift = STATS ? sbias[0] : 0.f;
      const f32x2 shift2 = {shift, shift};
      f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
      f32x4 bq[4];  // bias of channels 8g + 4h + {0..3} of this wave's 32
#pragma unroll
      for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(sbias + wn * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = (wm * NI + i) * 32 + pix32(r);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {  // two channels at a time: v_pk_add_f32 / v_pk_fma_f32 halve the VALU count
            const f32x2 v = f32x2{acc[i][4 * g + 2 * jj], acc[i][4 * g + 2 * jj + 1]} + f32x2{bq[g][2 * jj], bq[g][2 * jj + 1]};
            if constexpr (STATS) {
              const f32x2 d = v - shift2;
              s1v += d;
              s2v = __builtin_elementwise_fma(d, d, s2v);
            }
            w[2 * jj] = (__bf16)v[0]; w[2 * jj + 1] = (__bf16)v[1];
          }
          // 16-B piece (wid*4 + g) of the 256-B row, XOR-swizzled by the row so that both the 8-B writes of a lane group
          // (16 rows, same piece) and the 16-B row reads (same row, 16 pieces) are conflict-free
          *reinterpret_cast<bf16x4*>(C + row * (NW * 2) + (((wn * 4 + g) ^ (row & (PPR - 1))) << 4) + 8 * h) = w;
        }
      }
      if constexpr (STATS) {
        const double l1 = (double)s1v[0] + (double)s1v[1], l2 = (double)s2v[0] + (double)s2v[1];
        if constexpr (PAIR) {  // a lane's 128 values all belong to ONE sample (its tile column never changes): masked sums
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
      // element offset of tile pixel `row` (= 16 y + column) in the output / in z.  PAIR: columns 8..15 are sample n + 1
      const long long obase = ((long long)(cur.n * p.Hm + cur.y0) * p.Wm + cur.x0) * p.N + cur.tn * NW;
      auto pix_off = [&](int row) -> long long {
        if constexpr (PAIR) return obase + (long long)(((row >> 3) & 1) * 64 + (row >> 4) * 8 + (row & 7)) * p.N;
        else return obase + ((long long)(row >> 4) * p.Wm + (row & 15)) * p.N;
      };
      // FUSE: the z pieces of the row sweep are requested now — the accumulators are dead (staged), the loads land behind
      // the barrier instead of in front of every use
      u32x4 zq[FUSE ? NQ : 1];
      (void)zq;
      if constexpr (FUSE) {
#pragma unroll
        for (int q8 = 0; q8 < NQ; ++q8) {
          const int piece = tid + q8 * 256, row = piece / PPR, j = piece % PPR;
          zq[q8] = *reinterpret_cast<const u32x4*>(p.nf.z + pix_off(row) + j * 8);
        }
      }
      __syncthreads();  // tile complete in LDS (and the wave sums)
      float nf1 = 0.f, nf2 = 0.f;
      (void)nf1; (void)nf2;
#pragma unroll
      for (int q8 = 0; q8 < NQ; ++q8) {
        const int piece = tid + q8 * 256, row = piece / PPR, j = piece % PPR;
        const u32x4 v = *reinterpret_cast<const u32x4*>(C + row * (NW * 2) + ((j ^ (row & (PPR - 1))) << 4));
        const long long goff = pix_off(row) + j * 8;
        *reinterpret_cast<u32x4*>(p.out + goff) = v;
        if constexpr (FUSE) {
          // PAIR: the tile column of a thread's pieces is fixed ((tid >> 4) & 15): waves 0, 1 sweep sample n, waves 2, 3 n + 1
          const lg_const_f32p sp = lg_as_const(p.nf.stats + (long long)(cur.n + (PAIR ? wid >> 1 : 0)) * 8);   // scalar loads (lg_common.h)
          lg_nf_accum(v, zq[q8], lg_uniform(sp[0]), lg_uniform(sp[4]), lg_uniform(sp[2]), lg_uniform(sp[3]), p.nf.alpha, nf1, nf2);
        }
      }
      if constexpr (FUSE) {
        const double w1 = lg_wave_sum_d((double)nf1), w2 = lg_wave_sum_d((double)nf2);
        if (lane == 0) { sred[(k & 1) * 16 + wid] = w1; sred[(k & 1) * 16 + 4 + wid] = w2; }
      }
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
              double* o = p.spart + ((long long)(cur.n + sm) * p.nparts + cur.tn) * 3;
              o[0] = cnt; o[1] = (double)shift + md; o[2] = S2 - cnt * md * md;
            }
          } else {
            constexpr double cnt = 128.0 * NW;
            const double S1 = (q[0] + q[1]) + (q[2] + q[3]), S2 = (q[4] + q[5]) + (q[6] + q[7]);
            const double md = S1 / cnt;
            const int tin = (cur.y0 / TH) * p.tpi_x + cur.x0 / TW;
            double* o = p.spart + ((long long)cur.n * p.nparts + tin * p.ntn + cur.tn) * 3;
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
              double* o = p.nf.part + ((long long)(cur.n + sm) * p.nparts + cur.tn) * 2;
              o[0] = q[2 * sm] + q[2 * sm + 1]; o[1] = q[4 + 2 * sm] + q[4 + 2 * sm + 1];
            }
          } else {
            const int tin = (cur.y0 / TH) * p.tpi_x + cur.x0 / TW;
            double* o = p.nf.part + ((long long)cur.n * p.nparts + tin * p.ntn + cur.tn) * 2;
            o[0] = (q[0] + q[1]) + (q[2] + q[3]); o[1] = (q[4] + q[5]) + (q[6] + q[7]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
      if (more && nxt.tn != cur.tn) load_bias(nxt.tn);
      ++k;
      D3_STAMP();  // epilogue done
    }
    cur = nxt;
    c = c2;
  };
  if constexpr (NTAP % RING == 0) {
    for (int s = 0; s < total; ++s) slice(std::integral_constant<int, 0>{}, s);  // ONE instantiation of the slice body
  } else {
    for (int s = 0; s < total; s += 2) {
      slice(std::integral_constant<int, 0>{}, s);
      if (s + 1 < total) slice(std::integral_constant<int, NTAP % RING>{}, s + 1);
    }
  }
  if (p.clk && wid == 0 && lane == 0) {   // one lane per block: three no-return adds at the end of a ~100-us block life
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (t1 > clk_t0 && r1 > clk_r0) {
      atomicAdd(p.clk, t1 - clk_t0); atomicAdd(p.clk + 1, r1 - clk_r0); atomicAdd(p.clk + 2, 1ull);
    }
  }
#ifdef LG_D3_STAMPS
  if (p.stamps && wid == 0 && lane == 0) {
    p.stamps[(long long)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memtime();
    p.stamps[(long long)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace

// LG_OK: launched.  LG_ERR_UNSUPPORTED: the caller falls back to the halo-tile kernel of conv_halo.hip.
extern "C" int lg_conv_down3_nf_try(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm,
                                    int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                    size_t nf_bytes, void* stream);
extern "C" int lg_conv_down3_try(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm,
                                 int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream) {
  return lg_conv_down3_nf_try(src16, wpack, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, nullptr, 0, stream);
}
// nf (optional; data-gradient use): also the norm-backward sums of the produced gradient ([B][*nparts_out][2] doubles)
static int down3_launch(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm, int Cs, int N,
                        void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf, size_t nf_bytes,
                        const float* nstats, float nalpha, void* stream, const void* g16 = nullptr, const float* bcoef = nullptr);
extern "C" int lg_conv_down3_supported(int B, int Hm, int Wm, int Cs, int N);
extern "C" int lg_conv_down3_nf_try(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm,
                                    int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf,
                                    size_t nf_bytes, void* stream) {
  return down3_launch(src16, wpack, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, nf, nf_bytes, nullptr, 0.f, stream);
}
// The forward pass fed with the RAW bf16 output z16 of the layer below and its statistics records (NORM form of the kernel):
// moments of the produced map are always fused (spart must hold them) — LG_ERR_UNSUPPORTED otherwise and outside the kernel's tiling.
extern "C" int lg_conv_down3_zn_try(const void* z16, const float* zstats, float alpha, const void* wpack, const float* bias, void* out16,
                                    int B, int Hm, int Wm, int Cs, int N, void* spart, size_t spart_bytes, int* nparts_out, void* stream) {
  if (nparts_out) *nparts_out = 0;
  if (!zstats || !spart || !nparts_out || N % 128 != 0 || lg_env_flag("LG_NO_D3_NORM")) return LG_ERR_UNSUPPORTED;
  return down3_launch(z16, wpack, bias, out16, B, Hm, Wm, Cs, N, spart, spart_bytes, nparts_out, nullptr, 0, zstats, alpha, stream);
}
extern "C" int lg_conv_down3_zn_supported(int B, int Hm, int Wm, int Cs, int N) {
  return (!lg_env_flag("LG_NO_D3_NORM") && N % 128 == 0 && Hm % TH == 0 && Wm % TW == 0 && lg_conv_down3_supported(B, Hm, Wm, Cs, N)) ? 1 : 0;
}
// BWDNORM form (NORM = 2): the data gradient of a transposed conv fed with (z16, g16, coef) of its level instead of dz16 — the
// sums of the NEXT level's norm backward are always fused (nf), 8 x 16 tiles only; LG_ERR_UNSUPPORTED otherwise.
extern "C" int lg_conv_down3_bn_supported(int B, int Hm, int Wm, int Cs, int N) {
  const bool pair = Hm == 8 && Wm == 8 && B % 2 == 0;
  // N = 64 only: at N = 128 the form was built and measured slower than apply pass + conv (kernel header); with more than one column
  // tile every tile would redo the dz arithmetic of the halo it shares
  return (!lg_env_flag("LG_NO_D3_BWDNORM") && !pair && Hm % TH == 0 && Wm % TW == 0 && N == 64 &&
          lg_conv_down3_supported(B, Hm, Wm, Cs, N)) ? 1 : 0;
}
extern "C" int lg_conv_down3_bn_try(const void* z16, const void* g16, const float* bcoef, float alpha, const void* wpack, void* out16,
                                    int B, int Hm, int Wm, int Cs, int N, int* nparts_out, const LgNormFuse* nf, size_t nf_bytes,
                                    void* stream) {
  if (nparts_out) *nparts_out = 0;
  if (!g16 || !bcoef || !nf || !nparts_out || !lg_conv_down3_bn_supported(B, Hm, Wm, Cs, N)) return LG_ERR_UNSUPPORTED;
  return down3_launch(z16, wpack, nullptr, out16, B, Hm, Wm, Cs, N, nullptr, 0, nparts_out, nf, nf_bytes, nullptr, alpha, stream, g16, bcoef);
}
static int down3_launch(const void* src16, const void* wpack, const float* bias, void* out16, int B, int Hm, int Wm, int Cs, int N,
                        void* spart, size_t spart_bytes, int* nparts_out, const LgNormFuse* nf, size_t nf_bytes,
                        const float* nstats, float nalpha, void* stream, const void* g16, const float* bcoef) {
  if (nparts_out) *nparts_out = 0;
  static int off = -1;
  if (off < 0) off = lg_env_flag("LG_NO_DOWN3") ? 1 : 0;  // A/B switch
  if (off || !src16 || !wpack || !out16) return LG_ERR_UNSUPPORTED;
  const bool pair = Hm == 8 && Wm == 8 && B % 2 == 0;  // 8 x 8 maps: a tile = two samples side by side
  static int no64 = -1;
  if (no64 < 0) no64 = lg_env_flag("LG_NO_DOWN3_N64") ? 1 : 0;
  const bool n64 = N % 128 != 0 && N % 64 == 0 && !pair && !no64;  // 64-column tiles (2 x 2 waves)
  if (((Hm % TH || Wm % TW) && !pair) || Cs % KC || (N % 128 && !n64) || B <= 0) return LG_ERR_UNSUPPORTED;
  if ((long long)4 * Hm * Wm * Cs * 2 * 2 >= (1ll << 31)) return LG_ERR_UNSUPPORTED;  // buffer descriptor: two samples below the OOB offset
  D3Params p{};
  p.src = (const __bf16*)src16; p.wp = (const char*)wpack; p.bias = bias; p.out = (__bf16*)out16;
  p.B = B; p.Hm = Hm; p.Wm = Wm; p.Hs = 2 * Hm; p.Ws = 2 * Wm; p.Cs = Cs; p.N = N; p.N32 = N / 32; p.KB = Cs / 16;
  p.tpi_x = pair ? 1 : Wm / TW; p.tpi = pair ? 1 : p.tpi_x * (Hm / TH); p.ntn = n64 ? N / 64 : N / 128;
  const long long nitems = (long long)(pair ? B / 2 : B) * p.tpi * p.ntn;
  if (nitems <= 0 || nitems >= (1ll << 30)) return LG_ERR_UNSUPPORTED;
  p.nitems = (int)nitems; p.nparts = p.tpi * p.ntn;
  { static int stg = -1; if (stg < 0) { const char* e = getenv("LG_D3_STAGGER"); stg = e ? atoi(e) : 6; } p.stagger = stg; }
  const bool fuse = nf && nf->z && nf->stats && nf->part && nparts_out && (size_t)B * p.nparts * 2 * sizeof(double) <= nf_bytes;
  const bool stats = !fuse && spart && nparts_out && (size_t)B * p.nparts * 3 * sizeof(double) <= spart_bytes;
  // moments asked for (the caller may then write z as bf16 ONLY) but the workspace cannot hold this tiling's records: decline, so
  // that the dispatch chain tries the next kernel instead of launching without them (lg_conv_fwd_stats_fused's promise)
  if (!nf && spart && nparts_out && !stats) return LG_ERR_UNSUPPORTED;
  p.spart = stats ? (double*)spart : nullptr;
  if (fuse) p.nf = *nf;
  if (nstats && (!stats || n64 || pair)) return LG_ERR_UNSUPPORTED;
  if (g16 && (!fuse || pair || nstats || !n64 || p.ntn != 1)) return LG_ERR_UNSUPPORTED;
  p.nstats = nstats; p.nalpha = nalpha;
  p.gsrc = (const __bf16*)g16; p.bcoef = bcoef;
  p.clk = lg_clock_census();
  { static int lo = -1; if (lo < 0) lo = lg_env_flag("LG_D3_LDS_ORDER") ? 1 : 0; p.lds_order = lo; }   // (cached per call site)
  { static int ol = -1; if (ol < 0) { const char* e = getenv("LG_D3_LISTS"); ol = (e && *e >= '0' && *e <= '2') ? *e - '0' : 0; } p.lists = ol; }
  { static int sm = -1; if (sm < 0) sm = lg_env_flag("LG_D3_SLICE_MAJOR") ? 1 : 0; p.slice_major = (sm && !pair && !nstats && !g16) ? 1 : 0; }   // layout experiment
  static int bpc = 0;   // resident blocks per CU
  if (!bpc) {
    bpc = 2;
    if (const char* e = getenv("LG_D3_BLOCKS_PER_CU")) bpc = atoi(e) > 0 ? atoi(e) : 2;  // probe: 1 = a lone wave per SIMD
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_down3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_down3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<true, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<true>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<false, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<true>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<false, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<true>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<true, false, false, 64>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<false, false, false, 64>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<false, true, false, 64>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<true, false, false, 128, 1>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>((conv_down3_kernel<false, true, false, 64, 2>)), hipFuncAttributeMaxDynamicSharedMemorySize, D3L<false>::LDS_BYTES);
  }
  const int nblk = bpc * lg_grid_cus();   // every block resident from the start, also beside communication kernels (runtime.hip)
  const int grid = p.nitems < nblk ? p.nitems : nblk;
  hipStream_t st = (hipStream_t)stream;
  constexpr int LDS0 = D3L<false>::LDS_BYTES, LDS1 = D3L<true>::LDS_BYTES;
  if (nstats) {
    hipLaunchKernelGGL((conv_down3_kernel<true, false, false, 128, 1>), dim3(grid), dim3(256), LDS0, st, p);
  } else if (g16) {
    hipLaunchKernelGGL((conv_down3_kernel<false, true, false, 64, 2>), dim3(grid), dim3(256), LDS0, st, p);
  } else if (pair) {
    if (fuse) hipLaunchKernelGGL((conv_down3_kernel<false, true, true>), dim3(grid), dim3(256), LDS1, st, p);
    else if (stats) hipLaunchKernelGGL((conv_down3_kernel<true, false, true>), dim3(grid), dim3(256), LDS1, st, p);
    else hipLaunchKernelGGL((conv_down3_kernel<false, false, true>), dim3(grid), dim3(256), LDS1, st, p);
  } else if (n64) {
    if (fuse) hipLaunchKernelGGL((conv_down3_kernel<false, true, false, 64>), dim3(grid), dim3(256), LDS0, st, p);
    else if (stats) hipLaunchKernelGGL((conv_down3_kernel<true, false, false, 64>), dim3(grid), dim3(256), LDS0, st, p);
    else hipLaunchKernelGGL((conv_down3_kernel<false, false, false, 64>), dim3(grid), dim3(256), LDS0, st, p);
  } else if (fuse) hipLaunchKernelGGL((conv_down3_kernel<false, true>), dim3(grid), dim3(256), LDS0, st, p);
  else if (stats) hipLaunchKernelGGL(conv_down3_kernel<true>, dim3(grid), dim3(256), LDS0, st, p);
  else hipLaunchKernelGGL(conv_down3_kernel<false>, dim3(grid), dim3(256), LDS0, st, p);
  LG_CHECK_LAUNCH("lg_conv_down3");
  lg_note_kernel(g16 ? "conv_down3_kernel<NW=64,BWDNORM>" : nstats ? "conv_down3_kernel<NW=128,NORM>" : pair ? "conv_down3_kernel<PAIR>" : n64 ? "conv_down3_kernel<NW=64>" : "conv_down3_kernel<NW=128>");
  if (stats || fuse) *nparts_out = p.nparts;
  return LG_OK;
}

extern "C" int lg_conv_down3_supported(int B, int Hm, int Wm, int Cs, int N) {
  const bool pair = Hm == 8 && Wm == 8 && B % 2 == 0;
  const bool n64 = N % 128 != 0 && N % 64 == 0 && !pair && !lg_env_flag("LG_NO_DOWN3_N64");
  return (!lg_env_flag("LG_NO_DOWN3") && B > 0 && ((Hm % TH == 0 && Wm % TW == 0) || pair) && Cs % KC == 0 && (N % 128 == 0 || n64)) ? 1 : 0;
}
