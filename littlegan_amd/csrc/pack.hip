// Weight packing: fp32 master kernels [5][5][Cb][Cs] (TF HWIO for Conv2D with Cb=in,Cs=out;
// TF HWOI for Conv2DTranspose with Cb=out,Cs=in — /root/reference/model.py:15,39-40,86-87)
// -> MFMA B-operand images in the compute dtype, FRAGMENT-ORDERED (see fo_decode): [tap][n/32][k/EPB][lane][EPL].
//   down pack: n = Cs index, k = Cb index   (conv fwd / convT dgrad)
//              Cb == 3: patch form [5][npad(Cs)][16] row-major, k = kx*3+c, k=15 zero
//   up pack  : n = Cb index, k = Cs index   (convT fwd / conv dgrad)
// One pack per layer per step (weights change every step); ~90 MB of traffic for the whole model.
#include "lg_common.h"

extern "C" int lg_npad(int n);

namespace {

// Fragment order: one 1-KiB block = the B operand of ONE MFMA (32 n x 16 k bf16, or 32 n x 8 k f32), stored in lane
// order (lane = 32*h + r holds n = 32*n32 + r, k = EPB*kb + EPL*h + j, j < EPL) so that a wave fetches it with one
// fully coalesced global_load_dwordx4.  Layout [tap][n32][kb][lane][EPL].
template <typename T>
__device__ __forceinline__ void fo_decode(long long i, int npad, int K, int& t, int& n, int& k) {
  constexpr int EPL = 16 / (int)sizeof(T);  // elements per lane (16 B)
  constexpr int EPB = 2 * EPL;              // k elements per block
  const int j = (int)(i % EPL);
  long long rem = i / EPL;
  const int lane = (int)(rem % 64); rem /= 64;
  const int KB = K / EPB;
  const int kb = (int)(rem % KB); rem /= KB;
  const int N32 = npad / 32;
  const int n32 = (int)(rem % N32);
  t = (int)(rem / N32);
  n = n32 * 32 + (lane & 31);
  k = kb * EPB + (lane >> 5) * EPL + j;
}

template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, T* __restrict__ down, T* __restrict__ up, int Cb, int Cs,
                            int npad_s, int npad_b, long long n_down, long long n_up, float* __restrict__ raw, long long n_raw) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_down + n_up + n_raw; i += stride) {
    if (i >= n_down + n_up) {  // 3-channel layers: the verbatim fp32 copy (same launch: no runtime copy kernel in the step)
      raw[i - n_down - n_up] = w[i - n_down - n_up];
    } else if (i < n_down) {
      float v = 0.f;
      if (Cb == 3) {  // [5][npad_s][16]
        const int j = (int)(i % 16);
        const long long rem = i / 16;
        const int n = (int)(rem % npad_s), ky = (int)(rem / npad_s);
        if (j < 15 && n < Cs) v = w[((long long)(ky * 5 + j / 3) * 3 + (j % 3)) * Cs + n];
      } else {  // fragment-ordered [25][npad_s/32][Cb/EPB][64 lanes][EPL]: (n, k) = (cs index, cb index)
        int t, n, k;
        fo_decode<T>(i, npad_s, Cb, t, n, k);
        if (n < Cs) v = w[((long long)t * Cb + k) * Cs + n];
      }
      down[i] = (T)v;
    } else {  // fragment-ordered [25][npad_b/32][Cs/EPB][64][EPL]: (n, k) = (cb index, cs index)
      const long long u = i - n_down;
      int t, n, k;
      fo_decode<T>(u, npad_b, Cs, t, n, k);
      float v = 0.f;
      if (n < Cb) v = w[((long long)t * Cb + n) * Cs + k];
      up[u] = (T)v;
    }
  }
}

inline long long down_elems(int cb, int cs) {
  return cb == 3 ? 5ll * lg_npad(cs) * 16 : 25ll * lg_npad(cs) * cb;
}
inline long long up_elems(int cb, int cs) { return 25ll * lg_npad(cb) * cs; }

}  // namespace

// byte offset of the up pack inside a layer pack (down pack is at 0); both 256-B aligned
extern "C" size_t lg_conv_pack_up_offset(int cb, int cs, int dtype) {
  const size_t esz = dtype == LG_DT_F32 ? 4 : 2;
  return (down_elems(cb, cs) * esz + 255) / 256 * 256;
}

// cb == 3 layers also carry a verbatim fp32 copy of the kernel [5][5][3][cs] for the VALU kernels of n3_kernels.hip
extern "C" size_t lg_conv_pack_raw_offset(int cb, int cs, int dtype) {
  const size_t esz = dtype == LG_DT_F32 ? 4 : 2;
  return lg_conv_pack_up_offset(cb, cs, dtype) + (up_elems(cb, cs) * esz + 255) / 256 * 256;
}

extern "C" size_t lg_conv_pack_bytes(int cb, int cs, int dtype) {
  return lg_conv_pack_raw_offset(cb, cs, dtype) + (cb == 3 ? ((size_t)75 * cs * 4 + 255) / 256 * 256 : 0);
}

extern "C" int lg_conv_pack(const float* w, void* pack, int cb, int cs, int dtype, void* stream) {
  LG_CHECK_ARG(w && pack, "lg_conv_pack: null pointer");
  LG_CHECK_ARG(cb > 0 && cs > 0 && (cb == 3 || cb % 32 == 0) && cs % 32 == 0,
               "lg_conv_pack: unsupported channels cb=%d cs=%d", cb, cs);
  LG_CHECK_ARG(dtype == LG_DT_F32 || dtype == LG_DT_BF16, "lg_conv_pack: bad dtype %d", dtype);
  const long long nd = down_elems(cb, cs), nu = up_elems(cb, cs);
  char* up = (char*)pack + lg_conv_pack_up_offset(cb, cs, dtype);
  const long long nr = cb == 3 ? 75ll * cs : 0;
  float* raw = cb == 3 ? (float*)((char*)pack + lg_conv_pack_raw_offset(cb, cs, dtype)) : nullptr;
  const int blocks = (int)((nd + nu + nr + 255) / 256 < 4096 ? (nd + nu + nr + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LG_DT_F32)
    hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, st, w, (float*)pack, (float*)up, cb, cs,
                       lg_npad(cs), lg_npad(cb), nd, nu, raw, nr);
  else
    hipLaunchKernelGGL(pack_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, w, (__bf16*)pack, (__bf16*)up, cb, cs,
                       lg_npad(cs), lg_npad(cb), nd, nu, raw, nr);
  LG_CHECK_LAUNCH("lg_conv_pack");
  return LG_OK;
}
