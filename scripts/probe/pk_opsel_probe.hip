// Round-5 probe: is `v_pk_add_f32 vD, vD, vS op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` (both results subtract the HIGH register of the
// VGPR pair vS — the form the non-deterministic BWDNORM build of conv_down3 used for `g' - m1`) reliable when another
// wave of the same SIMD runs MFMAs?  Blocks in the lower half of the grid check the instruction in a loop against the scalar
// expression; blocks in the upper half (the second block of every CU) run an MFMA / LDS / VALU stream for the same time.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probe/bin/pk_opsel_probe scripts/probe/pk_opsel_probe.hip ; run: pk_opsel_probe [iters] [mode]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void probe(unsigned* err, unsigned* rec, int iters, int mode, int hog_all) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x;
  const bool hog = hog_all ? false : (int)blockIdx.x >= (int)gridDim.x / 2;
  if (hog) {   // the partner wave: what the other block of the CU does during a tap phase
    f32x16 acc = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (tid + i)); b[i] = (__bf16)(0.002f * (tid - i)); }
    float* s = reinterpret_cast<float*>(smem);
    s[tid] = tid;
    __syncthreads();
    for (int it = 0; it < iters * 6; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
      a[it & 7] = (__bf16)s[(tid + it) & 255];
    }
    if (acc[0] == 123.456f) err[1] = 1;   // keep the loop
    return;
  }
  // checker: p = (p0, p1); bm = (b, m1); expected (p0 - m1, p1 - m1)
  unsigned bad = 0;
  float x = 0.37f + 0.001f * tid + 0.01f * blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    x = x * 1.0001f + 0.013f;
    if (x > 4.f) x -= 3.9f;
    const float g0 = x, g1 = -x * 0.7f, ag0 = 0.3f * g0, y0 = (it & 1) ? 0.5f : -0.5f, y1 = (it & 2) ? 0.5f : -0.5f;
    f32x2 bm = {0.15f, 0.0004475f + 1e-6f * (it & 15)};
    f32x2 ml = {0.0005129f, 2.89e-12f};
    f32x2 p = {(y0 > 0.f) ? g0 : ag0, g1};   // the select (v_cndmask) in front of the packed subtraction, as in lg_bwdnorm8
    f32x2 q = {g1, g0}, r = {0.25f, -0.125f};
    // fixed physical registers (64-bit inline-asm operands came back with both halves bound to ONE register): v[40:41] = p, v[42:43] = q,
    // v[44:45] = r, v[46:47] = (b, m1), v[48:49] = (m2, m1l)
    float o0, o1, o2;
    if (mode == 0) {   // the failing build's packed operations: m1 / m1l = the HIGH register of their pairs, subtracted from both results
      asm volatile(
          "v_mov_b32 v40, %[p0]\n\tv_mov_b32 v41, %[p1]\n\tv_mov_b32 v42, %[q0]\n\tv_mov_b32 v43, %[q1]\n\tv_mov_b32 v44, %[r0]\n\tv_mov_b32 v45, %[r1]\n\t"
          "v_mov_b32 v46, %[b]\n\tv_mov_b32 v47, %[m1]\n\tv_mov_b32 v48, %[m2]\n\tv_mov_b32 v49, %[m1l]\n\t"
          "v_pk_mul_f32 v[42:43], v[48:49], v[42:43] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_mul_f32 v[44:45], v[46:47], v[44:45] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 v[40:41], v[40:41], v[46:47] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 v[40:41], v[40:41], v[48:49] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "s_nop 0\n\t"
          "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n\t"
          "s_nop 1\n\t"
          "v_mov_b32 %[o0], v40\n\tv_mov_b32 %[o1], v41\n\tv_mov_b32 %[o2], v44\n\t"
          : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2)
          : [p0] "v"(p[0]), [p1] "v"(p[1]), [q0] "v"(q[0]), [q1] "v"(q[1]), [r0] "v"(r[0]), [r1] "v"(r[1]), [b] "v"(bm[0]), [m1] "v"(bm[1]), [m2] "v"(ml[0]), [m1l] "v"(ml[1])
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    } else {           // the same arithmetic with m1 / m1l in the LOW register of their pairs (the form every working build used)
      asm volatile(
          "v_mov_b32 v40, %[p0]\n\tv_mov_b32 v41, %[p1]\n\tv_mov_b32 v42, %[q0]\n\tv_mov_b32 v43, %[q1]\n\tv_mov_b32 v44, %[r0]\n\tv_mov_b32 v45, %[r1]\n\t"
          "v_mov_b32 v46, %[m1]\n\tv_mov_b32 v47, %[b]\n\tv_mov_b32 v48, %[m1l]\n\tv_mov_b32 v49, %[m2]\n\t"
          "v_pk_mul_f32 v[42:43], v[48:49], v[42:43] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_mul_f32 v[44:45], v[46:47], v[44:45] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 v[40:41], v[40:41], v[46:47] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 v[40:41], v[40:41], v[48:49] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "s_nop 0\n\t"
          "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n\t"
          "s_nop 1\n\t"
          "v_mov_b32 %[o0], v40\n\tv_mov_b32 %[o1], v41\n\tv_mov_b32 %[o2], v44\n\t"
          : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2)
          : [p0] "v"(p[0]), [p1] "v"(p[1]), [q0] "v"(q[0]), [q1] "v"(q[1]), [r0] "v"(r[0]), [r1] "v"(r[1]), [b] "v"(bm[0]), [m1] "v"(bm[1]), [m2] "v"(ml[0]), [m1l] "v"(ml[1])
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    }
    (void)y1;
    const float s0 = (y0 > 0.f) ? g0 : ag0;
    const float q0 = -(ml[0] * g1), q1 = -(ml[0] * g0);
    const float e0 = ((s0 - bm[1]) - ml[1]) + q0, e1 = ((g1 - bm[1]) - ml[1]) + q1;
    if (__builtin_bit_cast(unsigned, e0) != __builtin_bit_cast(unsigned, o0) || __builtin_bit_cast(unsigned, e1) != __builtin_bit_cast(unsigned, o1)) {
      ++bad;
      const unsigned ix = atomicAdd(err + 2, 1u);
      if (ix < 64) {
        unsigned* o = rec + ix * 8;
        o[0] = blockIdx.x; o[1] = tid; o[2] = it; o[3] = __builtin_bit_cast(unsigned, e0); o[4] = __builtin_bit_cast(unsigned, o0);
        o[5] = __builtin_bit_cast(unsigned, e1); o[6] = __builtin_bit_cast(unsigned, o1); o[7] = __builtin_bit_cast(unsigned, s0);
      }
    }
    x += 1e-7f * (o0 + o1 + o2);
  }
  if (bad) atomicAdd(err, bad);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  unsigned *err, *rec;
  hipMalloc(&err, 16); hipMalloc(&rec, 64 * 8 * 4);
  for (int hog_all = 0; hog_all < 2; ++hog_all)
    for (int mode = 0; mode < 2; ++mode) {
      hipMemset(err, 0, 16); hipMemset(rec, 0, 64 * 8 * 4);
      hipLaunchKernelGGL(probe, dim3(512), dim3(256), 1024, 0, err, rec, iters, mode, hog_all);
      hipError_t e = hipDeviceSynchronize();
      unsigned h[4], r[64 * 8];
      hipMemcpy(h, err, 16, hipMemcpyDeviceToHost); hipMemcpy(r, rec, sizeof(r), hipMemcpyDeviceToHost);
      printf("%s, %s: %s, wrong results %u of %lld\n", hog_all ? "checkers on both blocks of a CU" : "checker + MFMA partner per SIMD",
             mode ? "low-register broadcast (op_sel_hi:[1,0])" : "HIGH-register broadcast (op_sel:[0,1])", hipGetErrorString(e), h[0],
             (long long)iters * 256 * (hog_all ? 512 : 256));
      for (unsigned i = 0; i < (h[2] < 6 ? h[2] : 6); ++i) {
        const unsigned* o = r + i * 8;
        printf("   block %u thread %u (lane %u) iteration %u: low expected %.9g got %.9g (selected value %.9g) | high expected %.9g got %.9g\n", o[0], o[1], o[1] & 63, o[2],
               *(const float*)&o[3], *(const float*)&o[4], *(const float*)&o[7], *(const float*)&o[5], *(const float*)&o[6]);
      }
    }
  return 0;
}
