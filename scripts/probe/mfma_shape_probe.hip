// Probe (round 3): what would the 16x16x32 bf16 MFMA shape buy in the tap loops of conv_down3 / conv_up3 / conv_up4?
// Both kernels below run the SAME work per step with the SAME operand traffic as those loops — a 32-channel x 128-pixel wave tile,
// pixel fragments re-read from LDS with ds_read_b128 (1 KB per 32 KFLOP... i.e. 8 reads per step), weight fragments in registers —
// once as 8 x v_mfma_f32_32x32x16_bf16 and once as 16 x v_mfma_f32_16x16x32_bf16, on random data, two waves per SIMD.
// Reports TFLOP/s and the clock the chip holds (s_memtime / s_memrealtime) for each.   hipcc --offload-arch=gfx950 -O3 -o probe ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int steps) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) reinterpret_cast<u32x4*>(lds)[i] = src[(blockIdx.x * 2048 + i) & 0xffff];
  __syncthreads();
  bf16x8 wf[4];
  for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(bf16x8, src[(lane + 64 * i + tid) & 0xffff]);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const char* base = lds + lane * 16;
  if constexpr (SHAPE == 64) {   // 32x32x16 with every pixel fragment used TWICE (a 64-channel x 64-pixel wave tile): half the LDS reads
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int s = 0; s < steps; ++s) {
      const int o = (s & 3) * 8192;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + o + (k * 2 + i) * 1024);
          acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k], a, acc[2 * i], 0, 0, 0);
          acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 + k], a, acc[2 * i + 1], 0, 0, 0);
        }
      }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
    out[blockIdx.x * 256 + tid] = t;
  } else if constexpr (SHAPE == 0) {   // no LDS reads at all: operands in registers (the matrix pipe's own ceiling on random data)
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base), a1 = *reinterpret_cast<const bf16x8*>(base + 1024);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k], (i & 1) ? a1 : a0, acc[i], 0, 0, 0);
      }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
    out[blockIdx.x * 256 + tid] = t;
  } else if constexpr (SHAPE == 32) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int s = 0; s < steps; ++s) {
      const int o = (s & 3) * 8192;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + o + (k * 4 + i) * 1024);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k], a, acc[i], 0, 0, 0);   // (compile-time index: a runtime-indexed register array goes to scratch)
        }
      }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
    out[blockIdx.x * 256 + tid] = t;
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int s = 0; s < steps; ++s) {
      const int o = (s & 3) * 8192;
#pragma unroll
      for (int i = 0; i < 8; ++i) {   // 8 pixel fragments (32 k x 16 pixels), each against the 2 weight fragments (16 channels x 32 k)
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + o + i * 1024);
        acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], a, acc[2 * i], 0, 0, 0);
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], a, acc[2 * i + 1], 0, 0, 0);
      }
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) t += acc[i][e];
    out[blockIdx.x * 256 + tid] = t;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const u32x4* src, float* out, unsigned long long* clk, int steps, const char* name) {
  const int grid = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  double elapsed = 0.0;
  while (elapsed < 2500.0) {   // >= 2.5 s of back-to-back launches: the clock the chip holds, not its ramp
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<SHAPE>, dim3(grid), dim3(256), 0, 0, src, out, clk, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    elapsed += ms;
  }
  std::vector<unsigned long long> h(grid * 2);
  hipMemcpy(h.data(), clk, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int b = 0; b < grid; ++b) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
  const double flops = (double)grid * 4 /*waves*/ * steps * 8.0 * 32768.0;   // 8 x (32x32x16) per step and wave either way
  const double us = ms / 20 * 1e3;
  printf("%s: %.1f us per launch, %.0f TFLOP/s, clock %.3f GHz, %.1f cycles per step and wave (8 x 32 = 256 if the pipe were the wave's alone)\n",
         name, us, flops / us / 1e6, cyc / real * 0.1, cyc / grid / steps);
}

int main() {
  u32x4* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 65536 * 16); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 512 * 2 * 8);
  std::vector<unsigned> h(65536 * 4);
  srand(7);
  for (auto& v : h) {   // random bf16 pairs in [-1, 1): sign, exponent 0x7e.. mantissa random
    unsigned a = (rand() & 0x807f) | (0x3f00 - ((rand() & 3) << 7)), b = (rand() & 0x807f) | (0x3f00 - ((rand() & 3) << 7));
    v = a | (b << 16);
  }
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int steps = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    run<32>(src, out, clk, steps, "v_mfma_f32_32x32x16_bf16, 8 per step ");
    run<16>(src, out, clk, steps, "v_mfma_f32_16x16x32_bf16, 16 per step");
    run<64>(src, out, clk, steps, "32x32x16, fragment used twice (4 reads)");
    run<0>(src, out, clk, steps, "32x32x16, operands in registers       ");
  }
  return 0;
}
