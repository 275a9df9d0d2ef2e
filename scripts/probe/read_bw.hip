// standalone probe: achievable HBM READ bandwidth of plain streaming kernels on this box (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UNR>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ x, float* __restrict__ out, unsigned n4) {
  const unsigned stride = gridDim.x * 256 * UNR;
  f32x4 s = {0, 0, 0, 0};
  for (unsigned i0 = blockIdx.x * 256 * UNR + threadIdx.x; i0 < n4; i0 += stride) {
    f32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = (i0 + u * 256 < n4) ? x[i0 + u * 256] : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int UNR>
void run(const f32x4* x, float* out, unsigned n4, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rd<UNR>, dim3(blocks), dim3(256), 0, 0, x, out, n4);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(rd<UNR>, dim3(blocks), dim3(256), 0, 0, x, out, n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("UNR=%d blocks=%5d : %.2f TB/s\n", UNR, blocks, (double)n4 * 16 * 10 / (ms * 1e-3) / 1e12);
}
int main() {
  const unsigned n4 = 1u << 26;  // 1 GiB
  f32x4* x; float* out;
  hipMalloc(&x, (size_t)n4 * 16); hipMalloc(&out, (size_t)32768 * 256 * sizeof(float));  // one float per thread of the largest grid below
  hipMemset(x, 0, (size_t)n4 * 16);
  for (int blocks : {2048, 8192, 32768}) {
    run<1>(x, out, n4, blocks); run<2>(x, out, n4, blocks); run<4>(x, out, n4, blocks); run<8>(x, out, n4, blocks);
  }
  return 0;
}
