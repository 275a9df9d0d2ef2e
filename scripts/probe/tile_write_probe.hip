// Round 5: how fast can convT4 forward's OUTPUT pattern be written at all?  512 persistent workgroups of 256 threads write the 128 x 128 x 32 bf16 map
// of B samples as 16 x 32-pixel tiles (16 rows of 2 KB, 8 KB apart: what conv_up3<64,32>'s row sweep does), 8 x 16-byte pieces per thread and tile,
// against the same bytes written as one contiguous stream; optionally with the layer's halo READS (10 x 18 source pixels x 128 B per tile) beside them.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probe/bin/tile_write_probe scripts/probe/tile_write_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void tiles(u32x4* out, const u32x4* src, int B, int mode, unsigned* sink) {
  const int tid = threadIdx.x, G = gridDim.x;
  const int ntiles = B * 8 * 4;                     // 64 x 64 source -> 8 x 4 tiles of 8 x 16 source pixels
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  unsigned acc = 0;
  for (int t = blockIdx.x; t < ntiles; t += G) {
    const int n = t >> 5, tt = t & 31, ty = tt >> 2, tx = tt & 3;
    if (mode & 2) {   // the halo reads: 180 pixels x 8 pieces of 16 B = 1440 pieces, 6 per thread (source [B][64][64][64] bf16)
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int q = tid + u * 256;
        if (q < 1440) {
          const int px = q >> 3, pc = q & 7, hy = px / 18, hx = px - hy * 18;
          const int sy = ty * 8 - 1 + hy, sx = tx * 16 - 1 + hx;
          if ((unsigned)sy < 64u && (unsigned)sx < 64u) { const u32x4 r = src[((long long)(n * 64 + sy) * 64 + sx) * 8 + pc]; acc += r[0]; }
        }
      }
      v[1] = acc;
    }
    if (mode & 1) {   // tile pattern: piece q8 of thread: output pixel o = tid / 4 + q8 * 64 (row o >> 5, column o & 31), 16-byte column tid & 3
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) {
        const int o = (tid >> 2) + q8 * 64, j = tid & 3;
        const long long pix = ((long long)(n * 128 + ty * 16 + (o >> 5)) * 128 + tx * 32 + (o & 31));
        __builtin_nontemporal_store(v, out + pix * 4 + j);
      }
    } else {          // the same 32 KB per tile as one contiguous run
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) __builtin_nontemporal_store(v, out + (long long)t * 2048 + q8 * 256 + tid);
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;
  u32x4 *out, *src; unsigned* sink;
  const size_t ob = (size_t)B * 128 * 128 * 32 * 2, sb = (size_t)B * 64 * 64 * 64 * 2;
  hipMalloc(&out, ob); hipMalloc(&src, sb); hipMalloc(&sink, 16);
  hipMemset(src, 1, sb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"contiguous stream, writes only", "tile pattern, writes only", "contiguous stream + halo reads", "tile pattern + halo reads"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tiles, dim3(512), dim3(256), 0, 0, out, src, B, mode, sink);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiles, dim3(512), dim3(256), 0, 0, out, src, B, mode, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0, bytes = (double)ob + ((mode & 2) ? (double)sb * 1.41 : 0.0);
    printf("B=%d  %-34s %7.1f us  %.2f TB/s (%.0f MB)\n", B, names[mode], us, bytes / us / 1e6, bytes / 1e6);
  }
  return 0;
}
