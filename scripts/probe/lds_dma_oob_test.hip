#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00027000);
  // lanes 0..47 in range, lanes 48..63 out of range
  const unsigned voff = lane < 48 ? (unsigned)((63 - lane) * 16) : 0x40000000u;   // reversed order: per-lane source
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *src, *out; hipMalloc(&src, 64 * 16); hipMalloc(&out, 512 * 4);
  unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = i; hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 64 * 16);
  unsigned o[512]; hipMemcpy(o, out, 2048, hipMemcpyDeviceToHost);
  printf("lane0: %u %u %u %u (expect 252..255)\n", o[0], o[1], o[2], o[3]);
  printf("lane47: %u %u %u %u (expect 64..67)\n", o[47*4], o[47*4+1], o[47*4+2], o[47*4+3]);
  printf("lane48: %x %x (0 if OOB lanes write zeros, deadbeef if skipped)\n", o[48*4], o[48*4+1]);
  printf("beyond wave: %x\n", o[256]);
  return 0;
}
