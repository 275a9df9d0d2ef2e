// Run-time services of the library that are not part of the step's arithmetic:
//   * the CU budget of the persistent kernels (lg_set_reserved_cus / lg_grid_cus): under data parallelism RCCL's ring
//     kernels run on a side stream WHILE the backward convs run (littlegan_amd/dist.py); a persistent grid sized to every CU
//     assumes all of its blocks are resident from t = 0, and a block displaced by a communication workgroup becomes a
//     serial tail of one block life.  Persistent launchers size their grids to lg_grid_cus() = CUs - reserved.
//   * lg_contention_probe: the one-GPU rehearsal of that situation (bench.py --dp-contention K): K workgroups that stream
//     read-add-write over a gradient-sized range, i.e. what one RCCL ring step does to the CUs it occupies.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "lg_common.h"
#include "../../include/littlegan_hip.h"

static int g_reserved_cus = 0;
static unsigned long long* g_clock_census = nullptr;

static int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
      cus = pr.multiProcessorCount;
  }
  return cus;
}

extern "C" int lg_device_cus(void) { return device_cus(); }

extern "C" int lg_set_reserved_cus(int n) {
  LG_CHECK_ARG(n >= 0 && n < device_cus(), "lg_set_reserved_cus: %d of %d CUs", n, device_cus());
  g_reserved_cus = n;
  return LG_OK;
}

extern "C" int lg_grid_cus(void) { return device_cus() - g_reserved_cus; }

// One switch table for the whole library: every A/B environment switch is read ONCE, at its first use, through this
// function, so a *_supported query and the launch it promises can never see two different values.
extern "C" int lg_env_flag(const char* name) {
  struct Slot { const char* name; int val; };
  static Slot slots[64];
  static int n = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n; ++i)
    if (slots[i].name == name || strcmp(slots[i].name, name) == 0) return slots[i].val;
  const char* e = getenv(name);
  const int v = (e && *e) ? 1 : 0;   // set AND non-empty (round 4; `LG_NO_X=` no longer switches anything)
  if (n >= 64) {   // a full table would silently re-read the environment on every call: the "read once" promise would be gone
    fprintf(stderr, "littlegan_hip: lg_env_flag table full (64 switches) at %s — enlarge it\n", name);
    abort();
  }
  slots[n].name = name; slots[n].val = v; ++n;
  return v;
}

namespace {

__global__ __launch_bounds__(512) void contention_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4,
                                                         int passes) {
  // dst = 0.5 dst + src: bounded whatever the number of passes; 16-B accesses, grid-stride
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (int p = 0; p < passes; ++p)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const f32x4 a = reinterpret_cast<const f32x4*>(src)[i];
      f32x4 d = reinterpret_cast<f32x4*>(dst)[i];
      d = d * 0.5f + a;
      reinterpret_cast<f32x4*>(dst)[i] = d;
    }
}

// One wave that spins for `spin_ticks` of the 100 MHz clock and reports d s_memtime / d s_memrealtime: launched on the COMPUTE
// stream right behind a kernel, it reads the clock the chip holds at that point of the step (the power controller moves the
// clock on a millisecond scale, the sample takes 10-20 us).  out2[0] += d memtime, out2[1] += d realtime (one lane, plain adds:
// samples on one stream run in order).
__global__ __launch_bounds__(64) void clock_sample_kernel(unsigned long long* __restrict__ out2, unsigned spin_ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  for (int i = 0; i < 100000 && r1 - r0 < spin_ticks; ++i) {   // bounded spin: every wave reaches the exit
    __builtin_amdgcn_s_sleep(8);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && t1 > t0) { out2[0] += t1 - t0; out2[1] += r1 - r0; out2[2] += 1; }
}

}  // namespace

extern "C" int lg_contention_probe(float* dst, const float* src, long long n, int workgroups, int threads, int passes,
                                   void* stream) {
  LG_CHECK_ARG(dst && src && n >= 4 && n % 4 == 0, "lg_contention_probe: n must be a positive multiple of 4");
  LG_CHECK_ARG(workgroups > 0 && workgroups <= 4096 && (threads == 256 || threads == 512) && passes > 0 && passes <= 64,
               "lg_contention_probe: %d workgroups x %d threads x %d passes", workgroups, threads, passes);
  hipLaunchKernelGGL(contention_kernel, dim3(workgroups), dim3(threads), 0, (hipStream_t)stream, dst, src, n / 4, passes);
  LG_CHECK_LAUNCH("lg_contention_probe");
  return LG_OK;
}

extern "C" int lg_clock_sample(unsigned long long* out3, int spin_us, void* stream) {
  LG_CHECK_ARG(out3 && spin_us > 0 && spin_us <= 1000, "lg_clock_sample: bad args (spin_us in 1..1000)");
  hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out3, (unsigned)spin_us * 100u);
  LG_CHECK_LAUNCH("lg_clock_sample");
  return LG_OK;
}

// In-kernel clock census (bench.py's `clock` field): while a buffer is registered, every block of the persistent conv kernels that
// support it (conv_down3.hip: the dominant kernel of the step) adds, at its end, {d s_memtime, d s_memrealtime, 1} of its own life to
// buf3 — sum[0] / sum[1] x 100 MHz is the shader clock those blocks saw WHILE THEY RAN in the real step.  (Round 4 also had a probe wave
// resident on a side stream, lg_clock_probe; its one cross-check was invalid — device-wide synchronisations in the load loop waited for
// the resident wave, so 99.5 % of its windows saw an idle chip — and it was removed in round 5.)  nullptr = off (the default; two scalar
// instructions per block remain).
extern "C" int lg_set_clock_census(unsigned long long* buf3) { g_clock_census = buf3; return LG_OK; }
extern "C" unsigned long long* lg_clock_census(void) { return g_clock_census; }
