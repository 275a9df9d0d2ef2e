// Losses and optimizer of the LittleGAN step.
//   bce_heads : tf.keras.losses.binary_crossentropy (TF-1.15 backend form) + reduce_mean, as combined
//               in /root/reference/eager_trainer.py:85-102, fused with the sigmoid backward of the heads:
//                 p^ = clip(p,1e-7,1-1e-7); l = -(t log(p^+1e-7) + (1-t) log(1-p^+1e-7))
//               loss (+)= w_pr * mean_b l(t_pr, p[b,0]) + w_c * mean_{b,j} l(t_c[b,j], p[b,1+j])
//               dz = dloss/dp * p(1-p)   (gradient w.r.t. the pre-sigmoid logits)
//   l1_tanh_bwd: loss (+)= lambda*mean|t - img| ; dpre = (g_in - lambda*sign(t-img)/n) * (1-img^2)
//               (eager_trainer.py:96,101 + the tanh of model.py:87)
//   clip_adam : tf.clip_by_value (eager_trainer.py:146-148) + tf.compat.v1.train.AdamOptimizer
//               (eager_trainer.py:28-30,164-168): lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps outside sqrt,
//               one beta-power pair per optimizer kept on the device so the step replays as a graph.
#include "lg_common.h"

#define LG_BCE_EPS 1e-7f

namespace {

__global__ __launch_bounds__(1024) void bce_heads_kernel(const float* __restrict__ p, const float* __restrict__ t_c,
                                                        float t_pr, float w_pr, float w_c, float* __restrict__ loss,
                                                        float* __restrict__ dz, int B, int c, int accumulate) {
  const int J = c + 1, n = B * J;
  const float s_pr = w_pr / (float)B, s_c = w_c / (float)(B * c);
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {  // one block (the loss is ONE scalar): 1024 threads, <= 21 trips at B=512
    const int b = i / J, j = i - b * J;
    const float sc = j == 0 ? s_pr : s_c;
    float g = 0.f;
    if (sc != 0.f) {
      const float t = j == 0 ? t_pr : t_c[b * c + j - 1];
      const float pv = p[i];
      const float pc = fminf(fmaxf(pv, LG_BCE_EPS), 1.f - LG_BCE_EPS);
      acc += -sc * (t * logf(pc + LG_BCE_EPS) + (1.f - t) * logf(1.f - pc + LG_BCE_EPS));
      const bool inside = pv >= LG_BCE_EPS && pv <= 1.f - LG_BCE_EPS;
      if (inside) g = -sc * (t / (pc + LG_BCE_EPS) - (1.f - t) / (1.f - pc + LG_BCE_EPS)) * pv * (1.f - pv);
    }
    dz[i] = g;
  }
  __shared__ float sred[16];
  float red[1] = {acc};
  lg_block_sum<1>(red, sred);
  if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + red[0];
}

// partial[blk] = sum |t - img| ; dpre written
__global__ __launch_bounds__(256) void l1_tanh_bwd_kernel(const float* __restrict__ t, const float* __restrict__ img,
                                                          const float* __restrict__ g_in, float* __restrict__ dpre,
                                                          float* __restrict__ partial, long long n4, float gscale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 tv = *reinterpret_cast<const f32x4*>(t + i * 4);
    const f32x4 iv = *reinterpret_cast<const f32x4*>(img + i * 4);
    f32x4 gv = g_in ? *reinterpret_cast<const f32x4*>(g_in + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = tv[k] - iv[k];
      acc += fabsf(d);
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      o[k] = (gv[k] - gscale * sg) * (1.f - iv[k] * iv[k]);
    }
    if (dpre) *reinterpret_cast<f32x4*>(dpre + i * 4) = o;
  }
  __shared__ float sred[16];
  float red[1] = {acc};
  lg_block_sum<1>(red, sred);
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ partial, int nb, float scale,
                                                       float* __restrict__ loss, int accumulate) {
  __shared__ double sd[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += (double)partial[i];
  sd[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + (float)(sd[0] * (double)scale);
}

// state = {beta1_power, beta2_power}
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        const float* __restrict__ state, float lr, float b1, float b2,
                                                        float eps, float clip, float gscale) {
  const float lr_t = lr * sqrtf(1.f - state[1]) / (1.f - state[0]);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gv = g[i] * gscale;
    if (clip > 0.f) gv = fminf(fmaxf(gv, -clip), clip);
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv; v[i] = vv;
    w[i] -= lr_t * mv / (sqrtf(vv) + eps);
  }
}

__global__ void adam_advance_kernel(float* state, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { state[0] *= b1; state[1] *= b2; }
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float a, float b,
                                                    long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}

}  // namespace

extern "C" int lg_bce_heads_loss_fwd_bwd(const float* p, const float* t_c, float t_pr, float w_pr, float w_c,
                                         float* loss, float* dz, int B, int c, int accumulate, void* stream) {
  LG_CHECK_ARG(p && loss && dz, "lg_bce_heads_loss_fwd_bwd: null pointer");
  LG_CHECK_ARG(B > 0 && c >= 1, "lg_bce_heads_loss_fwd_bwd: bad shape B=%d c=%d", B, c);
  LG_CHECK_ARG(t_c || w_c == 0.f, "lg_bce_heads_loss_fwd_bwd: t_c is null but w_c != 0");
  hipLaunchKernelGGL(bce_heads_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, t_c, t_pr, w_pr, w_c, loss, dz, B,
                     c, accumulate);
  LG_CHECK_LAUNCH("lg_bce_heads_loss_fwd_bwd");
  return LG_OK;
}

extern "C" size_t lg_l1_workspace_bytes(void) { return 1024 * sizeof(float); }

// loss (+)= lambda*mean|t-img| ; dpre = (g_in - lambda*sign(t-img)/n)*(1-img^2)  (dpre may be null: loss only)
extern "C" int lg_l1_tanh_loss_fwd_bwd(const float* t, const float* img, const float* g_in, float* dpre, float* loss,
                                       void* workspace, size_t ws_bytes, long long n, float lambda, int accumulate,
                                       void* stream) {
  LG_CHECK_ARG(t && img && loss && workspace, "lg_l1_tanh_loss_fwd_bwd: null pointer");
  LG_CHECK_ARG(n > 0 && n % 4 == 0, "lg_l1_tanh_loss_fwd_bwd: n=%lld must be a positive multiple of 4", n);
  LG_CHECK_ARG(ws_bytes >= lg_l1_workspace_bytes(), "lg_l1_tanh_loss_fwd_bwd: workspace too small");
  const long long n4 = n / 4;
  int nb = (int)((n4 + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_tanh_bwd_kernel, dim3(nb), dim3(256), 0, st, t, img, g_in, dpre, (float*)workspace, n4,
                     lambda / (float)n);
  LG_CHECK_LAUNCH("lg_l1_tanh_loss_fwd_bwd");
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nb, lambda / (float)n, loss,
                     accumulate);
  LG_CHECK_LAUNCH("lg_l1_tanh_loss_fwd_bwd(final)");
  return LG_OK;
}

extern "C" int lg_clip_adam_update(float* w, const float* g, float* m, float* v, long long n, const float* state,
                                   float lr, float b1, float b2, float eps, float clip, float gscale, void* stream) {
  LG_CHECK_ARG(w && g && m && v && state, "lg_clip_adam_update: null pointer");
  LG_CHECK_ARG(n > 0, "lg_clip_adam_update: n=%lld", n);
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, state, lr, b1, b2,
                     eps, clip, gscale);
  LG_CHECK_LAUNCH("lg_clip_adam_update");
  return LG_OK;
}

extern "C" int lg_adam_advance(float* state, float b1, float b2, void* stream) {
  LG_CHECK_ARG(state, "lg_adam_advance: null pointer");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, b1, b2);
  LG_CHECK_LAUNCH("lg_adam_advance");
  return LG_OK;
}

// y = a*x + b*y
extern "C" int lg_axpby(float* y, const float* x, float a, float b, long long n, void* stream) {
  LG_CHECK_ARG(x && y && n > 0, "lg_axpby: bad args");
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, y, x, a, b, n);
  LG_CHECK_LAUNCH("lg_axpby");
  return LG_OK;
}
