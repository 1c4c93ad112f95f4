// Optimizer step on the flat fp32 parameter / gradient buffers (train.py:468-478): global-norm clipping
// (torch.nn.utils.clip_grad_norm_, max_norm 2.5) fused into a decoupled-weight-decay Adam update (AdamW).
// The flat gradient buffer is the same memory RCCL all-reduces, so the whole step is two kernels.
#include "vkas_common.h"

namespace {

__global__ __launch_bounds__(256) void l2norm_sq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
  double acc = 0.0;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = g4[i];
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc += (double)v * v;
  }
  __shared__ double red[4];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const double* __restrict__ sumsq, float max_norm, float grad_scale,
                                                    float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2) {
  // clip coefficient of torch.nn.utils.clip_grad_norm_: min(1, max_norm / (total_norm + 1e-6))
  float coef = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float total = (float)sqrt(*sumsq) * grad_scale;
    const float c = max_norm / (total + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i];
    pi *= 1.f - lr * wd;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = pi - (lr / bc1) * mi / denom;
  }
}

}  // namespace

extern "C" int vkas_l2norm_sq(const float* g, long n, double* sumsq, void* stream) {
  VKAS_CHECK(g && sumsq && n >= 0 && vkas_aligned16(g), "vkas_l2norm_sq: bad arguments");
  hipStream_t st = vkas_stream(stream);
  (void)hipMemsetAsync(sumsq, 0, sizeof(double), st);
  if (n == 0) return VKAS_OK;
  long grid = vkas_cdiv(n / 4 + 1, 256);
  if (grid > 1024) grid = 1024;
  l2norm_sq_kernel<<<(unsigned)grid, 256, 0, st>>>(g, n, sumsq);
  VKAS_LAUNCH_CHECK("l2norm_sq");
  return VKAS_OK;
}

extern "C" int vkas_adamw_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float max_norm,
                               float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int step, void* stream) {
  VKAS_CHECK(p && g && m && v && n >= 0 && step >= 1, "vkas_adamw_step: bad arguments");
  if (n == 0) return VKAS_OK;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  long grid = vkas_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  adamw_kernel<<<(unsigned)grid, 256, 0, vkas_stream(stream)>>>(p, g, m, v, n, sumsq, max_norm, grad_scale, lr, beta1,
                                                               beta2, eps, weight_decay, bc1, bc2);
  VKAS_LAUNCH_CHECK("adamw_step");
  return VKAS_OK;
}
