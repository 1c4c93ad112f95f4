// Primitive loss callables of vkit_open_model/loss_function (the classes the composite adaptive-scaling losses are
// built from and that loss_function/__init__.py:12-18 exports): focal-with-logits, dice, (smooth) L1, L2 and
// soft-target cross entropy, each as one reduction pass (fp64 sums) + a one-thread finalize, and one elementwise
// backward pass that uses the saved sums.  fp32 inputs; gradients are produced for `pred` only (targets are data).
#include "vkas_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }

template <int N>
__device__ __forceinline__ void block_accumulate(double* acc, double* __restrict__ sums) {
  __shared__ double red[4][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double v = wave_sum_d(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double v = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
    atomicAdd(&sums[threadIdx.x], v);
  }
}

// per-element value e(pred, gt) and de/dpred of the mean-type terms
template <int KIND>
__device__ __forceinline__ float elem_value(float x, float t, float p0, float p1) {
  if constexpr (KIND == VKAS_LOSS_FOCAL) {  // focal_with_logits.py:36-42 (torchvision sigmoid_focal_loss closed form)
    const float p = sigmoidf_(x);
    const float ce = softplusf_(x) - x * t;
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float at = p0 >= 0.f ? p0 * t + (1.f - p0) * (1.f - t) : 1.f;
    return at * ce * powf(1.f - pt, p1);
  } else if constexpr (KIND == VKAS_LOSS_L1) {  // l1.py:38-39 (F.l1_loss)
    return fabsf(x - t);
  } else if constexpr (KIND == VKAS_LOSS_SMOOTH_L1) {  // l1.py:41 (F.smooth_l1_loss, beta = p0)
    const float a = fabsf(x - t);
    return a < p0 ? 0.5f * a * a / p0 : a - 0.5f * p0;
  } else {  // l2.py:29-32 (F.mse_loss)
    return (x - t) * (x - t);
  }
}

template <int KIND>
__device__ __forceinline__ float elem_grad(float x, float t, float p0, float p1) {
  if constexpr (KIND == VKAS_LOSS_FOCAL) {
    const float p = sigmoidf_(x);
    const float dp = p * (1.f - p);
    const float ce = softplusf_(x) - x * t;
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float at = p0 >= 0.f ? p0 * t + (1.f - p0) * (1.f - t) : 1.f;
    const float om = 1.f - pt;
    // d/dx [ce * om^g] = (p - t) om^g - ce g om^(g-1) dpt/dx,  dpt/dx = dp (2t - 1)
    const float tail = p1 != 0.f ? ce * p1 * powf(om, p1 - 1.f) * dp * (2.f * t - 1.f) : 0.f;
    return at * ((p - t) * powf(om, p1) - tail);
  } else if constexpr (KIND == VKAS_LOSS_L1) {
    const float d = x - t;
    return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  } else if constexpr (KIND == VKAS_LOSS_SMOOTH_L1) {
    const float d = x - t;
    return fabsf(d) < p0 ? d / p0 : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  } else {
    return 2.f * (x - t);
  }
}

// sums: mean-type kinds [0] sum e*m  [1] sum m ; dice [0] sum p g m^2  [1] sum p m  [2] sum g m
template <int KIND>
__global__ __launch_bounds__(256) void prim_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                       const float* __restrict__ mask, long n, float p0, float p1,
                                                       double* __restrict__ sums) {
  double acc[3] = {0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = pred[i], t = gt[i], m = mask ? mask[i] : 1.f;
    if constexpr (KIND == VKAS_LOSS_DICE) {  // dice.py:28-34
      acc[0] += (double)(x * m * t * m);
      acc[1] += (double)(x * m);
      acc[2] += (double)(t * m);
    } else {
      acc[0] += (double)(elem_value<KIND>(x, t, p0, p1) * m);
      acc[1] += (double)m;
    }
  }
  block_accumulate<3>(acc, sums);
}

__global__ void prim_finalize_kernel(int kind, int has_mask, long n, float eps, const double* __restrict__ sums,
                                     float* __restrict__ loss) {
  double l;
  if (kind == VKAS_LOSS_DICE) l = 1.0 - 2.0 * sums[0] / (sums[1] + sums[2] + (double)eps);
  else if (has_mask) l = sums[0] / (sums[1] + (double)eps);
  else l = sums[0] / (double)(n > 0 ? n : 1);
  *loss = (float)l;
}

template <int KIND>
__global__ __launch_bounds__(256) void prim_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                       const float* __restrict__ mask, long n, float p0, float p1,
                                                       float eps, const double* __restrict__ sums,
                                                       const float* __restrict__ dloss, float* __restrict__ dpred) {
  const float go = dloss[0];
  float a = 0.f, b = 0.f, inv = 0.f;
  if constexpr (KIND == VKAS_LOSS_DICE) {
    const double U = sums[1] + sums[2] + (double)eps;
    a = (float)(-2.0 / U);               // d/d(p m) = a * (g m) + b
    b = (float)(2.0 * sums[0] / (U * U));
  } else {
    inv = mask ? (float)(1.0 / (sums[1] + (double)eps)) : 1.f / (float)(n > 0 ? n : 1);
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = pred[i], t = gt[i], m = mask ? mask[i] : 1.f;
    float g;
    if constexpr (KIND == VKAS_LOSS_DICE) g = m * (a * t * m + b);
    else g = elem_grad<KIND>(x, t, p0, p1) * m * inv;
    dpred[i] = g * go;
  }
}

// F.cross_entropy(logits (R, C), target): soft targets (R, C) fp32 or class indices (R,) int64; mean over rows
// (cross_entropy_with_logits.py:16-19).  One lane per row; C <= 64.  sums[0] = sum of row losses.
template <bool HARD>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                     long rows, int C, double* __restrict__ sums) {
  double acc[1] = {0};
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    const float* x = logits + r * C;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(x[c] - mx);
    const float lse = mx + logf(se);
    float l = 0.f;
    if constexpr (HARD) {
      const long k = reinterpret_cast<const long*>(target)[r];
      l = lse - x[k];
    } else {
      const float* t = reinterpret_cast<const float*>(target) + r * C;
      for (int c = 0; c < C; ++c) l += t[c] * (lse - x[c]);
    }
    acc[0] += (double)l;
  }
  block_accumulate<1>(acc, sums);
}

__global__ void ce_finalize_kernel(long rows, const double* __restrict__ sums, float* __restrict__ loss) {
  *loss = (float)(sums[0] / (double)(rows > 0 ? rows : 1));
}

template <bool HARD>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                     long rows, int C, const float* __restrict__ dloss,
                                                     float* __restrict__ dlogits) {
  const float go = dloss[0] / (float)(rows > 0 ? rows : 1);
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    const float* x = logits + r * C;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(x[c] - mx);
    const float inv = 1.f / se;
    float ts = 1.f;
    long k = -1;
    if constexpr (HARD) {
      k = reinterpret_cast<const long*>(target)[r];
    } else {
      const float* t = reinterpret_cast<const float*>(target) + r * C;
      ts = 0.f;
      for (int c = 0; c < C; ++c) ts += t[c];
    }
    for (int c = 0; c < C; ++c) {
      const float sm = __expf(x[c] - mx) * inv;
      const float tc = HARD ? (c == k ? 1.f : 0.f) : reinterpret_cast<const float*>(target)[r * C + c];
      dlogits[r * C + c] = (sm * ts - tc) * go;  // d/dx_c sum_j t_j (lse - x_j)
    }
  }
}

static unsigned prim_grid(long n) {
  long b = vkas_cdiv(n > 0 ? n : 1, 256 * 4);
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

#define VKAS_PRIM_KIND(kind, ...)                                         \
  switch (kind) {                                                         \
    case VKAS_LOSS_FOCAL: { constexpr int KIND = VKAS_LOSS_FOCAL; __VA_ARGS__ } break;         \
    case VKAS_LOSS_DICE: { constexpr int KIND = VKAS_LOSS_DICE; __VA_ARGS__ } break;           \
    case VKAS_LOSS_L1: { constexpr int KIND = VKAS_LOSS_L1; __VA_ARGS__ } break;               \
    case VKAS_LOSS_SMOOTH_L1: { constexpr int KIND = VKAS_LOSS_SMOOTH_L1; __VA_ARGS__ } break; \
    case VKAS_LOSS_L2: { constexpr int KIND = VKAS_LOSS_L2; __VA_ARGS__ } break;               \
    default:                                                              \
      vkas_set_error("vkas_elementwise_loss: unknown kind %d", (int)(kind)); \
      return VKAS_E_ARG;                                                  \
  }

extern "C" int vkas_elementwise_loss_fwd(int kind, const float* pred, const float* gt, const float* mask, long n, float p0,
                                         float p1, float eps, double* sums, float* loss, void* stream) {
  VKAS_CHECK(pred && gt && sums && loss, "vkas_elementwise_loss_fwd: null pointer");
  VKAS_CHECK(n >= 0, "vkas_elementwise_loss_fwd: negative size");
  VKAS_CHECK(kind != VKAS_LOSS_SMOOTH_L1 || p0 > 0.f, "vkas_elementwise_loss_fwd: smooth-L1 beta must be positive");
  hipStream_t st = vkas_stream(stream);
  (void)hipMemsetAsync(sums, 0, 4 * sizeof(double), st);
  if (n > 0) {
    VKAS_PRIM_KIND(kind, { prim_fwd_kernel<KIND><<<prim_grid(n), 256, 0, st>>>(pred, gt, mask, n, p0, p1, sums); })
  }
  prim_finalize_kernel<<<1, 1, 0, st>>>(kind, mask != nullptr, n, eps, sums, loss);
  VKAS_LAUNCH_CHECK("elementwise_loss_fwd");
  return VKAS_OK;
}

extern "C" int vkas_elementwise_loss_bwd(int kind, const float* pred, const float* gt, const float* mask, long n, float p0,
                                         float p1, float eps, const double* sums, const float* dloss, float* dpred,
                                         void* stream) {
  VKAS_CHECK(pred && gt && sums && dloss && dpred, "vkas_elementwise_loss_bwd: null pointer");
  VKAS_CHECK(n >= 0, "vkas_elementwise_loss_bwd: negative size");
  if (n == 0) return VKAS_OK;
  hipStream_t st = vkas_stream(stream);
  VKAS_PRIM_KIND(kind, { prim_bwd_kernel<KIND><<<prim_grid(n), 256, 0, st>>>(pred, gt, mask, n, p0, p1, eps, sums, dloss, dpred); })
  VKAS_LAUNCH_CHECK("elementwise_loss_bwd");
  return VKAS_OK;
}

extern "C" int vkas_cross_entropy_fwd(const float* logits, const void* target, int hard, long rows, int classes,
                                      double* sums, float* loss, void* stream) {
  VKAS_CHECK(logits && target && sums && loss, "vkas_cross_entropy_fwd: null pointer");
  VKAS_CHECK(rows >= 0 && classes >= 1 && classes <= 64, "vkas_cross_entropy_fwd: bad shape rows=%ld classes=%d", rows, classes);
  hipStream_t st = vkas_stream(stream);
  (void)hipMemsetAsync(sums, 0, 4 * sizeof(double), st);
  if (rows > 0) {
    if (hard) ce_fwd_kernel<true><<<prim_grid(rows * 4), 256, 0, st>>>(logits, target, rows, classes, sums);
    else ce_fwd_kernel<false><<<prim_grid(rows * 4), 256, 0, st>>>(logits, target, rows, classes, sums);
  }
  ce_finalize_kernel<<<1, 1, 0, st>>>(rows, sums, loss);
  VKAS_LAUNCH_CHECK("cross_entropy_fwd");
  return VKAS_OK;
}

extern "C" int vkas_cross_entropy_bwd(const float* logits, const void* target, int hard, long rows, int classes,
                                      const float* dloss, float* dlogits, void* stream) {
  VKAS_CHECK(logits && target && dloss && dlogits, "vkas_cross_entropy_bwd: null pointer");
  VKAS_CHECK(rows >= 0 && classes >= 1 && classes <= 64, "vkas_cross_entropy_bwd: bad shape rows=%ld classes=%d", rows, classes);
  if (rows == 0) return VKAS_OK;
  hipStream_t st = vkas_stream(stream);
  if (hard) ce_bwd_kernel<true><<<prim_grid(rows * 4), 256, 0, st>>>(logits, target, rows, classes, dloss, dlogits);
  else ce_bwd_kernel<false><<<prim_grid(rows * 4), 256, 0, st>>>(logits, target, rows, classes, dloss, dlogits);
  VKAS_LAUNCH_CHECK("cross_entropy_bwd");
  return VKAS_OK;
}
