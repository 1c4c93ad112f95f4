// Fused dense losses of the adaptive-scaling model (loss_function/adaptive_scaling.py), fp32 maps, fp64 sums.
// Forward: one pass over the cropped maps (+ one over the label points) accumulating the global sums each
// term needs, then a one-thread finalize that forms the scalar.  Backward: one elementwise pass that uses the
// saved sums (the dice / masked-mean denominators), writing complete gradient maps (zeros outside the crop).
// Each term cites the reference line it restates.
#include "vkas_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sl1(float d, float beta) {  // F.smooth_l1_loss element (l1.py:36)
  const float a = fabsf(d);
  return a < beta ? 0.5f * a * a / beta : a - 0.5f * beta;
}
__device__ __forceinline__ float dsl1(float d, float beta) {
  return fabsf(d) < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
}

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

// ------------------------------------------------------------------------------------------------ rough
// sums: [0] focal  [1] sum p*g  [2] sum p  [3] sum g  [4] sum smoothl1*mask  [5] sum mask
__global__ __launch_bounds__(256) void rough_fwd_kernel(const float* __restrict__ mf, const float* __restrict__ hf,
                                                        const float* __restrict__ gm, const float* __restrict__ gs,
                                                        int B, int H, int W, int up, int left, int CH, int CW,
                                                        vkas_rough_loss_cfg cfg, double* __restrict__ sums) {
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const long n = (long)B * CH * CW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int cx = (int)(i % CW);
    const long r = i / CW;
    const int cy = (int)(r % CH);
    const int b = (int)(r / CH);
    const long fi = ((long)b * H + up + cy) * W + left + cx;
    const float x = mf[fi], h = hf[fi], t = gm[i], s = gs[i];
    const float p = sigmoidf_(x);
    // focal (focal_with_logits.py:36-42 -> torchvision sigmoid_focal_loss)
    const float ce = softplusf_(x) - x * t;
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float at = cfg.focal_alpha * t + (1.f - cfg.focal_alpha) * (1.f - t);
    acc[0] += (double)(at * ce * powf(1.f - pt, cfg.focal_gamma));
    // dice (dice.py:32-34)
    acc[1] += (double)(p * t);
    acc[2] += (double)p;
    acc[3] += (double)t;
    // log-space smooth L1 on the height (adaptive_scaling.py:110-128)
    const float m = (h > cfg.height_min && s > cfg.score_min && t != 0.f) ? 1.f : 0.f;
    const float d = logf(fmaxf(h, cfg.height_min)) - logf(fmaxf(s, cfg.score_min));
    acc[4] += (double)(sl1(d, 1.f) * m);
    acc[5] += (double)m;
  }
  block_accumulate<6>(acc, sums);
}

__global__ void rough_finalize_kernel(const double* __restrict__ sums, long n, vkas_rough_loss_cfg cfg,
                                      float* __restrict__ loss) {
  double l = 0.0;
  if (cfg.focal_factor > 0.f) l += (double)cfg.focal_factor * sums[0] / (double)n;
  if (cfg.dice_factor > 0.f) l += (double)cfg.dice_factor * (1.0 - 2.0 * sums[1] / (sums[2] + sums[3] + 1e-6));
  if (cfg.l1_factor > 0.f) l += (double)cfg.l1_factor * sums[4] / (sums[5] + 1e-6);
  *loss = (float)(l * (double)cfg.out_scale);
}

__global__ __launch_bounds__(256) void rough_bwd_kernel(const float* __restrict__ mf, const float* __restrict__ hf,
                                                        const float* __restrict__ gm, const float* __restrict__ gs,
                                                        int B, int H, int W, int up, int left, int CH, int CW,
                                                        vkas_rough_loss_cfg cfg, const double* __restrict__ sums,
                                                        const float* __restrict__ dloss, float* __restrict__ dmf,
                                                        float* __restrict__ dhf) {
  const long n = (long)B * H * W;
  const float go = dloss[0] * cfg.out_scale;
  const double U = sums[2] + sums[3] + 1e-6;
  const float dice_a = (float)(-2.0 / U), dice_b = (float)(2.0 * sums[1] / (U * U));  // dD/dp = a*g + b
  const float inv_n = 1.f / (float)((long)B * CH * CW);
  const float inv_m = (float)(1.0 / (sums[5] + 1e-6));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x_ = (int)(i % W);
    const long r = i / W;
    const int y_ = (int)(r % H);
    const int b = (int)(r / H);
    const int cy = y_ - up, cx = x_ - left;
    float gmf = 0.f, ghf = 0.f;
    if ((unsigned)cy < (unsigned)CH && (unsigned)cx < (unsigned)CW) {
      const long ci = ((long)b * CH + cy) * CW + cx;
      const float x = mf[i], h = hf[i], t = gm[ci], s = gs[ci];
      const float p = sigmoidf_(x);
      const float dp = p * (1.f - p);
      if (cfg.focal_factor > 0.f) {
        const float ce = softplusf_(x) - x * t;
        const float pt = p * t + (1.f - p) * (1.f - t);
        const float at = cfg.focal_alpha * t + (1.f - cfg.focal_alpha) * (1.f - t);
        const float om = 1.f - pt;
        const float df = at * ((p - t) * powf(om, cfg.focal_gamma) -
                               ce * cfg.focal_gamma * powf(om, cfg.focal_gamma - 1.f) * dp * (2.f * t - 1.f));
        gmf += cfg.focal_factor * df * inv_n;
      }
      if (cfg.dice_factor > 0.f) gmf += cfg.dice_factor * (dice_a * t + dice_b) * dp;
      if (cfg.l1_factor > 0.f && h > cfg.height_min && s > cfg.score_min && t != 0.f) {
        const float d = logf(h) - logf(fmaxf(s, cfg.score_min));
        ghf = cfg.l1_factor * dsl1(d, 1.f) / h * inv_m;
      }
    }
    dmf[i] = gmf * go;
    dhf[i] = ghf * go;
  }
}

// ---------------------------------------------------------------------------------------------- precise
// sums: [0] sum (p-s)^2 m  [1] sum m  [2] sum (p-s)^2 (1-m)  [3] sum (1-m)
//       [4] offset smooth-L1  [5] distance regulariser  [6] soft-target CE  [7] corner-distance smooth-L1
__global__ __launch_bounds__(256) void precise_dense_fwd_kernel(const float* __restrict__ prob,
                                                                const float* __restrict__ gs,
                                                                const float* __restrict__ gm, int B, int H, int W, int up,
                                                                int left, int CH, int CW, double* __restrict__ sums) {
  double acc[4] = {0, 0, 0, 0};
  const long n = (long)B * CH * CW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int cx = (int)(i % CW);
    const long r = i / CW;
    const int cy = (int)(r % CH);
    const int b = (int)(r / CH);
    const float p = sigmoidf_(prob[((long)b * H + up + cy) * W + left + cx]);
    const float e = (p - gs[i]) * (p - gs[i]);  // l2.py:32
    const float m = gm[i];
    acc[0] += (double)(e * m);
    acc[1] += (double)m;
    acc[2] += (double)(e * (1.f - m));
    acc[3] += (double)(1.f - m);
  }
  block_accumulate<4>(acc, sums);
}

// label coordinates come from the data pipeline; clamp so that a corrupt index can never fault the GPU
__device__ __forceinline__ long clamp_idx(long v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

struct PointVals {
  float off[2], ang[4], dst[4];
};
__device__ __forceinline__ PointVals gather_point(const float* offset, const float* angle, const float* dist, int b,
                                                  long y, long x, int H, int W) {
  // get_label_point_feature (adaptive_scaling.py:167-179): full-size maps, full-map coordinates
  PointVals v;
  const long hw = (long)H * W, pix = clamp_idx(y, H) * W + clamp_idx(x, W);
#pragma unroll
  for (int c = 0; c < 2; ++c) v.off[c] = offset[((long)b * 2 + c) * hw + pix];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v.ang[c] = angle[((long)b * 4 + c) * hw + pix];
    v.dst[c] = dist[((long)b * 4 + c) * hw + pix];
  }
  return v;
}

__global__ __launch_bounds__(256) void precise_points_fwd_kernel(const float* __restrict__ offset,
                                                                 const float* __restrict__ angle,
                                                                 const float* __restrict__ dist,
                                                                 const int64_t* __restrict__ py,
                                                                 const int64_t* __restrict__ px,
                                                                 const float* __restrict__ go, const float* __restrict__ ga,
                                                                 const float* __restrict__ gd, int B, int H, int W, int P,
                                                                 float beta, double* __restrict__ sums) {
  double acc[4] = {0, 0, 0, 0};
  const long n = (long)B * P;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / P);
    const PointVals v = gather_point(offset, angle, dist, b, py[i], px[i], H, W);
    acc[0] += (double)(sl1(v.off[0] - go[i * 2], beta) + sl1(v.off[1] - go[i * 2 + 1], beta));  // :309-313
    const float nrm = sqrtf(v.off[0] * v.off[0] + v.off[1] * v.off[1]);
    acc[1] += (double)sl1(nrm - v.dst[0], beta);  // :315-326
    float mx = fmaxf(fmaxf(v.ang[0], v.ang[1]), fmaxf(v.ang[2], v.ang[3]));
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) se += __expf(v.ang[c] - mx);
    const float lse = mx + logf(se);
    float ce = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) ce -= ga[i * 4 + c] * (v.ang[c] - lse);  // :328-333, soft targets
    acc[2] += (double)ce;
    float dd = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dd += sl1(v.dst[c + 1] - gd[i * 3 + c], beta);  // :335-339
    acc[3] += (double)dd;
  }
  block_accumulate<4>(acc, sums + 4);
}

__global__ void precise_finalize_kernel(const double* __restrict__ s, long npts, vkas_precise_loss_cfg cfg,
                                        float* __restrict__ loss) {
  double l = 0.0;
  if (cfg.pos_l2 > 0.f) l += (double)cfg.pos_l2 * s[0] / (s[1] + 1e-6);
  if (cfg.neg_l2 > 0.f) l += (double)cfg.neg_l2 * s[2] / (s[3] + 1e-6);
  if (cfg.offset_l1 > 0.f) l += (double)cfg.offset_l1 * s[4] / (double)(npts * 2);
  if (cfg.reg_l1 > 0.f) l += (double)cfg.reg_l1 * s[5] / (double)npts;
  if (cfg.angle_ce > 0.f) l += (double)cfg.angle_ce * s[6] / (double)npts;
  if (cfg.dist_l1 > 0.f) l += (double)cfg.dist_l1 * s[7] / (double)(npts * 3);
  *loss = (float)(l * (double)cfg.loss_factor * (double)cfg.out_scale);  // :344
}

__global__ __launch_bounds__(256) void precise_dense_bwd_kernel(const float* __restrict__ prob,
                                                                const float* __restrict__ gs,
                                                                const float* __restrict__ gm, int B, int H, int W, int up,
                                                                int left, int CH, int CW, vkas_precise_loss_cfg cfg,
                                                                const double* __restrict__ sums,
                                                                const float* __restrict__ dloss,
                                                                float* __restrict__ dprob) {
  const long n = (long)B * H * W;
  const float go = dloss[0] * cfg.out_scale * cfg.loss_factor;
  const float kp = cfg.pos_l2 > 0.f ? (float)((double)cfg.pos_l2 / (sums[1] + 1e-6)) : 0.f;
  const float kn = cfg.neg_l2 > 0.f ? (float)((double)cfg.neg_l2 / (sums[3] + 1e-6)) : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x_ = (int)(i % W);
    const long r = i / W;
    const int y_ = (int)(r % H);
    const int b = (int)(r / H);
    const int cy = y_ - up, cx = x_ - left;
    float g = 0.f;
    if ((unsigned)cy < (unsigned)CH && (unsigned)cx < (unsigned)CW) {
      const long ci = ((long)b * CH + cy) * CW + cx;
      const float p = sigmoidf_(prob[i]);
      const float m = gm[ci];
      g = 2.f * (p - gs[ci]) * (kp * m + kn * (1.f - m)) * p * (1.f - p);
    }
    dprob[i] = g * go;
  }
}

__global__ __launch_bounds__(256) void precise_points_bwd_kernel(const float* __restrict__ offset,
                                                                 const float* __restrict__ angle,
                                                                 const float* __restrict__ dist,
                                                                 const int64_t* __restrict__ py,
                                                                 const int64_t* __restrict__ px,
                                                                 const float* __restrict__ gof, const float* __restrict__ ga,
                                                                 const float* __restrict__ gd, int B, int H, int W, int P,
                                                                 vkas_precise_loss_cfg cfg, const float* __restrict__ dloss,
                                                                 float* __restrict__ doff, float* __restrict__ dang,
                                                                 float* __restrict__ ddst) {
  const long n = (long)B * P;
  const float go = dloss[0] * cfg.out_scale * cfg.loss_factor;
  const float beta = cfg.smooth_beta;
  const float k_off = cfg.offset_l1 / (float)(n * 2), k_reg = cfg.reg_l1 / (float)n, k_ce = cfg.angle_ce / (float)n,
              k_d = cfg.dist_l1 / (float)(n * 3);
  const long hw = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / P);
    const long pix = clamp_idx(py[i], H) * W + clamp_idx(px[i], W);
    const PointVals v = gather_point(offset, angle, dist, b, py[i], px[i], H, W);
    const float nrm = sqrtf(v.off[0] * v.off[0] + v.off[1] * v.off[1]);
    const float dreg = dsl1(nrm - v.dst[0], beta) * k_reg;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float g = dsl1(v.off[c] - gof[i * 2 + c], beta) * k_off;
      if (nrm > 0.f) g += dreg * v.off[c] / nrm;
      atomicAdd(&doff[((long)b * 2 + c) * hw + pix], g * go);
    }
    float mx = fmaxf(fmaxf(v.ang[0], v.ang[1]), fmaxf(v.ang[2], v.ang[3]));
    float e[4], se = 0.f, st = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      e[c] = __expf(v.ang[c] - mx);
      se += e[c];
      st += ga[i * 4 + c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      atomicAdd(&dang[((long)b * 4 + c) * hw + pix], (st * e[c] / se - ga[i * 4 + c]) * k_ce * go);
    atomicAdd(&ddst[((long)b * 4 + 0) * hw + pix], -dreg * go);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      atomicAdd(&ddst[((long)b * 4 + c + 1) * hw + pix], dsl1(v.dst[c + 1] - gd[i * 3 + c], beta) * k_d * go);
  }
}

static inline unsigned grid_for(long n) {
  long g = vkas_cdiv(n, 256);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

static int crop_check(const char* who, int B, int H, int W, int up, int left, int CH, int CW) {
  VKAS_CHECK(B > 0 && H > 0 && W > 0 && CH > 0 && CW > 0 && up >= 0 && left >= 0 && up + CH <= H && left + CW <= W,
             "%s: crop (%d,%d)+(%dx%d) outside the %dx%d map", who, up, left, CH, CW, H, W);
  return VKAS_OK;
}

}  // namespace

extern "C" int vkas_rough_loss_fwd(const float* mask_feat, const float* height_feat, const float* gt_mask,
                                   const float* gt_score, int B, int H, int W, int up, int left, int CH, int CW,
                                   const vkas_rough_loss_cfg* cfg, double* sums, float* loss, void* stream) {
  VKAS_CHECK(mask_feat && height_feat && gt_mask && gt_score && cfg && sums && loss, "vkas_rough_loss_fwd: null pointer");
  int rc = crop_check("vkas_rough_loss_fwd", B, H, W, up, left, CH, CW);
  if (rc) return rc;
  hipStream_t st = vkas_stream(stream);
  (void)hipMemsetAsync(sums, 0, 8 * sizeof(double), st);
  const long n = (long)B * CH * CW;
  rough_fwd_kernel<<<grid_for(n), 256, 0, st>>>(mask_feat, height_feat, gt_mask, gt_score, B, H, W, up, left, CH, CW, *cfg,
                                                sums);
  rough_finalize_kernel<<<1, 1, 0, st>>>(sums, n, *cfg, loss);
  VKAS_LAUNCH_CHECK("rough_loss_fwd");
  return VKAS_OK;
}

extern "C" int vkas_rough_loss_bwd(const float* mask_feat, const float* height_feat, const float* gt_mask,
                                   const float* gt_score, int B, int H, int W, int up, int left, int CH, int CW,
                                   const vkas_rough_loss_cfg* cfg, const double* sums, const float* dloss,
                                   float* d_mask_feat, float* d_height_feat, void* stream) {
  VKAS_CHECK(mask_feat && height_feat && gt_mask && gt_score && cfg && sums && dloss && d_mask_feat && d_height_feat,
             "vkas_rough_loss_bwd: null pointer");
  int rc = crop_check("vkas_rough_loss_bwd", B, H, W, up, left, CH, CW);
  if (rc) return rc;
  rough_bwd_kernel<<<grid_for((long)B * H * W), 256, 0, vkas_stream(stream)>>>(
      mask_feat, height_feat, gt_mask, gt_score, B, H, W, up, left, CH, CW, *cfg, sums, dloss, d_mask_feat, d_height_feat);
  VKAS_LAUNCH_CHECK("rough_loss_bwd");
  return VKAS_OK;
}

// smallest distance of any label point to the border of the (H, W) map, negative when a point lies outside: the range check
// the reference's advanced indexing performs on the host (loss_function/adaptive_scaling.py:235-262), as one workgroup
__global__ __launch_bounds__(256) void points_margin_kernel(const int64_t* __restrict__ py, const int64_t* __restrict__ px,
                                                            long n, int H, int W, int64_t* __restrict__ margin) {
  __shared__ long long part[256];
  long long m = 0x7fffffffffffffffLL;
  for (long i = threadIdx.x; i < n; i += 256) {
    const long long y = py[i], x = px[i];
    const long long a = y < (H - 1) - y ? y : (H - 1) - y, b = x < (W - 1) - x ? x : (W - 1) - x;
    const long long c = a < b ? a : b;
    m = c < m ? c : m;
  }
  part[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && part[threadIdx.x + s] < part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) margin[0] = (int64_t)part[0];
}

extern "C" int vkas_points_margin(const int64_t* py, const int64_t* px, long n, int H, int W, int64_t* margin, void* stream) {
  VKAS_CHECK(py && px && margin && n > 0 && H > 0 && W > 0, "vkas_points_margin: bad arguments");
  points_margin_kernel<<<1, 256, 0, vkas_stream(stream)>>>(py, px, n, H, W, margin);
  VKAS_LAUNCH_CHECK("points_margin");
  return VKAS_OK;
}

extern "C" int vkas_precise_loss_fwd(const float* prob, const float* offset, const float* angle, const float* dist,
                                     const float* gt_score, const float* gt_mask, const int64_t* py, const int64_t* px,
                                     const float* gt_offsets, const float* gt_angles, const float* gt_dists, int B, int H,
                                     int W, int up, int left, int CH, int CW, int P, const vkas_precise_loss_cfg* cfg,
                                     double* sums, float* loss, void* stream) {
  VKAS_CHECK(prob && offset && angle && dist && gt_score && gt_mask && py && px && gt_offsets && gt_angles && gt_dists &&
                 cfg && sums && loss,
             "vkas_precise_loss_fwd: null pointer");
  int rc = crop_check("vkas_precise_loss_fwd", B, H, W, up, left, CH, CW);
  if (rc) return rc;
  VKAS_CHECK(P > 0, "vkas_precise_loss_fwd: P must be positive");
  hipStream_t st = vkas_stream(stream);
  (void)hipMemsetAsync(sums, 0, 8 * sizeof(double), st);
  precise_dense_fwd_kernel<<<grid_for((long)B * CH * CW), 256, 0, st>>>(prob, gt_score, gt_mask, B, H, W, up, left, CH, CW,
                                                                       sums);
  precise_points_fwd_kernel<<<grid_for((long)B * P), 256, 0, st>>>(offset, angle, dist, py, px, gt_offsets, gt_angles,
                                                                  gt_dists, B, H, W, P, cfg->smooth_beta, sums);
  precise_finalize_kernel<<<1, 1, 0, st>>>(sums, (long)B * P, *cfg, loss);
  VKAS_LAUNCH_CHECK("precise_loss_fwd");
  return VKAS_OK;
}

extern "C" int vkas_precise_loss_bwd(const float* prob, const float* offset, const float* angle, const float* dist,
                                     const float* gt_score, const float* gt_mask, const int64_t* py, const int64_t* px,
                                     const float* gt_offsets, const float* gt_angles, const float* gt_dists, int B, int H,
                                     int W, int up, int left, int CH, int CW, int P, const vkas_precise_loss_cfg* cfg,
                                     const double* sums, const float* dloss, float* d_prob, float* d_offset,
                                     float* d_angle, float* d_dist, void* stream) {
  VKAS_CHECK(prob && offset && angle && dist && gt_score && gt_mask && py && px && gt_offsets && gt_angles && gt_dists &&
                 cfg && sums && dloss && d_prob && d_offset && d_angle && d_dist,
             "vkas_precise_loss_bwd: null pointer");
  int rc = crop_check("vkas_precise_loss_bwd", B, H, W, up, left, CH, CW);
  if (rc) return rc;
  VKAS_CHECK(P > 0, "vkas_precise_loss_bwd: P must be positive");
  hipStream_t st = vkas_stream(stream);
  const size_t hw = (size_t)H * W * sizeof(float);
  (void)hipMemsetAsync(d_offset, 0, (size_t)B * 2 * hw, st);
  (void)hipMemsetAsync(d_angle, 0, (size_t)B * 4 * hw, st);
  (void)hipMemsetAsync(d_dist, 0, (size_t)B * 4 * hw, st);
  precise_dense_bwd_kernel<<<grid_for((long)B * H * W), 256, 0, st>>>(prob, gt_score, gt_mask, B, H, W, up, left, CH, CW,
                                                                     *cfg, sums, dloss, d_prob);
  precise_points_bwd_kernel<<<grid_for((long)B * P), 256, 0, st>>>(offset, angle, dist, py, px, gt_offsets, gt_angles,
                                                                  gt_dists, B, H, W, P, *cfg, dloss, d_offset, d_angle,
                                                                  d_dist);
  VKAS_LAUNCH_CHECK("precise_loss_bwd");
  return VKAS_OK;
}
