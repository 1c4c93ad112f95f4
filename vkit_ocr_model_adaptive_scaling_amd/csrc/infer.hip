// Device-side post-processing of the adaptive-scaling inference path (inferencing/adaptive_scaling.py:129-188,318-396):
// the reference brings the raw maps to the host and applies sigmoid / threshold / softmax / padding masks with torch-CPU
// and numpy; here each pass is ONE elementwise kernel on the fp32 NCHW head outputs, so only final-size results cross
// PCIe (1 byte / pixel for the mask).
#include "vkas_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// mask = sigmoid(logit) >= thr (uint8); height kept where >= height_min; both forced to 0 on rows >= valid_h[b] /
// columns >= valid_w[b] ("force padding to be negative", :157-168) - valid_* in feature pixels, per image.
__global__ __launch_bounds__(256) void rough_post_kernel(const float* __restrict__ mask_logit,
                                                         const float* __restrict__ height, int B, int H, int W,
                                                         const int* __restrict__ valid_h, const int* __restrict__ valid_w,
                                                         float mask_thr, float height_min,
                                                         unsigned char* __restrict__ out_mask,
                                                         float* __restrict__ out_height) {
  const long n = (long)B * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long r = i / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const bool inside = y < (valid_h ? valid_h[b] : H) && x < (valid_w ? valid_w[b] : W);
    const float h = height[i];
    out_mask[i] = (inside && sigmoidf_(mask_logit[i]) >= mask_thr) ? 1 : 0;
    out_height[i] = (inside && h >= height_min) ? h : 0.f;
  }
}

// prob = sigmoid (0 in the padding); offset (B,2,H,W) -> (B,H,W,2); angle (B,4,H,W) -> softmax over the 4 -> (B,H,W,4);
// distance (B,4,H,W) -> (B,H,W,4)   (:343-396)
__global__ __launch_bounds__(256) void precise_post_kernel(const float* __restrict__ prob_logit,
                                                           const float* __restrict__ offset,
                                                           const float* __restrict__ angle, const float* __restrict__ dist,
                                                           int B, int H, int W, const int* __restrict__ valid_h,
                                                           const int* __restrict__ valid_w, float* __restrict__ out_prob,
                                                           float* __restrict__ out_offset, float* __restrict__ out_angle,
                                                           float* __restrict__ out_dist) {
  const long hw = (long)H * W, n = (long)B * hw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long p = i % hw;
    const int b = (int)(i / hw);
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const bool inside = y < (valid_h ? valid_h[b] : H) && x < (valid_w ? valid_w[b] : W);
    out_prob[i] = inside ? sigmoidf_(prob_logit[i]) : 0.f;
    const float* o = offset + (long)b * 2 * hw + p;
    *reinterpret_cast<float2*>(out_offset + i * 2) = make_float2(o[0], o[hw]);
    const float* a = angle + (long)b * 4 * hw + p;
    const float a0 = a[0], a1 = a[hw], a2 = a[2 * hw], a3 = a[3 * hw];
    const float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    const float e0 = __expf(a0 - mx), e1 = __expf(a1 - mx), e2 = __expf(a2 - mx), e3 = __expf(a3 - mx);
    const float inv = 1.f / (e0 + e1 + e2 + e3);
    *reinterpret_cast<float4*>(out_angle + i * 4) = make_float4(e0 * inv, e1 * inv, e2 * inv, e3 * inv);
    const float* d = dist + (long)b * 4 * hw + p;
    *reinterpret_cast<float4*>(out_dist + i * 4) = make_float4(d[0], d[hw], d[2 * hw], d[3 * hw]);
  }
}

static unsigned post_grid(long n) {
  long b = vkas_cdiv(n > 0 ? n : 1, 256 * 4);
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int vkas_rough_postprocess(const float* mask_logit, const float* height, int B, int H, int W, const int* valid_h,
                                      const int* valid_w, float mask_thr, float height_min, unsigned char* out_mask,
                                      float* out_height, void* stream) {
  VKAS_CHECK(mask_logit && height && out_mask && out_height, "vkas_rough_postprocess: null pointer");
  VKAS_CHECK(B >= 0 && H > 0 && W > 0, "vkas_rough_postprocess: bad dims");
  if (B == 0) return VKAS_OK;
  rough_post_kernel<<<post_grid((long)B * H * W), 256, 0, vkas_stream(stream)>>>(mask_logit, height, B, H, W, valid_h, valid_w,
                                                                               mask_thr, height_min, out_mask, out_height);
  VKAS_LAUNCH_CHECK("rough_postprocess");
  return VKAS_OK;
}

extern "C" int vkas_precise_postprocess(const float* prob_logit, const float* offset, const float* angle, const float* dist,
                                        int B, int H, int W, const int* valid_h, const int* valid_w, float* out_prob,
                                        float* out_offset, float* out_angle, float* out_dist, void* stream) {
  VKAS_CHECK(prob_logit && offset && angle && dist && out_prob && out_offset && out_angle && out_dist,
             "vkas_precise_postprocess: null pointer");
  VKAS_CHECK(B >= 0 && H > 0 && W > 0, "vkas_precise_postprocess: bad dims");
  VKAS_CHECK(vkas_aligned16(out_angle) && vkas_aligned16(out_dist) && (((uintptr_t)out_offset) & 7u) == 0,
             "vkas_precise_postprocess: outputs must be 16-byte aligned");
  if (B == 0) return VKAS_OK;
  precise_post_kernel<<<post_grid((long)B * H * W), 256, 0, vkas_stream(stream)>>>(prob_logit, offset, angle, dist, B, H, W,
                                                                                 valid_h, valid_w, out_prob, out_offset,
                                                                                 out_angle, out_dist);
  VKAS_LAUNCH_CHECK("precise_postprocess");
  return VKAS_OK;
}
