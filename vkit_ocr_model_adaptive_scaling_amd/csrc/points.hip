// Label-point backward of the regression heads.
//
// AdaptiveScalingPreciseLossFunction reads the up-left-offset / corner-angle / corner-distance maps ONLY at the label
// points (loss_function/adaptive_scaling.py:167-179,235-262: get_label_point_feature, P = 200 points per image,
// train.py:58), so the gradient those three heads receive is zero on all but B*P of the B*H*W pixels.  Everything between
// the loss and the head convolution is per pixel (Softplus, NCHW permute, Linear, GELU, LayerNorm), hence d(conv output)
// of those heads is zero off the points as well and their share of the head convolution's backward
//   gw[n][ky][kx][c] = sum_p dz[p][n] * x[p + (ky-1, kx-1)][c]        dx[p + (ky-1, kx-1)][c] += dz[p][n] * w[n][c][ky][kx]
// only has B*P non-zero rows.  These kernels compact those rows so that the existing head-tail and GEMM kernels run on
// (B*P) x ... operands instead of (B*H*W) x ...: exact (the dropped products are products with zero), and ~600 of the 773
// conv output channels of the precise pass leave the dense backward.
//   vkas_points_prepare        pixel index of every label point; duplicate points (their gradients are already summed in
//                              the dense map) are marked so that each pixel is taken once; pixel -> owner-point map
//   vkas_points_gather_rows    z / LayerNorm statistics / d(proj) rows at the points -> compact operands of vkas_head_tail_bwd
//   vkas_points_gather_patches 3x3 input patches at the points -> compact (rows, 9*Cp) operand of the weight-gradient GEMM
//   vkas_points_scatter3x3     compact fp32 (rows, 9*Cp) input-gradient contributions -> added onto dx, every touched pixel summed
//                              in fp32 by exactly one workgroup (no atomics, deterministic)
#include "vkas_common.h"

namespace {

constexpr int PT_EMPTY = 0x7f7f7f7f;  // map value of a pixel without a label point (memset pattern 0x7f)
constexpr int PT_PAD = INT_MIN;       // pix value of the padding rows behind the last point

__device__ __forceinline__ long pt_clamp(long v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }  // as loss.hip clamp_idx

__global__ __launch_bounds__(256) void points_claim_kernel(const long* __restrict__ py, const long* __restrict__ px, int P,
                                                           int H, int W, long n, int* __restrict__ map) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long q = (i / P) * H * W + pt_clamp(py[i], H) * W + pt_clamp(px[i], W);
  atomicMin(&map[q], (int)i);
}

// pix[i] = q for the lowest-numbered point of pixel q ("owner"), -1 - q for the other points of that pixel, PT_PAD for padding
__global__ __launch_bounds__(256) void points_pix_kernel(const long* __restrict__ py, const long* __restrict__ px, int P, int H,
                                                         int W, long n, long Mp, const int* __restrict__ map,
                                                         int* __restrict__ pix) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= Mp) return;
  if (i >= n) {
    pix[i] = PT_PAD;
    return;
  }
  const long q = (i / P) * H * W + pt_clamp(py[i], H) * W + pt_clamp(px[i], W);
  pix[i] = map[q] == (int)i ? (int)q : -1 - (int)q;
}

struct GatherRowsArgs {
  const float* dproj[4];
  int n_heads;
};

// one 64-lane workgroup per compact row
template <typename T>
__global__ __launch_bounds__(64) void points_gather_rows_kernel(const T* __restrict__ z, long ldz, int c0, int Ns,
                                                                const float* __restrict__ stats, GatherRowsArgs a, long M,
                                                                const int* __restrict__ pix, long Mp, T* __restrict__ zs,
                                                                float* __restrict__ stats_s, float* __restrict__ dproj_s) {
  const long i = blockIdx.x;
  const int p = pix[i];
  const bool pad = p == PT_PAD, own = p >= 0;
  const long q = pad ? 0 : (own ? p : -1 - p);
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int v = threadIdx.x; v < (Ns >> 3); v += 64) {
    float r[8];
    if (pad) store8(zs + i * Ns + v * 8, zero8);
    else {
      load8(z + q * ldz + c0 + v * 8, r);
      store8(zs + i * Ns + v * 8, r);
    }
  }
  for (int t = threadIdx.x; t < a.n_heads * 10; t += 64) {
    const int h = t / 10, r = t - h * 10;
    if (r < 2) stats_s[((long)h * Mp + i) * 2 + r] = pad ? 0.f : stats[((long)h * M + q) * 2 + r];
    else dproj_s[((long)h * Mp + i) * 8 + (r - 2)] = own ? a.dproj[h][q * 8 + (r - 2)] : 0.f;  // a duplicate contributes nothing
  }
}

template <typename T>
__global__ __launch_bounds__(64) void points_gather_patches_kernel(const T* __restrict__ x, long ldx, int Cp, int H, int W,
                                                                   const int* __restrict__ pix, T* __restrict__ xs) {
  const long i = blockIdx.x;
  const int p = pix[i];
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int vpt = Cp >> 3;
  const long q = p >= 0 ? p : 0;
  const int xx = (int)(q % W), yy = (int)((q / W) % H);
  for (int v = threadIdx.x; v < 9 * vpt; v += 64) {
    const int t = v / vpt, cv = v - t * vpt;
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool ok = p >= 0 && (unsigned)(yy + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W;  // zero padding of the conv
    float r[8];
    if (ok) {
      load8(x + (q + (long)dy * W + dx) * ldx + cv * 8, r);
      store8(xs + (i * 9 + t) * Cp + cv * 8, r);
    } else {
      store8(xs + (i * 9 + t) * Cp + cv * 8, zero8);
    }
  }
}

// D[i][t][:] (t = ky*3 + kx) belongs to pixel pixel(i) + (ky-1, kx-1).  Several (point, tap) pairs can land on one pixel: the
// first contributor in tap order owns it, sums all of them in fp32 and adds the sum onto dx once.
template <typename T>
__global__ __launch_bounds__(64) void points_scatter3x3_kernel(const float* __restrict__ D, const int* __restrict__ pix,
                                                               const int* __restrict__ map, int H, int W, int Cp,
                                                               T* __restrict__ dx, long lddx) {
  const long i = blockIdx.x / 9;
  const int t = (int)(blockIdx.x - i * 9);
  const int p = pix[i];
  if (p < 0) return;  // duplicates and padding rows carry zeros
  const long q = p;
  const int xx = (int)(q % W), yy = (int)((q / W) % H);
  const long img = q - ((long)yy * W + xx);
  const int ty = yy + t / 3 - 1, tx = xx + t % 3 - 1;
  if ((unsigned)ty >= (unsigned)H || (unsigned)tx >= (unsigned)W) return;
  int src[9];
  bool first = true, mine = false;
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int ry = ty - (u / 3 - 1), rx = tx - (u % 3 - 1);  // the point whose tap u lands on (ty, tx)
    int o = PT_EMPTY;
    if ((unsigned)ry < (unsigned)H && (unsigned)rx < (unsigned)W) o = map[img + (long)ry * W + rx];
    src[u] = o;
    if (o != PT_EMPTY && first) {
      first = false;
      mine = (o == (int)i && u == t);
    }
  }
  if (!mine) return;
  T* out = dx + (img + (long)ty * W + tx) * lddx;
  for (int v = threadIdx.x; v < (Cp >> 3); v += 64) {
    float acc[8];
    load8(out + v * 8, acc);
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      if (src[u] == PT_EMPTY) continue;
      const float4* r = reinterpret_cast<const float4*>(D + ((long)src[u] * 9 + u) * Cp + v * 8);
      const float4 r0 = r[0], r1 = r[1];
      acc[0] += r0.x; acc[1] += r0.y; acc[2] += r0.z; acc[3] += r0.w;
      acc[4] += r1.x; acc[5] += r1.y; acc[6] += r1.z; acc[7] += r1.w;
    }
    store8(out + v * 8, acc);
  }
}

// (rows, 8) fp32 vectors <-> the (M, 8) projected-channel maps of a head, at the label points
__global__ __launch_bounds__(256) void points_scatter_vec8_kernel(const float* __restrict__ src, const int* __restrict__ pix,
                                                                  long Mp, float* __restrict__ dst) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;  // (row, half)
  const long i = t >> 1;
  if (i >= Mp) return;
  const int p = pix[i];
  if (p < 0) return;  // padding rows, and duplicates of a pixel: its owner's row is the one computed from the real patch
  reinterpret_cast<float4*>(dst + (long)p * 8)[t & 1] = reinterpret_cast<const float4*>(src + i * 8)[t & 1];
}

__global__ __launch_bounds__(256) void points_gather_vec8_kernel(const float* __restrict__ src, const int* __restrict__ pix,
                                                                 long Mp, float* __restrict__ dst) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long i = t >> 1;
  if (i >= Mp) return;
  const int p = pix[i];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p >= 0) v = reinterpret_cast<const float4*>(src + (long)p * 8)[t & 1];  // a pixel's gradient is taken once (its owner)
  reinterpret_cast<float4*>(dst + i * 8)[t & 1] = v;
}

}  // namespace

extern "C" int vkas_points_scatter_vec8(const float* src, const int* pix, long Mp, float* dst, void* stream) {
  VKAS_CHECK(src && pix && dst && Mp > 0 && vkas_aligned16(src) && vkas_aligned16(dst), "vkas_points_scatter_vec8: bad arguments");
  points_scatter_vec8_kernel<<<(unsigned)vkas_cdiv(2 * Mp, 256), 256, 0, vkas_stream(stream)>>>(src, pix, Mp, dst);
  VKAS_LAUNCH_CHECK("points_scatter_vec8");
  return VKAS_OK;
}

extern "C" int vkas_points_gather_vec8(const float* src, const int* pix, long Mp, float* dst, void* stream) {
  VKAS_CHECK(src && pix && dst && Mp > 0 && vkas_aligned16(src) && vkas_aligned16(dst), "vkas_points_gather_vec8: bad arguments");
  points_gather_vec8_kernel<<<(unsigned)vkas_cdiv(2 * Mp, 256), 256, 0, vkas_stream(stream)>>>(src, pix, Mp, dst);
  VKAS_LAUNCH_CHECK("points_gather_vec8");
  return VKAS_OK;
}

extern "C" int vkas_points_prepare(const long* py, const long* px, int B, int P, int H, int W, int* map, int* pix, long Mp,
                                   void* stream) {
  VKAS_CHECK(py && px && map && pix, "vkas_points_prepare: null pointer");
  const long n = (long)B * P, M = (long)B * H * W;
  VKAS_CHECK(B > 0 && P > 0 && H > 0 && W > 0 && M < PT_EMPTY && n < PT_EMPTY && Mp >= n,
             "vkas_points_prepare: bad sizes B=%d P=%d H=%d W=%d Mp=%ld", B, P, H, W, Mp);
  hipStream_t st = vkas_stream(stream);
  (void)hipMemsetAsync(map, 0x7f, (size_t)M * sizeof(int), st);
  points_claim_kernel<<<(unsigned)vkas_cdiv(n, 256), 256, 0, st>>>(py, px, P, H, W, n, map);
  points_pix_kernel<<<(unsigned)vkas_cdiv(Mp, 256), 256, 0, st>>>(py, px, P, H, W, n, Mp, map, pix);
  VKAS_LAUNCH_CHECK("points_prepare");
  return VKAS_OK;
}

extern "C" int vkas_points_gather_rows(const void* z, long ldz, int c0, int Ns, const float* stats,
                                       const float* const* dproj, int n_heads, long M, const int* pix, long Mp, void* zs,
                                       float* stats_s, float* dproj_s, int dtype, void* stream) {
  VKAS_CHECK(z && stats && dproj && pix && zs && stats_s && dproj_s, "vkas_points_gather_rows: null pointer");
  VKAS_CHECK(n_heads >= 1 && n_heads <= 4 && Ns > 0 && Ns % 8 == 0 && c0 >= 0 && c0 % 8 == 0 && c0 + Ns <= ldz && ldz % 8 == 0 &&
                 M > 0 && Mp > 0 && vkas_aligned16(z) && vkas_aligned16(zs),
             "vkas_points_gather_rows: bad sizes c0=%d Ns=%d ldz=%ld heads=%d", c0, Ns, ldz, n_heads);
  GatherRowsArgs a;
  a.n_heads = n_heads;
  for (int h = 0; h < 4; ++h) {
    a.dproj[h] = h < n_heads ? dproj[h] : nullptr;
    VKAS_CHECK(h >= n_heads || a.dproj[h], "vkas_points_gather_rows: null d(proj) of head %d", h);
  }
  hipStream_t st = vkas_stream(stream);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_points_gather_rows", {
    if (sizeof(T) != 2) {
      vkas_set_error("vkas_points_gather_rows: 16-bit activations only");
      return VKAS_E_ARG;
    }
    points_gather_rows_kernel<T><<<(unsigned)Mp, 64, 0, st>>>((const T*)z, ldz, c0, Ns, stats, a, M, pix, Mp, (T*)zs, stats_s,
                                                              dproj_s);
  })
  VKAS_LAUNCH_CHECK("points_gather_rows");
  return VKAS_OK;
}

extern "C" int vkas_points_gather_patches(const void* x, long ldx, int Cp, int B, int H, int W, const int* pix, long Mp,
                                          void* xs, int dtype, void* stream) {
  VKAS_CHECK(x && pix && xs, "vkas_points_gather_patches: null pointer");
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && ldx >= Cp && ldx % 8 == 0 && B > 0 && H > 0 && W > 0 && Mp > 0 && vkas_aligned16(x) &&
                 vkas_aligned16(xs),
             "vkas_points_gather_patches: bad sizes Cp=%d ldx=%ld", Cp, ldx);
  hipStream_t st = vkas_stream(stream);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_points_gather_patches", {
    points_gather_patches_kernel<T><<<(unsigned)Mp, 64, 0, st>>>((const T*)x, ldx, Cp, H, W, pix, (T*)xs);
  })
  VKAS_LAUNCH_CHECK("points_gather_patches");
  return VKAS_OK;
}

extern "C" int vkas_points_scatter3x3(const float* D, const int* pix, const int* map, long Mp, int B, int H, int W, int Cp,
                                      void* dx, long lddx, int dtype, void* stream) {
  VKAS_CHECK(D && pix && map && dx, "vkas_points_scatter3x3: null pointer");
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && lddx >= Cp && lddx % 8 == 0 && B > 0 && H > 0 && W > 0 && Mp > 0 && Mp * 9 < 0x7fffffffL &&
                 vkas_aligned16(D) && vkas_aligned16(dx),
             "vkas_points_scatter3x3: bad sizes Cp=%d lddx=%ld Mp=%ld", Cp, lddx, Mp);
  hipStream_t st = vkas_stream(stream);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_points_scatter3x3", {
    points_scatter3x3_kernel<T><<<(unsigned)(Mp * 9), 64, 0, st>>>(D, pix, map, H, W, Cp, (T*)dx, lddx);
  })
  VKAS_LAUNCH_CHECK("points_scatter3x3");
  return VKAS_OK;
}
