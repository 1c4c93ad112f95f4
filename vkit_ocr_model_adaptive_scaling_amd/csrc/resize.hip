// F.interpolate on NHWC activations: bilinear, align_corners=False (upernext.py:79,178-195,237-244) and
// nearest (fpn.py:125-142,197-204), plus nn.AdaptiveAvgPool2d (upernext.py:62).  All are gathers: a thread
// owns one 8-channel vector of one destination pixel, consecutive lanes walk channels then x, so every
// load/store is a coalesced 16-byte access.  The backward kernels are gathers too (each source pixel sums
// the destination pixels that referenced it, enumerated from the same index function as the forward), so
// no float atomics and bit-reproducible results.
#include "vkas_common.h"

namespace {

// source coordinate of F.interpolate(bilinear, align_corners=False): src = max((dst+0.5)*in/out-0.5, 0)
__device__ __forceinline__ void bilinear_src(int dst, float scale, int n_in, int& i0, int& i1, float& w1) {
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  w1 = src - (float)i0;
}
// nearest: src = min(floor(dst * in/out), in-1), evaluated in integers
__device__ __forceinline__ int nearest_src(int dst, int n_in, int n_out) {
  const int s = (int)(((long)dst * n_in) / n_out);
  return s < n_in - 1 ? s : n_in - 1;
}

// weight with which destination index `dst` reads source index `src_i` along one axis (0 if it does not)
__device__ __forceinline__ float axis_weight(int dst, float scale, int n_in, int n_out, int src_i, int mode) {
  if (mode == 0) {
    int i0, i1;
    float w1;
    bilinear_src(dst, scale, n_in, i0, i1, w1);
    return (i0 == src_i ? 1.f - w1 : 0.f) + (i1 == src_i ? w1 : 0.f);
  }
  return nearest_src(dst, n_in, n_out) == src_i ? 1.f : 0.f;
}

// Workgroups are dealt round-robin over the 8 XCDs (private L2s) in launch order, x fastest: neighbouring image rows read
// the same source rows, so give every XCD one contiguous run of rows instead of every 8th row (any grid: a bijection).
__device__ __forceinline__ int xcd_row(unsigned bx, unsigned rows) {
  const unsigned xcd = bx & 7u, slot = bx >> 3;
  const unsigned q8 = rows >> 3, r8 = rows & 7u;
  return (int)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot);
}

// grid.x = B*Hout rows, grid.y = chunks of 256 (pixel, vector) pairs of one row: no 64-bit index arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                         int Hin, int Win, int Hout, int Wout, int nvec, int mode,
                                                         int accumulate) {
  const int idx = blockIdx.y * 256 + threadIdx.x;
  const int ox = idx / nvec;
  const int v = idx - ox * nvec;
  if (ox >= Wout) return;
  const int row = xcd_row(blockIdx.x, gridDim.x);
  const int b = row / Hout;
  const int oy = row - b * Hout;
  const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
  float o[8];
  const T* xb = x + (long)b * Hin * Win * ldx + v * 8;
  if (mode == 0) {
    int y0, y1, x0, x1;
    float wy, wx;
    bilinear_src(oy, sy, Hin, y0, y1, wy);
    bilinear_src(ox, sx, Win, x0, x1, wx);
    float a[8], bq[8], c[8], d[8];
    load8(xb + ((long)y0 * Win + x0) * ldx, a);
    load8(xb + ((long)y0 * Win + x1) * ldx, bq);
    load8(xb + ((long)y1 * Win + x0) * ldx, c);
    load8(xb + ((long)y1 * Win + x1) * ldx, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float top = a[k] * (1.f - wx) + bq[k] * wx;
      const float bot = c[k] * (1.f - wx) + d[k] * wx;
      o[k] = top * (1.f - wy) + bot * wy;
    }
  } else {
    const int iy = nearest_src(oy, Hin, Hout), ix = nearest_src(ox, Win, Wout);
    load8(xb + ((long)iy * Win + ix) * ldx, o);
  }
  T* dst = y + (((long)b * Hout + oy) * Wout + ox) * ldy + v * 8;
  if (accumulate) {
    float t[8];
    load8(dst, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] += t[k];
  }
  store8(dst, o);
}

// destination range [lo, hi] along one axis that reads source index i (tight; empty if lo > hi)
__device__ __forceinline__ void dst_range(int i, int n_in, int n_out, float scale, int mode, int& lo, int& hi) {
  const float r = (float)n_out / (float)n_in;
  if (mode == 0) {
    lo = (int)floorf(((float)i - 1.f) * r) - 2;
    hi = (int)ceilf(((float)i + 1.5f) * r) + 2;
  } else {
    lo = (int)floorf((float)i * r) - 2;
    hi = (int)ceilf(((float)i + 1.f) * r) + 2;
  }
  if (i == n_in - 1) hi = n_out - 1;  // the last source index also collects every clamped destination
  if (lo < 0) lo = 0;
  if (hi > n_out - 1) hi = n_out - 1;
  while (lo <= hi && axis_weight(lo, scale, n_in, n_out, i, mode) == 0.f) ++lo;
  while (hi >= lo && axis_weight(hi, scale, n_in, n_out, i, mode) == 0.f) --hi;
}

// dx[b,iy,ix] (+)= sum over destination pixels (oy,ox) of weight(oy->iy) * weight(ox->ix) * dy[b,oy,ox]
// grid.x = B*Hin rows, grid.y = chunks of 256 (pixel, vector) pairs of one source row.
// R = destination columns whose x weight is evaluated ONCE per source pixel and kept in registers (a x4 / x8 upsample - the
// necks' final resize to level-0 size, upernext.py:191-195 - has 8 / 16 of them): the index function is then evaluated
// nx + ny times per source pixel instead of nx * ny times (at x8 that arithmetic, not the 256 L1 / L2 resident loads, was
// 90% of the launch).  Columns beyond R (never at the instantiated factors) take the per-pair evaluation.  Same products,
// same order (oy ascending, then ox ascending) as before: bit-identical sums.
template <typename T, int R>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx,
                                                         long lddx, int Hin, int Win, int Hout, int Wout, int nvec,
                                                         int mode, int accumulate) {
  const int idx = blockIdx.y * 256 + threadIdx.x;
  const int ix = idx / nvec;
  const int v = idx - ix * nvec;
  if (ix >= Win) return;
  const int row = xcd_row(blockIdx.x, gridDim.x);
  const int b = row / Hin;
  const int iy = row - b * Hin;
  const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const T* db = dy + (long)b * Hout * Wout * lddy + v * 8;
  int oy_lo, oy_hi, ox_lo, ox_hi;
  dst_range(iy, Hin, Hout, sy, mode, oy_lo, oy_hi);
  dst_range(ix, Win, Wout, sx, mode, ox_lo, ox_hi);
  float wxr[R];
#pragma unroll
  for (int j = 0; j < R; ++j) wxr[j] = ox_lo + j <= ox_hi ? axis_weight(ox_lo + j, sx, Win, Wout, ix, mode) : 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float wyv = axis_weight(oy, sy, Hin, Hout, iy, mode);
    if (wyv == 0.f) continue;
    const T* rowp = db + ((long)oy * Wout + ox_lo) * lddy;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (wxr[j] == 0.f) continue;
      float t[8];
      load8(rowp + (long)j * lddy, t);
      const float wgt = wyv * wxr[j];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, t[k], acc[k]);
    }
    for (int ox = ox_lo + R; ox <= ox_hi; ++ox) {
      const float wxv = axis_weight(ox, sx, Win, Wout, ix, mode);
      if (wxv == 0.f) continue;
      float t[8];
      load8(db + ((long)oy * Wout + ox) * lddy, t);
      const float wgt = wyv * wxv;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, t[k], acc[k]);
    }
  }
  T* dst = dx + (((long)b * Hin + iy) * Win + ix) * lddx + v * 8;
  if (accumulate) {
    float t[8];
    load8(dst, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += t[k];
  }
  store8(dst, acc);
}

// Large ratios (the necks' x4 / x8 resize to level-0 size, upernext.py:191-195): a source pixel collects 8 x 8 resp. 16 x 16
// destinations, so the one-pass gather above has few threads (one per source vector) that each walk 64 - 256 loads in series
// and every destination vector is fetched by four source pixels: 1.1 TB/s.  Bilinear / nearest weights are separable, so the
// backward is two gathers: along x (a thread per (destination row, source column) sums its 2r destination columns - contiguous
// vectors, 16 x more threads) into an fp32 intermediate of Hout x Win pixels, then along y.  Deterministic (no atomics); the
// products w_y * (sum_x w_x * dy) replace sum_y sum_x (w_y * w_x) * dy, equal up to fp32 rounding.
template <typename T, int R>
__global__ __launch_bounds__(256) void resize_bwd_x_kernel(const T* __restrict__ dy, long lddy, float* __restrict__ tmp,
                                                           int Win, int Wout, int nvec, int mode) {
  const int idx = blockIdx.y * 256 + threadIdx.x;
  const int ix = idx / nvec;
  const int v = idx - ix * nvec;
  if (ix >= Win) return;
  const int row = xcd_row(blockIdx.x, gridDim.x);   // b * Hout + oy
  const float sx = (float)Win / (float)Wout;
  int ox_lo, ox_hi;
  dst_range(ix, Win, Wout, sx, mode, ox_lo, ox_hi);
  const T* rowp = dy + ((long)row * Wout + ox_lo) * lddy + v * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (ox_lo + j > ox_hi) break;
    const float w = axis_weight(ox_lo + j, sx, Win, Wout, ix, mode);
    if (w == 0.f) continue;
    float t[8];
    load8(rowp + (long)j * lddy, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, t[k], acc[k]);
  }
  for (int ox = ox_lo + R; ox <= ox_hi; ++ox) {
    const float w = axis_weight(ox, sx, Win, Wout, ix, mode);
    if (w == 0.f) continue;
    float t[8];
    load8(dy + ((long)row * Wout + ox) * lddy + v * 8, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, t[k], acc[k]);
  }
  store8(tmp + ((long)row * Win + ix) * (nvec * 8) + v * 8, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_y_kernel(const float* __restrict__ tmp, T* __restrict__ dx, long lddx,
                                                           int Hin, int Win, int Hout, int nvec, int mode, int accumulate) {
  const int idx = blockIdx.y * 256 + threadIdx.x;
  const int ix = idx / nvec;
  const int v = idx - ix * nvec;
  if (ix >= Win) return;
  const int row = xcd_row(blockIdx.x, gridDim.x);
  const int b = row / Hin;
  const int iy = row - b * Hin;
  const float sy = (float)Hin / (float)Hout;
  int oy_lo, oy_hi;
  dst_range(iy, Hin, Hout, sy, mode, oy_lo, oy_hi);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float w = axis_weight(oy, sy, Hin, Hout, iy, mode);
    if (w == 0.f) continue;
    float t[8];
    load8(tmp + (((long)b * Hout + oy) * Win + ix) * (nvec * 8) + v * 8, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, t[k], acc[k]);
  }
  T* dst = dx + (((long)b * Hin + iy) * Win + ix) * lddx + v * 8;
  if (accumulate) {
    float t[8];
    load8(dst, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += t[k];
  }
  store8(dst, acc);
}

// ---- exact x2 bilinear (every top-down step and head upsample of upernext.py / fpn.py at the default factors) ---------------
// out[2i] = 0.25 in[i-1] + 0.75 in[i], out[2i+1] = 0.75 in[i] + 0.25 in[i+1] per axis (borders clamp).  A thread owns one
// SOURCE pixel vector and writes its 2 x 2 destinations from the 3 x 3 neighbourhood: 9 loads + 4 stores per 4 outputs instead
// of 16 + 4, and the per-output index arithmetic of the generic gather disappears.  Same products and sums, in the same
// order, as resize_fwd_kernel (bit-identical results).
template <typename T>
__global__ __launch_bounds__(256) void resize2x_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                           int Hin, int Win, int nvec, int accumulate) {
  const int idx = blockIdx.y * 256 + threadIdx.x;
  const int j = idx / nvec;
  const int v = idx - j * nvec;
  if (j >= Win) return;
  const int row = xcd_row(blockIdx.x, gridDim.x);
  const int b = row / Hin;
  const int i = row - b * Hin;
  const int im = i > 0 ? i - 1 : 0, ip = i + 1 < Hin ? i + 1 : Hin - 1;
  const int jm = j > 0 ? j - 1 : 0, jp = j + 1 < Win ? j + 1 : Win - 1;
  const T* xb = x + (long)b * Hin * Win * ldx + v * 8;
  float n[3][3][8];
  const int ys[3] = {im, i, ip}, xs[3] = {jm, j, jp};
  {
    Raw8<T> raw[3][3];  // all nine requests first, conversions afterwards
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) raw[r][c].load(xb + ((long)ys[r] * Win + xs[c]) * ldx);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) raw[r][c].unpack(n[r][c]);
  }
  // destination 2i + a reads (row A, row B, weight of B): a = 0: (i-1, i, 0.75), at i = 0 the generic index function gives
  // (0, 1, 0); a = 1: (i, i+1 clamped, 0.25).  Same along x.
  const float wy[2] = {i == 0 ? 0.f : 0.75f, 0.25f}, wx[2] = {j == 0 ? 0.f : 0.75f, 0.25f};
  const int Hout = 2 * Hin, Wout = 2 * Win;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // rows: a == 0 -> (n[0], i == 0 ? n[2] : n[1]); a == 1 -> (n[1], n[2]); columns likewise
        const float a00 = a == 0 ? (c2 == 0 ? n[0][0][k] : n[0][1][k]) : (c2 == 0 ? n[1][0][k] : n[1][1][k]);
        const float a01 = a == 0 ? (c2 == 0 ? (j == 0 ? n[0][2][k] : n[0][1][k]) : n[0][2][k])
                                 : (c2 == 0 ? (j == 0 ? n[1][2][k] : n[1][1][k]) : n[1][2][k]);
        const float r1c0 = a == 0 ? (i == 0 ? (c2 == 0 ? n[2][0][k] : n[2][1][k]) : (c2 == 0 ? n[1][0][k] : n[1][1][k]))
                                  : (c2 == 0 ? n[2][0][k] : n[2][1][k]);
        const float r1c1 = a == 0 ? (i == 0 ? (c2 == 0 ? (j == 0 ? n[2][2][k] : n[2][1][k]) : n[2][2][k])
                                            : (c2 == 0 ? (j == 0 ? n[1][2][k] : n[1][1][k]) : n[1][2][k]))
                                  : (c2 == 0 ? (j == 0 ? n[2][2][k] : n[2][1][k]) : n[2][2][k]);
        const float top = a00 * (1.f - wx[c2]) + a01 * wx[c2];
        const float bot = r1c0 * (1.f - wx[c2]) + r1c1 * wx[c2];
        o[k] = top * (1.f - wy[a]) + bot * wy[a];
      }
      T* dst = y + (((long)b * Hout + 2 * i + a) * Wout + 2 * j + c2) * ldy + v * 8;
      if (accumulate) {
        float t[8];
        load8(dst, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] += t[k];
      }
      store8(dst, o);
    }
  }
}

// Backward of the same: source pixel (i, j) collects destinations 2i-1 .. 2i+2 x 2j-1 .. 2j+2 (those inside the map) with
// the forward's weights - taken from the same index function, ascending, zero weights skipped: the sums of
// resize_bwd_kernel bit for bit, without its range search.  (A 2 x 2 source block per thread - 9 loads per source pixel
// instead of 16 - measured 4 % slower: the loads hit L1, the kernel is not bound by their count.)
template <typename T>
__global__ __launch_bounds__(256) void resize2x_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx,
                                                           long lddx, int Hin, int Win, int nvec, int accumulate) {
  const int idx = blockIdx.y * 256 + threadIdx.x;
  const int ix = idx / nvec;
  const int v = idx - ix * nvec;
  if (ix >= Win) return;
  const int row = xcd_row(blockIdx.x, gridDim.x);
  const int b = row / Hin;
  const int iy = row - b * Hin;
  const int Hout = 2 * Hin, Wout = 2 * Win;
  float wy4[4], wx4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oy = 2 * iy - 1 + q, ox = 2 * ix - 1 + q;
    wy4[q] = (oy >= 0 && oy < Hout) ? axis_weight(oy, 0.5f, Hin, Hout, iy, 0) : 0.f;
    wx4[q] = (ox >= 0 && ox < Wout) ? axis_weight(ox, 0.5f, Win, Wout, ix, 0) : 0.f;
  }
  const T* db = dy + ((long)b * Hout * Wout) * lddy + v * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // All sixteen destinations are requested before the first one is used (a destination outside the map - zero weight - reads
  // its clamped neighbour instead of branching around the load); the sums below still skip zero weights, so they are those of
  // the generic kernel term for term.
  Raw8<T> raw[4][4];
#pragma unroll
  for (int jy = 0; jy < 4; ++jy) {
    int oy = 2 * iy - 1 + jy;
    oy = oy < 0 ? 0 : (oy >= Hout ? Hout - 1 : oy);
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      int ox = 2 * ix - 1 + jx;
      ox = ox < 0 ? 0 : (ox >= Wout ? Wout - 1 : ox);
      raw[jy][jx].load(db + ((long)oy * Wout + ox) * lddy);
    }
  }
#pragma unroll
  for (int jy = 0; jy < 4; ++jy) {
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      const float wgt = wy4[jy] * wx4[jx];
      if (wgt == 0.f) continue;
      float t[8];
      raw[jy][jx].unpack(t);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, t[k], acc[k]);
    }
  }
  T* dst = dx + (((long)b * Hin + iy) * Win + ix) * lddx + v * 8;
  if (accumulate) {
    float t[8];
    load8(dst, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += t[k];
  }
  store8(dst, acc);
}

// Tiny sources (PPM maps of 1..6 pixels upsampled to 32x32): one workgroup per source pixel, the destination
// footprint is spread over the threads and reduced through LDS.  nvec <= 256.
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_small_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx,
                                                               long lddx, int Hin, int Win, int Hout, int Wout, int nvec,
                                                               int mode, int accumulate) {
  __shared__ float red[256 * 8];
  const int ix = blockIdx.x % Win;
  const int iy = (blockIdx.x / Win) % Hin;
  const int b = blockIdx.x / (Win * Hin);
  const int lanes_p = 256 / nvec;
  const int v = threadIdx.x % nvec;
  const int pl = threadIdx.x / nvec;
  const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
  int oy_lo, oy_hi, ox_lo, ox_hi;
  dst_range(iy, Hin, Hout, sy, mode, oy_lo, oy_hi);
  dst_range(ix, Win, Wout, sx, mode, ox_lo, ox_hi);
  const int nx = ox_hi - ox_lo + 1, ny = oy_hi - oy_lo + 1;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < lanes_p && nx > 0 && ny > 0) {
    const T* db = dy + (long)b * Hout * Wout * lddy + v * 8;
    for (int p = pl; p < nx * ny; p += lanes_p) {
      const int oy = oy_lo + p / nx, ox = ox_lo + p % nx;
      const float wgt = axis_weight(oy, sy, Hin, Hout, iy, mode) * axis_weight(ox, sx, Win, Wout, ix, mode);
      if (wgt == 0.f) continue;
      float t[8];
      load8(db + ((long)oy * Wout + ox) * lddy, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, t[k], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  if (pl == 0) {
    for (int r = 1; r < lanes_p; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += red[(r * nvec + v) * 8 + k];
    T* dst = dx + (((long)b * Hin + iy) * Win + ix) * lddx + v * 8;
    if (accumulate) {
      float t[8];
      load8(dst, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += t[k];
    }
    store8(dst, acc);
  }
}

// bin i of AdaptiveAvgPool: [floor(i*n/s), ceil((i+1)*n/s))
__device__ __forceinline__ void pool_bin(int i, int n, int s, int& lo, int& hi) {
  lo = (i * n) / s;
  hi = ((i + 1) * n + s - 1) / s;
}

// one workgroup per (b, bin, group of vb channel vectors): the bin's pixels are spread over 256 / vb lanes per vector and
// reduced through LDS (a 1x1 pooling of a 32x32 map is 1024 pixels per bin: with all 96 vectors in one workgroup only
// two lanes shared them).  nvec <= 256.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                          int B, int H, int W, int s, int nvec_total, int vb) {
  __shared__ float red[256 * 8];
  const int bj = blockIdx.x % s;
  const int bi = (blockIdx.x / s) % s;
  const int b = blockIdx.x / (s * s);
  const int v_first = blockIdx.y * vb;
  const int nvec = (nvec_total - v_first < vb) ? nvec_total - v_first : vb;  // vectors of this workgroup
  x += v_first * 8;
  y += v_first * 8;
  const int lanes_p = 256 / nvec;
  const int v = threadIdx.x % nvec;
  const int pl = threadIdx.x / nvec;
  int y0, y1, x0, x1;
  pool_bin(bi, H, s, y0, y1);
  pool_bin(bj, W, s, x0, x1);
  const int nx = x1 - x0, np = (y1 - y0) * nx;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < lanes_p) {
    for (int p = pl; p < np; p += lanes_p) {
      const int yy = y0 + p / nx, xx = x0 + p % nx;
      float t[8];
      load8(x + (((long)b * H + yy) * W + xx) * ldx + v * 8, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += t[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  if (pl == 0) {
    for (int r = 1; r < lanes_p; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += red[(r * nvec + v) * 8 + k];
    const float inv = 1.f / (float)np;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= inv;
    store8(y + (((long)b * s + bi) * s + bj) * ldy + v * 8, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx,
                                                          long lddx, int B, int H, int W, int s, int nvec,
                                                          int accumulate) {
  const long total = (long)B * H * W * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    long p = i / nvec;
    const int xx = (int)(p % W);
    p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int bi = 0; bi < s; ++bi) {
      int y0, y1;
      pool_bin(bi, H, s, y0, y1);
      if (yy < y0 || yy >= y1) continue;
      for (int bj = 0; bj < s; ++bj) {
        int x0, x1;
        pool_bin(bj, W, s, x0, x1);
        if (xx < x0 || xx >= x1) continue;
        float t[8];
        load8(dy + (((long)b * s + bi) * s + bj) * lddy + v * 8, t);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(inv, t[k], acc[k]);
      }
    }
    T* dst = dx + (((long)b * H + yy) * W + xx) * lddx + v * 8;
    if (accumulate) {
      float t[8];
      load8(dst, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += t[k];
    }
    store8(dst, acc);
  }
}

static inline unsigned grid_for(long total) {
  long g = vkas_cdiv(total, 256);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int vkas_resize_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int Hin, int Win, int Hout,
                               int Wout, int Cp, int mode, int accumulate, int dtype, void* stream);

static int rs_check(const char* who, const void* a, long lda, const void* b, long ldb, int B, int Hin, int Win, int Hout,
                    int Wout, int Cp) {
  VKAS_CHECK(a && b && vkas_aligned16(a) && vkas_aligned16(b), "%s: null/misaligned tensor", who);
  VKAS_CHECK(B >= 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "%s: bad spatial dims", who);
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && lda >= Cp && ldb >= Cp && lda % 8 == 0 && ldb % 8 == 0, "%s: bad channels/strides",
             who);
  return VKAS_OK;
}

extern "C" int vkas_resize_fwd(const void* x, long ldx, void* y, long ldy, int B, int Hin, int Win, int Hout, int Wout,
                               int Cp, int mode, int accumulate, int dtype, void* stream) {
  int rc = rs_check("vkas_resize_fwd", x, ldx, y, ldy, B, Hin, Win, Hout, Wout, Cp);
  if (rc) return rc;
  VKAS_CHECK(mode == 0 || mode == 1, "vkas_resize_fwd: bad mode %d", mode);
  if (B == 0) return VKAS_OK;
  VKAS_CHECK((long)Wout * (Cp / 8) < (1L << 30) && vkas_cdiv((long)Wout * (Cp / 8), 256) <= 65535, "vkas_resize_fwd: row too wide");
  static const bool no2x = getenv("VKAS_RESIZE_NO2X") != nullptr;
  if (mode == 0 && Hout == 2 * Hin && Wout == 2 * Win && Hin > 1 && Win > 1 && !no2x) {
    dim3 grid2((unsigned)((long)B * Hin), (unsigned)vkas_cdiv((long)Win * (Cp / 8), 256));
    VKAS_DISPATCH_DTYPE(dtype, "vkas_resize_fwd", {
      resize2x_fwd_kernel<T><<<grid2, 256, 0, vkas_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, Hin, Win, Cp / 8, accumulate);
    })
    VKAS_LAUNCH_CHECK("resize2x_fwd");
    return VKAS_OK;
  }
  dim3 grid((unsigned)((long)B * Hout), (unsigned)vkas_cdiv((long)Wout * (Cp / 8), 256));
  VKAS_DISPATCH_DTYPE(dtype, "vkas_resize_fwd", {
    resize_fwd_kernel<T><<<grid, 256, 0, vkas_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, Hin, Win, Hout, Wout, Cp / 8,
                                                               mode, accumulate);
  })
  VKAS_LAUNCH_CHECK("resize_fwd");
  return VKAS_OK;
}

// the two-pass backward pays when a source pixel has many destinations and the map is large enough to need the threads
static bool resize_bwd_separable(int B, int Hin, int Win, int Hout, int Wout, int Cp) {
  static const bool off = getenv("VKAS_RESIZE_NO_SEPARABLE") != nullptr;
  const bool small = (long)Hin * Win <= 64 && (long)Hout * Wout >= 16L * Hin * Win && Cp / 8 <= 256;
  return !off && !small && Hout >= 3 * Hin && Wout >= 3 * Win && (long)B * Hout * Wout * (Cp / 8) >= (1L << 16);
}

extern "C" size_t vkas_resize_bwd_ws_bytes(int B, int Hin, int Win, int Hout, int Wout, int Cp) {
  return resize_bwd_separable(B, Hin, Win, Hout, Wout, Cp) ? (size_t)B * Hout * Win * Cp * sizeof(float) : 0;
}

extern "C" int vkas_resize_bwd_ws(const void* dy, long lddy, void* dx, long lddx, float* ws, size_t ws_bytes, int B, int Hin,
                                  int Win, int Hout, int Wout, int Cp, int mode, int accumulate, int dtype, void* stream) {
  if (B == 0 || !ws || !resize_bwd_separable(B, Hin, Win, Hout, Wout, Cp))
    return vkas_resize_bwd(dy, lddy, dx, lddx, B, Hin, Win, Hout, Wout, Cp, mode, accumulate, dtype, stream);
  int rc = rs_check("vkas_resize_bwd_ws", dy, lddy, dx, lddx, B, Hin, Win, Hout, Wout, Cp);
  if (rc) return rc;
  VKAS_CHECK(mode == 0 || mode == 1, "vkas_resize_bwd_ws: bad mode %d", mode);
  VKAS_CHECK(ws_bytes >= vkas_resize_bwd_ws_bytes(B, Hin, Win, Hout, Wout, Cp) && vkas_aligned16(ws),
             "vkas_resize_bwd_ws: workspace too small / misaligned");
  VKAS_CHECK((long)Win * (Cp / 8) < (1L << 30) && vkas_cdiv((long)Win * (Cp / 8), 256) <= 65535, "vkas_resize_bwd_ws: row too wide");
  const unsigned chunks = (unsigned)vkas_cdiv((long)Win * (Cp / 8), 256);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_resize_bwd_ws", {
    dim3 gx((unsigned)((long)B * Hout), chunks), gy((unsigned)((long)B * Hin), chunks);
    if (2 * vkas_cdiv(Wout, Win) <= 8)
      resize_bwd_x_kernel<T, 8><<<gx, 256, 0, vkas_stream(stream)>>>((const T*)dy, lddy, ws, Win, Wout, Cp / 8, mode);
    else
      resize_bwd_x_kernel<T, 16><<<gx, 256, 0, vkas_stream(stream)>>>((const T*)dy, lddy, ws, Win, Wout, Cp / 8, mode);
    resize_bwd_y_kernel<T><<<gy, 256, 0, vkas_stream(stream)>>>(ws, (T*)dx, lddx, Hin, Win, Hout, Cp / 8, mode, accumulate);
  })
  VKAS_LAUNCH_CHECK("resize_bwd_ws");
  return VKAS_OK;
}

extern "C" int vkas_resize_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int Hin, int Win, int Hout,
                               int Wout, int Cp, int mode, int accumulate, int dtype, void* stream) {
  int rc = rs_check("vkas_resize_bwd", dy, lddy, dx, lddx, B, Hin, Win, Hout, Wout, Cp);
  if (rc) return rc;
  VKAS_CHECK(mode == 0 || mode == 1, "vkas_resize_bwd: bad mode %d", mode);
  if (B == 0) return VKAS_OK;
  VKAS_CHECK((long)Win * (Cp / 8) < (1L << 30) && vkas_cdiv((long)Win * (Cp / 8), 256) <= 65535, "vkas_resize_bwd: row too wide");
  const bool small = (long)Hin * Win <= 64 && (long)Hout * Wout >= 16L * Hin * Win && Cp / 8 <= 256;
  static const bool no2x = getenv("VKAS_RESIZE_NO2X") != nullptr;
  const bool x2 = mode == 0 && Hout == 2 * Hin && Wout == 2 * Win && Hin > 1 && Win > 1 && !small && !no2x;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_resize_bwd", {
    if (x2) {
      dim3 grid((unsigned)((long)B * Hin), (unsigned)vkas_cdiv((long)Win * (Cp / 8), 256));
      resize2x_bwd_kernel<T><<<grid, 256, 0, vkas_stream(stream)>>>((const T*)dy, lddy, (T*)dx, lddx, Hin, Win, Cp / 8,
                                                                   accumulate);
    } else if (small) {
      resize_bwd_small_kernel<T><<<(unsigned)((long)B * Hin * Win), 256, 0, vkas_stream(stream)>>>(
          (const T*)dy, lddy, (T*)dx, lddx, Hin, Win, Hout, Wout, Cp / 8, mode, accumulate);
    } else {
      dim3 grid((unsigned)((long)B * Hin), (unsigned)vkas_cdiv((long)Win * (Cp / 8), 256));
      // destination columns per source pixel: at most 2 * ceil(Wout / Win) for bilinear (+ the clamped border run)
      if (2 * vkas_cdiv(Wout, Win) <= 8)
        resize_bwd_kernel<T, 8><<<grid, 256, 0, vkas_stream(stream)>>>((const T*)dy, lddy, (T*)dx, lddx, Hin, Win, Hout, Wout,
                                                                      Cp / 8, mode, accumulate);
      else
        resize_bwd_kernel<T, 16><<<grid, 256, 0, vkas_stream(stream)>>>((const T*)dy, lddy, (T*)dx, lddx, Hin, Win, Hout,
                                                                       Wout, Cp / 8, mode, accumulate);
    }
  })
  VKAS_LAUNCH_CHECK("resize_bwd");
  return VKAS_OK;
}

extern "C" int vkas_adaptive_avgpool_fwd(const void* x, long ldx, void* y, long ldy, int B, int H, int W, int s, int Cp,
                                         int dtype, void* stream) {
  VKAS_CHECK(s > 0, "vkas_adaptive_avgpool_fwd: bad s");
  int rc = rs_check("vkas_adaptive_avgpool_fwd", x, ldx, y, ldy, B, H, W, s, s, Cp);
  if (rc) return rc;
  if (B == 0) return VKAS_OK;
  VKAS_CHECK(Cp / 8 <= 256, "vkas_adaptive_avgpool_fwd: at most 2048 channels");
  VKAS_DISPATCH_DTYPE(dtype, "vkas_adaptive_avgpool_fwd", {
    // vectors per workgroup: fewer when the bins are large, so that ~64 pixels remain per lane
    const int nvec = Cp / 8;
    const long bin_px = (long)vkas_cdiv(H, s) * vkas_cdiv(W, s);
    int vb = nvec;
    while (vb > 4 && bin_px * vb > 64L * 256) vb = (vb + 1) / 2;
    dim3 grid((unsigned)((long)B * s * s), (unsigned)vkas_cdiv(nvec, vb));
    avgpool_fwd_kernel<T><<<grid, 256, 0, vkas_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, B, H, W, s, nvec, vb);
  })
  VKAS_LAUNCH_CHECK("adaptive_avgpool_fwd");
  return VKAS_OK;
}

extern "C" int vkas_adaptive_avgpool_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int H, int W, int s,
                                         int Cp, int accumulate, int dtype, void* stream) {
  VKAS_CHECK(s > 0, "vkas_adaptive_avgpool_bwd: bad s");
  int rc = rs_check("vkas_adaptive_avgpool_bwd", dy, lddy, dx, lddx, B, H, W, s, s, Cp);
  if (rc) return rc;
  const long total = (long)B * H * W * (Cp / 8);
  if (total == 0) return VKAS_OK;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_adaptive_avgpool_bwd", {
    avgpool_bwd_kernel<T><<<grid_for(total), 256, 0, vkas_stream(stream)>>>((const T*)dy, lddy, (T*)dx, lddx, B, H, W, s,
                                                                           Cp / 8, accumulate);
  })
  VKAS_LAUNCH_CHECK("adaptive_avgpool_bwd");
  return VKAS_OK;
}
