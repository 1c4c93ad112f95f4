// Pieces shared by the implicit-GEMM kernels: the A-operand gather (convolution geometry) and the
// fused epilogues.  See include/vkas.h for the semantics of each VKAS_EPI_* mode.
#pragma once
#include "vkas_common.h"

#ifndef VKAS_ABL
#define VKAS_ABL 0  // timing-only ablation switches, see gemm_mfma.hip
#endif

struct RowCoord {
  int b, oy, ox;
  bool ok;
};

__device__ __forceinline__ RowCoord decode_row(long m, long M, const vkas_conv_geom& g) {
  RowCoord r;
  r.ok = m < M;
  const long mm = r.ok ? m : 0;
  const int hw = g.Hout * g.Wout;
  r.b = (int)(mm / hw);
  const int rem = (int)(mm - (long)r.b * hw);
  r.oy = rem / g.Wout;
  r.ox = rem - r.oy * g.Wout;
  return r;
}

// Element offset of input pixel feeding output row `r` through tap (ky, kx); -1 when the tap falls in
// the zero padding or the row does not exist.
__device__ __forceinline__ long tap_offset(const RowCoord& r, int ky, int kx, const vkas_conv_geom& g) {
  const int iy = r.oy * g.stride - g.pad + ky;
  const int ix = r.ox * g.stride - g.pad + kx;
  if (!r.ok || (unsigned)iy >= (unsigned)g.Hin || (unsigned)ix >= (unsigned)g.Win) return -1;
  return (((long)r.b * g.Hin + iy) * g.Win + ix) * (long)g.ldx;
}

// Store 4 consecutive output columns n..n+3 of row m.  v holds the fp32 accumulators.
template <typename T>
__device__ __forceinline__ void epi_store4(const vkas_epilogue& e, long m, int n, float* v) {
  if (e.bias) {
    const float4 b = *reinterpret_cast<const float4*>(e.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  T* out = reinterpret_cast<T*>(e.out);
  switch (e.mode) {
    case VKAS_EPI_NONE:
      store4(out + m * e.ldo + n, v);
      break;
    case VKAS_EPI_GELU: {
      if (out) store4(out + m * e.ldo + n, v);  // the pre-activation is only kept for a backward pass
      float gv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) gv[i] = gelu_t<T>(v[i]);
      store4(reinterpret_cast<T*>(e.out2) + m * e.ldo2 + n, gv);
      break;
    }
    case VKAS_EPI_SCALE_RES: {
      if (e.out2) store4(reinterpret_cast<T*>(e.out2) + m * e.ldo2 + n, v);
      float r[4];
      load4(reinterpret_cast<const T*>(e.aux) + m * e.ldaux + n, r);
      const float rs = e.rowscale ? e.rowscale[m / e.rows_per_image] : 1.0f;
      const float4 cs = *reinterpret_cast<const float4*>(e.colscale + n);
      r[0] += rs * cs.x * v[0]; r[1] += rs * cs.y * v[1]; r[2] += rs * cs.z * v[2]; r[3] += rs * cs.w * v[3];
      store4(out + m * e.ldo + n, r);
      break;
    }
    case VKAS_EPI_DGELU: {
      float h[4];
      load4(reinterpret_cast<const T*>(e.aux) + m * e.ldaux + n, h);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= dgelu_t<T>(h[i]);
      store4(out + m * e.ldo + n, v);
      break;
    }
    case VKAS_EPI_ADD: {
      float r[4];
      load4(reinterpret_cast<const T*>(e.aux) + m * e.ldaux + n, r);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += r[i];
      store4(out + m * e.ldo + n, v);
      break;
    }
    case VKAS_EPI_PATCH: {
      const int tap = n / e.patch_Cp;
      const int c = n - tap * e.patch_Cp;
      const int ky = tap / e.patch;
      const int kx = tap - ky * e.patch;
      const int hw = e.patch_Hs * e.patch_Ws;
      const int b = (int)(m / hw);
      const int rem = (int)(m - (long)b * hw);
      const int y = rem / e.patch_Ws;
      const int x = rem - y * e.patch_Ws;
      const long pix = ((long)b * e.patch_Hs * e.patch + (long)y * e.patch + ky) * ((long)e.patch_Ws * e.patch) +
                       (long)x * e.patch + kx;
      store4(out + pix * e.ldo + c, v);
      break;
    }
    default:
      break;
  }
}

// 8 consecutive output columns n..n+7 of row m (n % 8 == 0): the coalesced form used by the MFMA kernels, whose
// accumulators are first transposed through LDS so that a lane owns a 16-byte piece of an output row.
// aux_pre: the 8 values of e.aux at (m, n) when the caller has requested them ahead of time (nt_epilogue issues all of a
// lane's requests before the tile is staged, so they are in flight together instead of one latency per row piece)
template <typename T>
__device__ __forceinline__ void epi_store8(const vkas_epilogue& e, long m, int n, float* v, const Raw8<T>* aux_pre = nullptr) {
  if (e.bias) {
    float b[8];
    load8(e.bias + n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  T* out = reinterpret_cast<T*>(e.out);
  switch (e.mode) {
    case VKAS_EPI_NONE:
      store8(out + m * e.ldo + n, v);
      break;
    case VKAS_EPI_GELU: {
#if (VKAS_ABL & 128) == 0
      if (out) store8(out + m * e.ldo + n, v);  // the pre-activation is only kept for a backward pass
#endif
      float gv[8];
#pragma unroll
#if (VKAS_ABL & 32) == 0
      for (int i = 0; i < 8; ++i) gv[i] = gelu_t<T>(v[i]);
#else
      for (int i = 0; i < 8; ++i) gv[i] = v[i] * 0.5f;
#endif
#if (VKAS_ABL & 64) == 0
      store8(reinterpret_cast<T*>(e.out2) + m * e.ldo2 + n, gv);
#else
      if (gv[0] == 1.2345f) store8(reinterpret_cast<T*>(e.out2) + m * e.ldo2 + n, gv);
#endif
      break;
    }
    case VKAS_EPI_SCALE_RES: {
      if (e.out2) store8(reinterpret_cast<T*>(e.out2) + m * e.ldo2 + n, v);
      float r[8], cs[8];
      if (aux_pre) aux_pre->unpack(r);
      else load8(reinterpret_cast<const T*>(e.aux) + m * e.ldaux + n, r);
      load8(e.colscale + n, cs);
      const float rs = e.rowscale ? e.rowscale[m / e.rows_per_image] : 1.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] += rs * cs[i] * v[i];
      store8(out + m * e.ldo + n, r);
      break;
    }
    case VKAS_EPI_DGELU: {
      float h[8];
      if (aux_pre) aux_pre->unpack(h);
      else load8(reinterpret_cast<const T*>(e.aux) + m * e.ldaux + n, h);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= dgelu_t<T>(h[i]);
      store8(out + m * e.ldo + n, v);
      break;
    }
    case VKAS_EPI_ADD: {
      float r[8];
      if (aux_pre) aux_pre->unpack(r);
      else load8(reinterpret_cast<const T*>(e.aux) + m * e.ldaux + n, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += r[i];
      store8(out + m * e.ldo + n, v);
      break;
    }
    case VKAS_EPI_PATCH: {
      const int tap = n / e.patch_Cp;
      const int c = n - tap * e.patch_Cp;
      const int ky = tap / e.patch;
      const int kx = tap - ky * e.patch;
      const int hw = e.patch_Hs * e.patch_Ws;
      const int b = (int)(m / hw);
      const int rem = (int)(m - (long)b * hw);
      const int y = rem / e.patch_Ws;
      const int x = rem - y * e.patch_Ws;
      const long pix = ((long)b * e.patch_Hs * e.patch + (long)y * e.patch + ky) * ((long)e.patch_Ws * e.patch) +
                       (long)x * e.patch + kx;
      store8(out + pix * e.ldo + c, v);
      break;
    }
    default:
      break;
  }
}

int vkas_check_geom(const char* who, const void* x, const vkas_conv_geom* g, int Np);
int vkas_check_epilogue(const char* who, const vkas_epilogue* e, int Np);
