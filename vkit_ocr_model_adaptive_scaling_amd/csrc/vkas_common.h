// Shared host/device helpers for libvkas (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#include "../../include/vkas.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

void vkas_set_error(const char* fmt, ...);

#define VKAS_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      vkas_set_error(__VA_ARGS__);       \
      return VKAS_E_ARG;                 \
    }                                    \
  } while (0)

#define VKAS_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      vkas_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return VKAS_E_LAUNCH;                                                  \
    }                                                                        \
  } while (0)

#define VKAS_DISPATCH_DTYPE(dtype, NAME, ...)                  \
  if ((dtype) == VKAS_F32) {                                   \
    using T = float;                                           \
    __VA_ARGS__                                                \
  } else if ((dtype) == VKAS_BF16) {                           \
    using T = bf16_t;                                          \
    __VA_ARGS__                                                \
  } else if ((dtype) == VKAS_F16) {                            \
    using T = f16_t;                                           \
    __VA_ARGS__                                                \
  } else {                                                     \
    vkas_set_error("%s: unknown dtype %d", NAME, (int)(dtype)); \
    return VKAS_E_ARG;                                         \
  }

static inline bool vkas_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline hipStream_t vkas_stream(void* s) { return (hipStream_t)s; }
static inline long vkas_cdiv(long a, long b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// 8 consecutive elements <-> 8 floats.  Pointers must be 16-byte aligned (8-element granularity).
__device__ __forceinline__ void load8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void load8(const f16_t* p, float* v) {
  const f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void store8(f16_t* p, const float* v) {
  f16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (f16_t)v[i];
  *reinterpret_cast<f16x8*>(p) = a;
}
// 8 consecutive elements as loaded (16 bytes of 16-bit values, 32 bytes of floats), converted to 8 floats later: a kernel
// that issues all its loads in this form first keeps them in flight together (load8 converts at once, which makes the
// compiler wait for each load where it stands)
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
  __device__ __forceinline__ void keep_if(bool k) { if (!k) zero(); }
  __device__ __forceinline__ void unpack(float* v) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};
template <> struct Raw8<bf16_t> {
  bf16x8 a;
  __device__ __forceinline__ void load(const bf16_t* p) { a = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void zero() { for (int i = 0; i < 8; ++i) a[i] = (bf16_t)0.f; }
  __device__ __forceinline__ void keep_if(bool k) { if (!k) zero(); }
  __device__ __forceinline__ void unpack(float* v) const { for (int i = 0; i < 8; ++i) v[i] = (float)a[i]; }
};
template <> struct Raw8<f16_t> {
  f16x8 a;
  __device__ __forceinline__ void load(const f16_t* p) { a = *reinterpret_cast<const f16x8*>(p); }
  __device__ __forceinline__ void zero() { for (int i = 0; i < 8; ++i) a[i] = (f16_t)0.f; }
  __device__ __forceinline__ void keep_if(bool k) { if (!k) zero(); }
  __device__ __forceinline__ void unpack(float* v) const { for (int i = 0; i < 8; ++i) v[i] = (float)a[i]; }
};

__device__ __forceinline__ void load4(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float* v) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void load4(const f16_t* p, float* v) {
  const f16x4 a = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store4(f16_t* p, const float* v) {
  f16x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (f16_t)v[i];
  *reinterpret_cast<f16x4*>(p) = a;
}
__device__ __forceinline__ void store4(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float* v) {
  bf16x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x4*>(p) = a;
}

// exact (erf) GELU and its derivative: helper.py:100-101
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16-storage variants.  Phi(x) = 0.5 + x * P(x^2) on |x| <= 4.25 (degree-8 near-minimax fit, fp32 Horner; x clamped
// outside, where Phi is within 1.1e-5 of 0 / 1): |Phi error| <= 1.3e-5.  gelu(x) = max(x, -4.25) * Phi: the factor is
// clamped on the negative side too, so the left tail stays at -4.5e-5 instead of growing like 1.07e-5 * x (exact
// gelu -> 0 there); |gelu error| <= 5e-5 everywhere - below half a bf16 ulp for every |gelu(x)| > 0.03 - with 13 plain
// VALU operations and no transcendental (the erf / exp form cost as much as
// the MFMAs of a short-K layer GEMM in its epilogue).  phi(x) needs one v_exp_f32 and is only computed when the
// derivative is wanted.  fp32 storage keeps the exact forms.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf_x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
  const float u = xc * xc;
  float p = 5.5648210864e-11f;
  p = fmaf(p, u, -5.3277341912e-09f);
  p = fmaf(p, u, 2.2554223425e-07f);
  p = fmaf(p, u, -5.6264222408e-06f);
  p = fmaf(p, u, 9.3418689023e-05f);
  p = fmaf(p, u, -1.1085611169e-03f);
  p = fmaf(p, u, 9.8159725707e-03f);
  p = fmaf(p, u, -6.6344495031e-02f);
  p = fmaf(p, u, 3.9890234175e-01f);
  cdf = fmaf(xc, p, 0.5f);
  pdf_x = 0.39894228040143268f * __expf(-0.5f * x * x);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x);
template <> __device__ __forceinline__ float gelu_t<float>(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16_t>(float x) {
  float cdf, pdf;
  gelu_parts_fast(x, cdf, pdf);
  return fmaxf(x, -4.25f) * cdf;
}
template <> __device__ __forceinline__ float gelu_t<f16_t>(float x) { return gelu_t<bf16_t>(x); }
template <typename T> __device__ __forceinline__ float dgelu_t(float x);
template <> __device__ __forceinline__ float dgelu_t<float>(float x) { return dgelu_f(x); }
// gelu'(x) = Phi(x) + x phi(x) = 0.5 + x * Q(x^2) on |x| <= 4.5 (degree 9, |error| <= 1.8e-4 of a value in [−0.13, 1.13];
// clamped outside): the derivative-only sites (GELU' epilogue of the MLP dgrad) need no exp either.
template <> __device__ __forceinline__ float dgelu_t<bf16_t>(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
  const float u = xc * xc;
  float p = -2.2107989521e-11f;
  p = fmaf(p, u, 2.5213160948e-09f);
  p = fmaf(p, u, -1.2680528206e-07f);
  p = fmaf(p, u, 3.7219336900e-06f);
  p = fmaf(p, u, -7.1221943086e-05f);
  p = fmaf(p, u, 9.4054174145e-04f);
  p = fmaf(p, u, -8.8158577153e-03f);
  p = fmaf(p, u, 5.8609299903e-02f);
  p = fmaf(p, u, -2.6492567818e-01f);
  p = fmaf(p, u, 7.9762614621e-01f);
  return fmaf(xc, p, 0.5f);
}

// Two elements per instruction (round 4): CDNA3/4 issue v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at the rate of their scalar forms,
// i.e. a polynomial on a PAIR costs what it costs on one element; the constants ride in SGPRs (op_sel broadcast).  Same operations in
// the same order per component as gelu_t / dgelu_t: bit-identical results.  fp32 storage keeps the exact (erf) forms per component.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_splat(float c) { return f32x2{c, c}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T> __device__ __forceinline__ f32x2 gelu2_t(f32x2 x) {
  if constexpr (std::is_same<T, float>::value) {
    return f32x2{gelu_f(x.x), gelu_f(x.y)};
  } else {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -4.25f, 4.25f), __builtin_amdgcn_fmed3f(x.y, -4.25f, 4.25f)};
    const f32x2 u = xc * xc;
    f32x2 p = pk_splat(5.5648210864e-11f);
    p = pk_fma(p, u, pk_splat(-5.3277341912e-09f));
    p = pk_fma(p, u, pk_splat(2.2554223425e-07f));
    p = pk_fma(p, u, pk_splat(-5.6264222408e-06f));
    p = pk_fma(p, u, pk_splat(9.3418689023e-05f));
    p = pk_fma(p, u, pk_splat(-1.1085611169e-03f));
    p = pk_fma(p, u, pk_splat(9.8159725707e-03f));
    p = pk_fma(p, u, pk_splat(-6.6344495031e-02f));
    p = pk_fma(p, u, pk_splat(3.9890234175e-01f));
    const f32x2 cdf = pk_fma(xc, p, pk_splat(0.5f));
    // max(x, -4.25) as a median with a huge upper bound: one instruction, no canonicalising v_max in front of it
    const f32x2 xm = {__builtin_amdgcn_fmed3f(x.x, -4.25f, 3.0e38f), __builtin_amdgcn_fmed3f(x.y, -4.25f, 3.0e38f)};
    return xm * cdf;
  }
}
template <typename T> __device__ __forceinline__ f32x2 dgelu2_t(f32x2 x) {
  if constexpr (std::is_same<T, float>::value) {
    return f32x2{dgelu_f(x.x), dgelu_f(x.y)};
  } else {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x.y, -4.5f, 4.5f)};
    const f32x2 u = xc * xc;
    f32x2 p = pk_splat(-2.2107989521e-11f);
    p = pk_fma(p, u, pk_splat(2.5213160948e-09f));
    p = pk_fma(p, u, pk_splat(-1.2680528206e-07f));
    p = pk_fma(p, u, pk_splat(3.7219336900e-06f));
    p = pk_fma(p, u, pk_splat(-7.1221943086e-05f));
    p = pk_fma(p, u, pk_splat(9.4054174145e-04f));
    p = pk_fma(p, u, pk_splat(-8.8158577153e-03f));
    p = pk_fma(p, u, pk_splat(5.8609299903e-02f));
    p = pk_fma(p, u, pk_splat(-2.6492567818e-01f));
    p = pk_fma(p, u, pk_splat(7.9762614621e-01f));
    return pk_fma(xc, p, pk_splat(0.5f));
  }
}

// Sum over aligned groups of W consecutive lanes (W = 2 .. 64), result in every lane of the group.  Up to 16 lanes the
// exchange is a DPP operand of the add (quad permutes, half-row / row mirror: plain VALU speed); 32 adds one ds_swizzle
// (lane ^ 16) and 64 one ds_bpermute.  __shfl_xor lowers EVERY step to ds_bpermute_b32 (an LDS-crossbar round trip of ~60
// cycles): the per-row LayerNorm / head reductions were chains of 3-5 of them.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int W> __device__ __forceinline__ float group_sum(float v) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
  if constexpr (W >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (W >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (W >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
  if constexpr (W >= 16) v += dpp_f<0x140>(v);  // row_mirror
  if constexpr (W >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // lane ^ 16
  if constexpr (W >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <> __device__ __forceinline__ float dgelu_t<f16_t>(float x) { return dgelu_t<bf16_t>(x); }

// ---- LDS-DMA with the waits in the kernel's hands ------------------------------------------------------------------
// `buffer_load_dwordx4 ... lds` (1 KB per wave instruction: lane l's 16 bytes land at lds_addr + 16 l) written as inline
// assembly.  The compiler's own builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) makes SIInsertWaitcnts treat every later LDS
// access whose memory operand survived to that pass as possibly aliasing the DMA's destination: it then inserts
// `s_waitcnt vmcnt(0)` in front of it - found in round 4 in front of the fp32 bias read of mlp_chain_kernel, i.e. the "prefetch"
// of the next weight chunk was waited for at the top of every chunk.  A DMA the pass cannot see is ordered by the kernel's own
// counted `s_waitcnt vmcnt(N)` + barrier (which these kernels carry anyway); instructions the pass does know can only be
// over-waited for by it, never under-waited (vmcnt retires in order).
// rs: buffer descriptor words (base, base_hi | stride, num_records bytes, flags); lds_addr: wave-uniform LDS byte address.
__device__ __forceinline__ u32x4 vkas_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
  return r;
}
__device__ __forceinline__ void vkas_lds_dma16(u32x4 rs, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
template <typename T> __device__ __forceinline__ unsigned vkas_lds_addr(const T* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) T*)p;
}
