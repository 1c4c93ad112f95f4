// LayerNorm over the channel axis of NHWC activations (+ optional exact GELU), forward and backward:
// helper.py:96-101 as used in convnext.py:32,85,121, upernext.py:28,42, fpn.py:25,35,45.
// Also the backward of the layer-scale / stochastic-depth / residual epilogue (convnext.py:56-58).
//
// One pixel row is handled by a group of G lanes (G = smallest power of two >= Cp/8, max 64), each lane
// holding 8-channel vectors in registers, so a row is read exactly once; statistics are two-pass in
// registers (mean, then centred variance).  Backward keeps per-thread dgamma/dbeta partial sums over the
// rows a workgroup walks, reduces them through LDS and writes one partial row per workgroup; a fixed-order
// finalize kernel adds the partials (deterministic, no float atomics).
#include "vkas_common.h"

int vkas_colreduce_finalize(const float* partial, long P, int n, int ldp, float* out, int accumulate, hipStream_t st);

namespace {

constexpr int MAXV_LIMIT = 4;  // vectors per lane: Cp <= 8 * 64 * 4 = 2048

// runtime group width (wave-uniform): the same DPP / swizzle steps as group_sum<W> in vkas_common.h
__device__ __forceinline__ float group_sum(float v, int G) {
  if (G >= 2) v += dpp_f<0xB1>(v);
  if (G >= 4) v += dpp_f<0x4E>(v);
  if (G >= 8) v += dpp_f<0x141>(v);
  if (G >= 16) v += dpp_f<0x140>(v);
  if (G >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

static inline int pick_group(int nvec) {
  int g = 1;
  while (g < nvec && g < 64) g <<= 1;
  return g;
}

// Tuning knobs of the row kernels, measured with profiles/bench_norm.py at the four stage shapes of config #3 (round 3; the
// defaults are the fastest): rows in flight per lane group - backward 4 instead of 2: +0...+30 % time (registers cost two of
// the four waves per SIMD), forward 8: +25 %, 2: same; workgroups per launch 512 / 2048 instead of 1024: +0...+15 %;
// requesting the next iteration's rows before processing the current ones (LN_BWD_PREFETCH): +10...+30 %; scale-residual
// backward with 2 / 4 rows per lane and trip: no change.
#ifndef LN_BWD_R1
#define LN_BWD_R1 2
#endif
#ifndef LN_FWD_R1
#define LN_FWD_R1 4
#endif
#ifndef LN_BLOCKS
#define LN_BLOCKS 1024
#endif
#ifndef SR_UNROLL
#define SR_UNROLL 1
#endif
#ifndef LN_BWD_PREFETCH
#define LN_BWD_PREFETCH 0
#endif

// R rows per lane group are in flight together (all loads issued before the first reduction): a single 16-byte
// load per lane leaves the memory system mostly idle, 4 of them per lane reach the HBM-bound regime.
template <typename T, int MAXV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, long ldy,
                                                            float* __restrict__ stats, long M, int C, int Cp, int G,
                                                            int act_gelu) {
  const int nvec = Cp >> 3;
  const int gl = threadIdx.x & (G - 1);
  const int rpi = 256 / G;
  const long mb = ((long)blockIdx.x * R) * rpi + threadIdx.x / G;
  float v[R][MAXV][8];
  Raw8<T> raw[R][MAXV];  // every load of the R rows is issued before the first conversion
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long m = mb + (long)r * rpi;
    const long mc = m < M ? m : M - 1;
    s[r] = 0.f;
    // unconditional loads from clamped addresses (no load under a partial EXEC mask, cf. head.hip): what a lane beyond the
    // row count or the width reads is masked out of the sums below and never stored
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = gl + i * G;
      raw[r][i].load(x + mc * ldx + (vi < nvec ? vi : 0) * 8);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) raw[r][i].unpack(v[r][i]);
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = gl + i * G;
#pragma unroll
      for (int c = 0; c < 8; ++c) s[r] += (vi * 8 + c < C) ? v[r][i][c] : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = group_sum(s[r], G) / (float)C;  // mean
  float q[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    q[r] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = gl + i * G;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float d = v[r][i][c] - s[r];
        q[r] += (vi < nvec && vi * 8 + c < C) ? d * d : 0.f;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) q[r] = rsqrtf(group_sum(q[r], G) / (float)C + 1e-6f);  // rstd
  // gamma / beta are zero padded to Cp, so pad channels come out as act(0) = 0 without a branch
  float gv[MAXV][8], bv[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = gl + i * G;
    load8(gamma + (vi < nvec ? vi : 0) * 8, gv[i]);
    load8(beta + (vi < nvec ? vi : 0) * 8, bv[i]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long m = mb + (long)r * rpi;
    if (m >= M) continue;
    if (gl == 0 && stats) {
      stats[2 * m] = s[r];
      stats[2 * m + 1] = q[r];
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = gl + i * G;
      if (vi >= nvec) continue;
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float u = (v[r][i][c] - s[r]) * q[r] * gv[i][c] + bv[i][c];
        if (act_gelu) u = gelu_t<T>(u);
        o[c] = u;
      }
      store8(y + m * ldy + vi * 8, o);
    }
  }
}

template <typename T, int MAXV, int R>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ stats, const T* __restrict__ dy,
                                                            long lddy, T* __restrict__ dx, long lddx,
                                                            float* __restrict__ partial, long M, int C, int Cp, int G,
                                                            int act_gelu, long rows_per_block) {
  const int nvec = Cp >> 3;
  const int gl = threadIdx.x & (G - 1);
  const int rl = threadIdx.x / G;
  const int rpi = 256 / G;  // rows per sub-iteration
  const long mbeg = (long)blockIdx.x * rows_per_block;
  const long mend = mbeg + rows_per_block < M ? mbeg + rows_per_block : M;
  float dg[MAXV][8], db[MAXV][8], gm[MAXV][8], bt[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = gl + i * G;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dg[i][c] = 0.f;
      db[i][c] = 0.f;
    }
    // zero padded to Cp by the caller; a lane beyond the width reads the first slice (its products are masked by cok below)
    load8(gamma + (vi < nvec ? vi : 0) * 8, gm[i]);
    load8(beta + (vi < nvec ? vi : 0) * 8, bt[i]);
  }
  // LN_BWD_PREFETCH: the rows of the NEXT iteration are requested (as raw 16-byte pieces) before the current ones are
  // processed, so a lane keeps twice the bytes in flight
  Raw8<T> nx[R][MAXV], nd[R][MAXV];
  float nmean[R], nrstd[R];
  auto fetch = [&](long m0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long m = m0 + (long)r * rpi + rl;
      const long mc = m < mend ? m : mbeg;  // unconditional loads from clamped addresses; the cok masks below drop the rest
      const float2 st2 = *reinterpret_cast<const float2*>(stats + 2 * mc);
      nmean[r] = st2.x;
      nrstd[r] = st2.y;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vic = gl + i * G < nvec ? gl + i * G : 0;
        nx[r][i].load(x + mc * ldx + vic * 8);
        nd[r][i].load(dy + mc * lddy + vic * 8);
      }
    }
  };
  if (LN_BWD_PREFETCH) fetch(mbeg);
  for (long m0 = mbeg; m0 < mend; m0 += (long)R * rpi) {
    float xh[R][MAXV][8], g[R][MAXV][8];  // first hold the raw x / dy, then x-hat / effective gradient
    float mean[R], rstd[R];
    if (!LN_BWD_PREFETCH) fetch(m0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mean[r] = nmean[r];
      rstd[r] = nrstd[r];
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        nx[r][i].unpack(xh[r][i]);
        nd[r][i].unpack(g[r][i]);
      }
    }
    if (LN_BWD_PREFETCH && m0 + (long)R * rpi < mend) fetch(m0 + (long)R * rpi);
    float s1[R], s2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = m0 + (long)r * rpi + rl < mend;
      s1[r] = 0.f;
      s2[r] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vi = gl + i * G;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bool cok = ok && vi < nvec && (vi * 8 + c < C);
          const float h = cok ? (xh[r][i][c] - mean[r]) * rstd[r] : 0.f;
          float gg = cok ? g[r][i][c] : 0.f;
          if (act_gelu) gg *= dgelu_t<T>(h * gm[i][c] + bt[i][c]);
          xh[r][i][c] = h;
          g[r][i][c] = gg;
          dg[i][c] += gg * h;
          db[i][c] += gg;
          const float dxh = gg * gm[i][c];
          s1[r] += dxh;
          s2[r] += dxh * h;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s1[r] = group_sum(s1[r], G) / (float)C;
      s2[r] = group_sum(s2[r], G) / (float)C;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long m = m0 + (long)r * rpi + rl;
      if (m >= mend) continue;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vi = gl + i * G;
        if (vi >= nvec) continue;
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bool cok = vi * 8 + c < C;
          o[c] = cok ? rstd[r] * (g[r][i][c] * gm[i][c] - s1[r] - xh[r][i][c] * s2[r]) : 0.f;
        }
        store8(dx + m * lddx + vi * 8, o);
      }
    }
  }
  // reduce the per-thread column sums over the row lanes of the workgroup
  __shared__ float red[256 * 8];
  float* prow = partial + (long)blockIdx.x * 2 * Cp;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i * G < nvec) {  // uniform over the workgroup
      const int vi = gl + i * G;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) red[threadIdx.x * 8 + c] = pass ? db[i][c] : dg[i][c];
        __syncthreads();
        if (rl == 0 && vi < nvec) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float s = 0.f;
            for (int r = 0; r < rpi; ++r) s += red[(r * G + gl) * 8 + c];
            prow[(pass ? Cp : 0) + vi * 8 + c] = s;
          }
        }
      }
    }
  }
}

// dz = dout*rs[b]*cs[c]; partial sums of dout*rs*z (-> dscale) and dz (-> dbias)
template <typename T>
__global__ __launch_bounds__(256) void scale_res_bwd_kernel(const T* __restrict__ dout, long lddo,
                                                            const T* __restrict__ z, long ldz,
                                                            const float* __restrict__ colscale,
                                                            const float* __restrict__ rowscale, int rows_per_image,
                                                            T* __restrict__ dz, long lddz, float* __restrict__ partial,
                                                            long M, int Cp, long rows_per_block) {
  const int nvec = Cp >> 3;
  const int lanes_r = 256 / nvec;
  const int v = threadIdx.x % nvec;
  const int rl = threadIdx.x / nvec;
  float a1[8], a2[8], cs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { a1[c] = 0.f; a2[c] = 0.f; cs[c] = colscale[v * 8 + c]; }
  const long mbeg = (long)blockIdx.x * rows_per_block;
  const long mend = mbeg + rows_per_block < M ? mbeg + rows_per_block : M;
  if (rl < lanes_r) {
    // SR_UNROLL rows per lane and iteration, all their loads issued before the first use
    for (long m0 = mbeg + rl; m0 < mend; m0 += (long)SR_UNROLL * lanes_r) {
      Raw8<T> rd[SR_UNROLL], rz[SR_UNROLL];
      float rs[SR_UNROLL];
#pragma unroll
      for (int u = 0; u < SR_UNROLL; ++u) {
        const long m = m0 + (long)u * lanes_r;
        rd[u].zero();
        rz[u].zero();
        rs[u] = 0.f;
        if (m < mend) {
          rd[u].load(dout + m * lddo + v * 8);
          rz[u].load(z + m * ldz + v * 8);
          rs[u] = rowscale ? rowscale[m / rows_per_image] : 1.f;
        }
      }
#pragma unroll
      for (int u = 0; u < SR_UNROLL; ++u) {
        const long m = m0 + (long)u * lanes_r;
        if (m >= mend) break;
        float d[8], zz[8], o[8];
        rd[u].unpack(d);
        rz[u].unpack(zz);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float t = d[c] * rs[u];
          a1[c] += t * zz[c];
          o[c] = t * cs[c];
          a2[c] += o[c];
        }
        store8(dz + m * lddz + v * 8, o);
      }
    }
  }
  __shared__ float red[256 * 8];
  float* prow = partial + (long)blockIdx.x * 2 * Cp;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) red[threadIdx.x * 8 + c] = pass ? a2[c] : a1[c];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float s = 0.f;
        for (int r = 0; r < lanes_r; ++r) s += red[(r * nvec + v) * 8 + c];
        prow[(pass ? Cp : 0) + v * 8 + c] = s;
      }
    }
  }
}

// about 1024 workgroups per launch (4 per CU); never fewer than 32 rows per workgroup so that small feature maps
// (stage 2/3: 8K-32K pixels) still spread over the whole chip
static inline long rows_per_block_for(long M, long quantum) {
  long r = vkas_cdiv(M > 0 ? M : 1, LN_BLOCKS);
  if (r < 32) r = 32;
  return vkas_cdiv(r, quantum) * quantum;
}

}  // namespace

extern "C" int vkas_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                  float* stats, long M, int C, int Cp, int act_gelu, int dtype, void* stream) {
  VKAS_CHECK(x && y && gamma && beta, "vkas_layernorm_fwd: null pointer");
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && Cp <= 2048 && C > 0 && C <= Cp, "vkas_layernorm_fwd: bad C=%d Cp=%d", C, Cp);
  VKAS_CHECK(ldx >= Cp && ldy >= Cp && ldx % 8 == 0 && ldy % 8 == 0 && vkas_aligned16(x) && vkas_aligned16(y),
             "vkas_layernorm_fwd: bad strides/alignment");
  if (M <= 0) return VKAS_OK;
  const int G = pick_group(Cp >> 3);
  const long rows = 256 / G;
  const int vpl = (int)vkas_cdiv(Cp >> 3, G);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_layernorm_fwd", {
    hipStream_t st = vkas_stream(stream);
    if (vpl == 1)
      layernorm_fwd_kernel<T, 1, LN_FWD_R1><<<(unsigned)vkas_cdiv(M, rows * LN_FWD_R1), 256, 0, st>>>((const T*)x, ldx, gamma, beta, (T*)y, ldy, stats, M, C, Cp, G, act_gelu);
    else if (vpl == 2)
      layernorm_fwd_kernel<T, 2, 2><<<(unsigned)vkas_cdiv(M, rows * 2), 256, 0, st>>>((const T*)x, ldx, gamma, beta, (T*)y, ldy, stats, M, C, Cp, G, act_gelu);
    else
      layernorm_fwd_kernel<T, 4, 1><<<(unsigned)vkas_cdiv(M, rows), 256, 0, st>>>((const T*)x, ldx, gamma, beta, (T*)y, ldy, stats, M, C, Cp, G, act_gelu);
  })
  VKAS_LAUNCH_CHECK("layernorm_fwd");
  return VKAS_OK;
}

extern "C" size_t vkas_layernorm_bwd_ws_bytes(long M, int Cp) {
  const int G = pick_group(Cp >> 3);
  const long rpb = rows_per_block_for(M, LN_BWD_R1 * (256 / G));
  return (size_t)vkas_cdiv(M > 0 ? M : 1, rpb) * 2 * (size_t)Cp * sizeof(float);
}

extern "C" long vkas_layernorm_bwd_parts(long M, int Cp) {  // partial rows (2 Cp floats each: dgamma | dbeta) left in ws
  if (M <= 0) return 0;
  const int G = pick_group(Cp >> 3);
  return vkas_cdiv(M, rows_per_block_for(M, LN_BWD_R1 * (256 / G)));
}

extern "C" int vkas_layernorm_bwd(const void* x, long ldx, const float* gamma, const float* beta, const float* stats,
                                  const void* dy, long lddy, void* dx, long lddx, float* dgamma, float* dbeta,
                                  float* ws, size_t ws_bytes, long M, int C, int Cp, int act_gelu, int dtype,
                                  void* stream) {
  VKAS_CHECK(x && dy && dx && gamma && beta && stats && ws && (dgamma != nullptr) == (dbeta != nullptr),
             "vkas_layernorm_bwd: null pointer");
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && Cp <= 2048 && C > 0 && C <= Cp, "vkas_layernorm_bwd: bad C=%d Cp=%d", C, Cp);
  VKAS_CHECK(ldx >= Cp && lddy >= Cp && lddx >= Cp && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 &&
                 vkas_aligned16(x) && vkas_aligned16(dy) && vkas_aligned16(dx),
             "vkas_layernorm_bwd: bad strides/alignment");
  VKAS_CHECK(ws_bytes >= vkas_layernorm_bwd_ws_bytes(M, Cp), "vkas_layernorm_bwd: workspace too small");
  hipStream_t st = vkas_stream(stream);
  if (M <= 0) {
    if (dgamma) {
      (void)hipMemsetAsync(dgamma, 0, Cp * sizeof(float), st);
      (void)hipMemsetAsync(dbeta, 0, Cp * sizeof(float), st);
    }
    return VKAS_OK;
  }
  const int G = pick_group(Cp >> 3);
  const long rpb = rows_per_block_for(M, LN_BWD_R1 * (256 / G));
  const long P = vkas_cdiv(M, rpb);
  const int vpl = (int)vkas_cdiv(Cp >> 3, G);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_layernorm_bwd", {
    if (vpl == 1)
      layernorm_bwd_kernel<T, 1, LN_BWD_R1><<<(unsigned)P, 256, 0, st>>>((const T*)x, ldx, gamma, beta, stats, (const T*)dy, lddy,
                                                              (T*)dx, lddx, ws, M, C, Cp, G, act_gelu, rpb);
    else if (vpl == 2)
      layernorm_bwd_kernel<T, 2, 1><<<(unsigned)P, 256, 0, st>>>((const T*)x, ldx, gamma, beta, stats, (const T*)dy, lddy,
                                                              (T*)dx, lddx, ws, M, C, Cp, G, act_gelu, rpb);
    else
      layernorm_bwd_kernel<T, 4, 1><<<(unsigned)P, 256, 0, st>>>((const T*)x, ldx, gamma, beta, stats, (const T*)dy, lddy,
                                                              (T*)dx, lddx, ws, M, C, Cp, G, act_gelu, rpb);
  })
  VKAS_LAUNCH_CHECK("layernorm_bwd");
  if (!dgamma) return VKAS_OK;  // the caller sums the vkas_layernorm_bwd_parts() partial rows itself (vkas_finalize_many)
  if (dbeta == dgamma + Cp) return vkas_colreduce_finalize(ws, P, 2 * Cp, 2 * Cp, dgamma, 0, st);  // one launch
  int rc = vkas_colreduce_finalize(ws, P, Cp, 2 * Cp, dgamma, 0, st);
  if (rc) return rc;
  return vkas_colreduce_finalize(ws + Cp, P, Cp, 2 * Cp, dbeta, 0, st);
}

extern "C" size_t vkas_scale_res_bwd_ws_bytes(long M, int Cp) {
  const long rpb = rows_per_block_for(M, 1);
  return (size_t)vkas_cdiv(M > 0 ? M : 1, rpb) * 2 * (size_t)Cp * sizeof(float);
}

extern "C" long vkas_scale_res_bwd_parts(long M, int Cp) {
  (void)Cp;
  return M <= 0 ? 0 : vkas_cdiv(M, rows_per_block_for(M, 1));
}

extern "C" int vkas_scale_res_bwd(const void* dout, long lddo, const void* z, long ldz, const float* colscale,
                                  const float* rowscale, int rows_per_image, void* dz, long lddz, float* dscale,
                                  float* dbias, float* ws, size_t ws_bytes, long M, int Cp, int dtype, void* stream) {
  VKAS_CHECK(dout && z && colscale && dz && ws && (dscale != nullptr) == (dbias != nullptr), "vkas_scale_res_bwd: null pointer");
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && Cp <= 2048 && rows_per_image > 0, "vkas_scale_res_bwd: bad Cp=%d", Cp);
  VKAS_CHECK(lddo >= Cp && ldz >= Cp && lddz >= Cp && lddo % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0 &&
                 vkas_aligned16(dout) && vkas_aligned16(z) && vkas_aligned16(dz),
             "vkas_scale_res_bwd: bad strides/alignment");
  const long rpb = rows_per_block_for(M, 1);
  const long P = vkas_cdiv(M > 0 ? M : 1, rpb);
  VKAS_CHECK(ws_bytes >= (size_t)P * 2 * Cp * sizeof(float), "vkas_scale_res_bwd: workspace too small");
  hipStream_t st = vkas_stream(stream);
  if (M <= 0) {
    if (dscale) {
      (void)hipMemsetAsync(dscale, 0, Cp * sizeof(float), st);
      (void)hipMemsetAsync(dbias, 0, Cp * sizeof(float), st);
    }
    return VKAS_OK;
  }
  VKAS_DISPATCH_DTYPE(dtype, "vkas_scale_res_bwd", {
    scale_res_bwd_kernel<T><<<(unsigned)P, 256, 0, st>>>((const T*)dout, lddo, (const T*)z, ldz, colscale, rowscale,
                                                         rows_per_image, (T*)dz, lddz, ws, M, Cp, rpb);
  })
  VKAS_LAUNCH_CHECK("scale_res_bwd");
  if (!dscale) return VKAS_OK;  // partial rows stay in ws (vkas_scale_res_bwd_parts of them, 2 Cp floats: dscale | dbias)
  if (dbias == dscale + Cp) return vkas_colreduce_finalize(ws, P, 2 * Cp, 2 * Cp, dscale, 0, st);  // one launch
  int rc = vkas_colreduce_finalize(ws, P, Cp, 2 * Cp, dscale, 0, st);
  if (rc) return rc;
  return vkas_colreduce_finalize(ws + Cp, P, Cp, 2 * Cp, dbias, 0, st);
}
