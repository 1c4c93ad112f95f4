// Fused head tail, backward side (the forward side lives in the NT GEMM epilogue, gemm_mfma.hip):
//   z (pre-LN conv output, kept by the forward) --LayerNorm--> u --GELU--> a --Linear(C -> oc <= 4)--> proj
// Given d(proj) this kernel recomputes u and a from z and the saved row statistics and produces dz together with the
// parameter gradients (gamma, beta, Wproj, bproj) in one pass over z: the (M, C) activation and its gradient never exist
// in HBM (model/upernext.py:215-223, model/fpn.py:165-183; backward of helper.py:96-101 + helper.py:18-22).
#include "vkas_common.h"

int vkas_colreduce_finalize(const float* partial, long P, int n, int ldp, float* out, int accumulate, hipStream_t st);

namespace {

// G = lanes per pixel row, one 8-channel vector each: 32 for heads up to 256 (padded) channels - ConvNeXt-T / S: 192-194 -,
// 64 up to 512 (Base: 256-258, Large: 384+)
constexpr int R = 2;    // rows in flight per lane group
#ifndef HT_R1
#define HT_R1 3
#endif

__global__ void pack_head_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ wproj, const float* __restrict__ bproj, int C, int oc,
                                        int pw, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6 * pw + 8) return;
  float v = 0.f;
  if (i < pw) v = i < C ? gamma[i] : 0.f;
  else if (i < 2 * pw) v = (i - pw) < C ? beta[i - pw] : 0.f;
  else if (i < 6 * pw) {
    const int q = (i - 2 * pw) / pw, c = (i - 2 * pw) - q * pw;
    v = (q < oc && c < C) ? wproj[(long)q * C + c] : 0.f;
  } else {
    const int q = i - 6 * pw;
    v = q < oc ? bproj[q] : 0.f;
  }
  out[i] = v;
}

// One launch handles ALL heads of a pass: lane group g (32 lanes) of a workgroup works on head g % NH of row g / NH, so a
// workgroup reads and writes whole rows of the shared (M, sum np) buffers (the per-head column slices are not
// 128-byte aligned; per-head launches left every row with partially written cache lines).
struct HeadBwdArgs {
  int n_heads, pw;
  int n0[4], np[4], c[4];
  const float* dproj[4];  // (M, 8) fp32 each
};

// OCM: upper bound of the heads' out_channels in this launch (1 for the rough pass: a quarter of the projection math and
// 48 fewer accumulator registers).  The z / statistics / d(proj) loads of the NEXT row pair are issued before the current
// pair is processed: with ~2 waves per SIMD the kernel was bound by the latency of one 16-byte load per lane per iteration.
template <typename T, int NH, int OCM, int G>
__global__ __launch_bounds__(256) void head_tail_bwd_kernel(const T* __restrict__ z, long ldz,
                                                            const float* __restrict__ params,
                                                            const float* __restrict__ stats, HeadBwdArgs a,
                                                            T* __restrict__ dz, long lddz, float* __restrict__ partial,
                                                            long M, long rows_per_block) {
  static_assert(NH == 1 || NH == 2 || NH == 4, "lane groups per row");
  // rows in flight per lane group: the 1-channel heads (the dense launches of a train step) leave the registers for four -
  // at two waves per SIMD the kernel is bound by the bytes it keeps in flight, not by its arithmetic alone
  constexpr int RR = OCM == 1 ? HT_R1 : R;
  constexpr int rpi = 256 / G / NH;  // rows per sub-iteration
  constexpr int VEC = 16 / sizeof(T);
  typedef uint4 raw_t;               // 16 bytes of z: 8 bf16 (one per lane and row) / first half of 8 fp32
  const int gl = threadIdx.x & (G - 1);
  const int grp = threadIdx.x / G;
  const int head = grp % NH, rl = grp / NH;
  const bool hok = head < a.n_heads;
  const int pw = a.pw, PS = 6 * pw + 8;
  const int C = hok ? a.c[head] : 1, np = hok ? a.np[head] : 0, n0 = hok ? a.n0[head] : 0;
  const float* hstats = stats + (long)head * M * 2;
  const float* hdproj = hok ? a.dproj[head] : nullptr;
  const int nvec = np >> 3;
  const long mbeg = (long)blockIdx.x * rows_per_block;
  const long mend = mbeg + rows_per_block < M ? mbeg + rows_per_block : M;
  const bool vok = hok && gl < nvec;
  // gamma | beta | Wproj[4] of every head live in LDS (each lane re-reads its 8-channel slice per row)
  __shared__ __attribute__((aligned(16))) float sp[NH * 6 * G * 8];
  for (int i = threadIdx.x; i < NH * 6 * pw; i += 256) {
    const int h = i / (6 * pw), r = i - h * 6 * pw;
    sp[i] = h < a.n_heads ? params[(long)h * PS + r] : 0.f;
  }
  __syncthreads();
  const float* hp = sp + head * 6 * pw;
  float dg[8], db[8], dwp[OCM][8], dbp[OCM];
#pragma unroll
  for (int c = 0; c < 8; ++c) { dg[c] = 0.f; db[c] = 0.f; }
#pragma unroll
  for (int q = 0; q < OCM; ++q) {
    dbp[q] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) dwp[q][c] = 0.f;
  }
  // channel validity of this lane's 8 channels as a mask of multiplicative 0/1 (pad channels carry gamma = beta = 0 and
  // Wproj = 0, so only the LayerNorm sums need it)
  float cm[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) cm[c] = (vok && gl * 8 + c < C) ? 1.f : 0.f;
  const float invC = 1.f / (float)C;

  // in-flight loads of one row pair
  Raw8<T> xr[RR];  // as loaded: converted when the pair is processed, not when it is requested
  float mean[RR], rstd[RR];
  float4 dp[RR];
  auto fetch = [&](long m0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const long m = m0 + (long)r * rpi + rl;
      const bool ok = hok && m < mend;
      mean[r] = ok ? hstats[2 * m] : 0.f;
      rstd[r] = ok ? hstats[2 * m + 1] : 0.f;
      dp[r] = ok ? *reinterpret_cast<const float4*>(hdproj + m * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[r].zero();
      if (ok && vok) xr[r].load(z + m * ldz + n0 + gl * 8);
    }
  };
  fetch(mbeg);
  for (long m0 = mbeg; m0 < mend; m0 += (long)RR * rpi) {
    float x[RR][8], mu[RR], rs[RR];
    float4 d4v[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      mu[r] = mean[r];
      rs[r] = rstd[r];
      d4v[r] = dp[r];
      xr[r].unpack(x[r]);
    }
    if (m0 + (long)RR * rpi < mend) fetch(m0 + (long)RR * rpi);  // next pair: in flight behind this pair's arithmetic
    float gm[8], bt[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { gm[c] = 0.f; bt[c] = 0.f; }
    int lo = gl * 8;
    asm volatile("" : "+v"(lo));  // opaque per iteration: keeps the parameter loads in the loop (LICM would pin their VGPRs)
    if (vok) {
      load8(hp + lo, gm);
      load8(hp + pw + lo, bt);
    }
    float g[RR][8], s1[RR], s2[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const float d4[4] = {d4v[r].x, d4v[r].y, d4v[r].z, d4v[r].w};
      s1[r] = 0.f;
      s2[r] = 0.f;
      if (gl == 0) {
#pragma unroll
        for (int q = 0; q < OCM; ++q) dbp[q] += d4[q];
      }
      float act[8], gp[8], da[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        // (no validity masks in the sums: a pad channel has gamma = beta = Wproj = 0, so its d(act) and everything derived
        // from it is zero whatever h is; a row beyond the block has zero statistics and zero d(proj))
        const float h = (x[r][c] - mu[r]) * rs[r];
        const float u = fmaf(h, gm[c], bt[c]);
        if constexpr (sizeof(T) == 2) {
          float cdf, pdf;
          gelu_parts_fast(u, cdf, pdf);
          act[c] = fmaxf(u, -4.25f) * cdf;  // as gelu_t<T>
          gp[c] = fmaf(u, pdf, cdf);
        } else {
          const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752f));
          const float pdf = 0.39894228040143268f * __expf(-0.5f * u * u);
          act[c] = u * cdf;
          gp[c] = fmaf(u, pdf, cdf);
        }
        x[r][c] = h;
        da[c] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < OCM; ++q) {
        float wp[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) wp[c] = 0.f;
        if (vok) load8(hp + (2 + q) * pw + lo, wp);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          da[c] = fmaf(d4[q], wp[c], da[c]);
          dwp[q][c] = fmaf(d4[q], act[c], dwp[q][c]);
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float gg = da[c] * gp[c];
        g[r][c] = gg;
        dg[c] = fmaf(gg, x[r][c], dg[c]);
        db[c] += gg;
        const float dxh = gg * gm[c];
        s1[r] += dxh;
        s2[r] = fmaf(dxh, x[r][c], s2[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      s1[r] = group_sum<G>(s1[r]);
      s2[r] = group_sum<G>(s2[r]);
      s1[r] *= invC;
      s2[r] *= invC;
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const long m = m0 + (long)r * rpi + rl;
      if (m >= mend || !vok) continue;
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = cm[c] * rs[r] * (g[r][c] * gm[c] - s1[r] - x[r][c] * s2[r]);
      store8(dz + m * lddz + n0 + gl * 8, o);
    }
  }
  // reduce the per-thread column sums over the row lanes; partial row per head = dgamma | dbeta | dWp[4] | dbp[8]
  __shared__ float red[256 * 8];
  float* prow = partial + ((long)blockIdx.x * NH + head) * PS;
  auto reduce_store = [&](const float* vals, int dst_off) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) red[threadIdx.x * 8 + c] = vals[c];
    __syncthreads();
    if (rl == 0 && gl * 8 < pw) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < rpi; ++r) s += red[(((r * NH + head) * G) + gl) * 8 + c];
        prow[dst_off + gl * 8 + c] = vok ? s : 0.f;
      }
    }
  };
  reduce_store(dg, 0);
  reduce_store(db, pw);
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < OCM) reduce_store(dwp[q < OCM ? q : 0], (2 + q) * pw);
    else reduce_store(zero8, (2 + q) * pw);
  }
  float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < OCM; ++q) b8[q] = dbp[q];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 8; ++c) red[threadIdx.x * 8 + c] = b8[c];
  __syncthreads();
  if (rl == 0 && gl < 8) {
    float s = 0.f;
    for (int r = 0; r < rpi; ++r) s += red[((r * NH + head) * G) * 8 + gl];  // gl == 0 lanes hold the row sums of d(proj)
    prow[6 * pw + gl] = s;
  }
}

// Forward side as a kernel of its own, for heads wider than the 224 columns the GEMM epilogue holds (ConvNeXt-Base / Large:
// 256-258 / 384+ channels): the convolution writes z = acc + bias through its plain epilogue and this pass turns z into the
// projected channels and the LayerNorm statistics - the (M, C) activation still never exists, z is read once.
struct HeadFwdArgs {
  int n_heads, pw;
  int n0[4], np[4], c[4];
};
template <typename T, int NH, int G>
__global__ __launch_bounds__(256) void head_tail_fwd_kernel(const T* __restrict__ z, long ldz, const float* __restrict__ params,
                                                            float* __restrict__ stats, float* __restrict__ proj,
                                                            HeadFwdArgs a, long M) {
  constexpr int rpi = 256 / G / NH;
  const int gl = threadIdx.x & (G - 1);
  const int grp = threadIdx.x / G;
  const int head = grp % NH, rl = grp / NH;
  const bool hok = head < a.n_heads;
  const int pw = a.pw, PS = 6 * pw + 8;
  const int C = hok ? a.c[head] : 1, np = hok ? a.np[head] : 0, n0 = hok ? a.n0[head] : 0;
  const bool vok = hok && gl < (np >> 3);
  __shared__ __attribute__((aligned(16))) float sp[NH * (6 * G * 8 + 8)];
  for (int i = threadIdx.x; i < NH * PS; i += 256) {
    const int h = i / PS;
    sp[i] = h < a.n_heads ? params[i] : 0.f;
  }
  __syncthreads();
  const float* hp = sp + head * PS;
  float gm[8], bt[8], wp[4][8], cm[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    gm[c] = bt[c] = 0.f;
    cm[c] = (vok && gl * 8 + c < C) ? 1.f : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) wp[q][c] = 0.f;
  }
  if (vok) {
    load8(hp + gl * 8, gm);
    load8(hp + pw + gl * 8, bt);
#pragma unroll
    for (int q = 0; q < 4; ++q) load8(hp + (2 + q) * pw + gl * 8, wp[q]);
  }
  const float invC = 1.f / (float)C;
  for (long m = (long)blockIdx.x * rpi + rl; m < M; m += (long)gridDim.x * rpi) {
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = 0.f;
    if (vok) load8(z + m * ldz + n0 + gl * 8, x);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += x[c] * cm[c];
    const float mean = group_sum<G>(s) * invC;
    float q2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float d = (x[c] - mean) * cm[c];
      q2 = fmaf(d, d, q2);
    }
    const float rstd = rsqrtf(group_sum<G>(q2) * invC + 1e-6f);
    float pr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float act = gelu_t<T>((x[c] - mean) * rstd * gm[c] + bt[c]) * cm[c];  // pad channels: gamma = beta = 0
#pragma unroll
      for (int q = 0; q < 4; ++q) pr[q] = fmaf(act, wp[q][c], pr[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) pr[q] = group_sum<G>(pr[q]);
    if (gl == 0 && hok) {
      const float4 bp = *reinterpret_cast<const float4*>(hp + 6 * pw);
      float* po = proj + ((long)head * M + m) * 8;
      *reinterpret_cast<float4*>(po) = make_float4(pr[0] + bp.x, pr[1] + bp.y, pr[2] + bp.z, pr[3] + bp.w);
      *reinterpret_cast<float4*>(po + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (stats) {
        stats[((long)head * M + m) * 2] = mean;
        stats[((long)head * M + m) * 2 + 1] = rstd;
      }
    }
  }
}

static inline int ht_lanes(int pw) { return pw <= 256 ? 32 : 64; }
static inline long ht_rows_per_block(long M, int G) {
  long r = vkas_cdiv(M > 0 ? M : 1, 1024);
  if (r < 32) r = 32;
  const long q = (long)(HT_R1 > R ? HT_R1 : R) * (256 / G);  // multiple of the rows per iteration for every NH and OCM
  return vkas_cdiv(r, q) * q;
}

}  // namespace

extern "C" int vkas_pack_head_params(const float* gamma, const float* beta, const float* wproj, const float* bproj, int C,
                                     int oc, int pw, float* out, void* stream) {
  VKAS_CHECK(gamma && beta && wproj && bproj && out, "vkas_pack_head_params: null pointer");
  VKAS_CHECK(C > 0 && oc >= 1 && oc <= 4 && pw % 8 == 0 && pw >= C && pw <= 512, "vkas_pack_head_params: bad C=%d oc=%d pw=%d", C,
             oc, pw);
  pack_head_params_kernel<<<(unsigned)vkas_cdiv(6 * pw + 8, 256), 256, 0, vkas_stream(stream)>>>(gamma, beta, wproj, bproj, C,
                                                                                                oc, pw, out);
  VKAS_LAUNCH_CHECK("pack_head_params");
  return VKAS_OK;
}

extern "C" size_t vkas_head_tail_bwd_ws_bytes(long M, int pw) {
  return (size_t)vkas_cdiv(M > 0 ? M : 1, ht_rows_per_block(M, ht_lanes(pw))) * 4 * (size_t)(6 * pw + 8) * sizeof(float);
}

extern "C" int vkas_head_tail_fwd(const void* z, long ldz, const vkas_head_desc* hd, long M, int dtype, void* stream) {
  VKAS_CHECK(z && hd && hd->params && hd->proj, "vkas_head_tail_fwd: null pointer");
  const int pw = hd->pw, nh = hd->n_heads;
  VKAS_CHECK(nh >= 1 && nh <= 4 && pw % 8 == 0 && pw >= 8 && pw <= 512, "vkas_head_tail_fwd: bad descriptor");
  HeadFwdArgs a;
  a.n_heads = nh;
  a.pw = pw;
  for (int h = 0; h < 4; ++h) {
    a.n0[h] = h < nh ? hd->n0[h] : 0;
    a.np[h] = h < nh ? hd->np[h] : 0;
    a.c[h] = h < nh ? hd->c[h] : 1;
    if (h < nh) {
      VKAS_CHECK(a.np[h] % 8 == 0 && a.np[h] > 0 && a.np[h] <= pw && a.c[h] > 0 && a.c[h] <= a.np[h] && a.n0[h] % 8 == 0 &&
                     a.n0[h] + a.np[h] <= ldz && hd->oc[h] >= 1 && hd->oc[h] <= 4,
                 "vkas_head_tail_fwd: bad head %d", h);
    }
  }
  VKAS_CHECK(ldz % 8 == 0 && vkas_aligned16(z) && vkas_aligned16(hd->params) && vkas_aligned16(hd->proj),
             "vkas_head_tail_fwd: bad strides/alignment");
  if (M <= 0) return VKAS_OK;
  hipStream_t st = vkas_stream(stream);
  const int GL = ht_lanes(pw);
  const int NH = nh == 1 ? 1 : (nh == 2 ? 2 : 4);
  const long rows_per_iter = 256 / GL / NH;
  long blocks = vkas_cdiv(M, rows_per_iter * 8);  // ~8 rows per lane group
  if (blocks > 4096) blocks = 4096;
#define VKAS_HF(NHV)                                                                                                          \
  do {                                                                                                                        \
    if (GL == 32)                                                                                                             \
      head_tail_fwd_kernel<T, NHV, 32><<<(unsigned)blocks, 256, 0, st>>>((const T*)z, ldz, hd->params, hd->stats, hd->proj, a, M); \
    else                                                                                                                      \
      head_tail_fwd_kernel<T, NHV, 64><<<(unsigned)blocks, 256, 0, st>>>((const T*)z, ldz, hd->params, hd->stats, hd->proj, a, M); \
  } while (0)
  VKAS_DISPATCH_DTYPE(dtype, "vkas_head_tail_fwd", {
    if (NH == 1) VKAS_HF(1);
    else if (NH == 2) VKAS_HF(2);
    else VKAS_HF(4);
  })
#undef VKAS_HF
  VKAS_LAUNCH_CHECK("head_tail_fwd");
  return VKAS_OK;
}

extern "C" int vkas_head_tail_bwd(const void* z, long ldz, const vkas_head_desc* hd, const float* const* dproj, void* dz,
                                  long lddz, float* dparams, float* ws, size_t ws_bytes, long M, int dtype, void* stream) {
  VKAS_CHECK(z && hd && dproj && dz && dparams && ws && hd->params && hd->stats, "vkas_head_tail_bwd: null pointer");
  const int pw = hd->pw, nh = hd->n_heads;
  VKAS_CHECK(nh >= 1 && nh <= 4 && pw % 8 == 0 && pw >= 8 && pw <= 512, "vkas_head_tail_bwd: bad descriptor");
  HeadBwdArgs a;
  a.n_heads = nh;
  a.pw = pw;
  for (int h = 0; h < 4; ++h) {
    a.n0[h] = h < nh ? hd->n0[h] : 0;
    a.np[h] = h < nh ? hd->np[h] : 0;
    a.c[h] = h < nh ? hd->c[h] : 1;
    a.dproj[h] = h < nh ? dproj[h] : nullptr;
    if (h < nh) {
      VKAS_CHECK(a.np[h] % 8 == 0 && a.np[h] > 0 && a.np[h] <= pw && a.c[h] > 0 && a.c[h] <= a.np[h] && a.n0[h] % 8 == 0 &&
                     a.n0[h] + a.np[h] <= ldz && a.n0[h] + a.np[h] <= lddz && a.dproj[h] && vkas_aligned16(a.dproj[h]),
                 "vkas_head_tail_bwd: bad head %d", h);
    }
  }
  VKAS_CHECK(ldz % 8 == 0 && lddz % 8 == 0 && vkas_aligned16(z) && vkas_aligned16(dz) && vkas_aligned16(hd->params),
             "vkas_head_tail_bwd: bad strides/alignment");
  VKAS_CHECK(ws_bytes >= vkas_head_tail_bwd_ws_bytes(M, pw), "vkas_head_tail_bwd: workspace too small");
  hipStream_t st = vkas_stream(stream);
  const int PS = 6 * pw + 8;
  if (M <= 0) {
    (void)hipMemsetAsync(dparams, 0, (size_t)nh * PS * sizeof(float), st);
    return VKAS_OK;
  }
  const int GL = ht_lanes(pw);
  const long rpb = ht_rows_per_block(M, GL);
  const long P = vkas_cdiv(M, rpb);
  const int NH = nh == 1 ? 1 : (nh == 2 ? 2 : 4);
  int ocm = 1;
  for (int h = 0; h < nh; ++h) ocm = hd->oc[h] > ocm ? hd->oc[h] : ocm;
  VKAS_CHECK(ocm <= 4, "vkas_head_tail_bwd: out_channels %d > 4", ocm);
#define VKAS_HT(NHV, OCV)                                                                                                      \
  do {                                                                                                                         \
    if (GL == 32)                                                                                                              \
      head_tail_bwd_kernel<T, NHV, OCV, 32><<<(unsigned)P, 256, 0, st>>>((const T*)z, ldz, hd->params, hd->stats, a, (T*)dz, lddz, ws, M, rpb); \
    else                                                                                                                       \
      head_tail_bwd_kernel<T, NHV, OCV, 64><<<(unsigned)P, 256, 0, st>>>((const T*)z, ldz, hd->params, hd->stats, a, (T*)dz, lddz, ws, M, rpb); \
  } while (0)
  VKAS_DISPATCH_DTYPE(dtype, "vkas_head_tail_bwd", {
    if (NH == 1) { if (ocm == 1) VKAS_HT(1, 1); else VKAS_HT(1, 4); }
    else if (NH == 2) { if (ocm == 1) VKAS_HT(2, 1); else VKAS_HT(2, 4); }
    else { if (ocm == 1) VKAS_HT(4, 1); else VKAS_HT(4, 4); }
  })
#undef VKAS_HT
  VKAS_LAUNCH_CHECK("head_tail_bwd");
  // partial rows are (block, lane-group head) pairs: NH * PS floats per block; heads beyond n_heads are zero
  return vkas_colreduce_finalize(ws, P, nh * PS, NH * PS, dparams, 0, st);
}
