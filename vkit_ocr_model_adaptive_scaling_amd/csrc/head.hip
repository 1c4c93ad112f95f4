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
  const int nvec = np >> 3;
  const long mbeg = (long)blockIdx.x * rows_per_block;
  const long mend = mbeg + rows_per_block < M ? mbeg + rows_per_block : M;
  const bool vok = hok && gl < nvec;
  // gamma | beta | Wproj[4] of every head live in LDS (each lane re-reads its 8-channel slice per row)
  // Each of a head's six vectors takes G * 8 floats, zero beyond pw: every lane of a group reads its 8-channel slice
  // UNCONDITIONALLY (lanes past the head's width get zeros).  Round 4: with these 16-byte LDS reads under a partial EXEC mask
  // (lanes >= np / 8 off) the last active 8-lane beat of the upper half-wave sporadically delivered stale registers when a
  // second process kept the CU's LDS busy (profiles/repro/repro_step_determinism5.py) - full-wave reads do not.
  constexpr int GW = G * 8;
  __shared__ __attribute__((aligned(16))) float sp[NH * 6 * GW];
  for (int i = threadIdx.x; i < NH * 6 * GW; i += 256) {
    const int h = i / (6 * GW), r = i - h * 6 * GW, k = r / GW, j = r - k * GW;
    sp[i] = (h < a.n_heads && j < pw) ? params[(long)h * PS + k * pw + j] : 0.f;
  }
  __syncthreads();
  const float* hp = sp + head * 6 * GW;
  // element pairs (round 4): packed fp32 arithmetic - v_pk_fma_f32 / v_pk_mul_f32 issue at the rate of their scalar forms, and
  // this kernel is bound by its vector instructions.  GELU' is the degree-9 polynomial of the MLP's GELU' epilogue (|error| <=
  // 1.8e-4) instead of Phi + u phi with a v_exp_f32 (quarter rate, and not packable).
  f32x2 dg[4], db[4], dwp[OCM][4];
  float dbp[OCM];
#pragma unroll
  for (int c = 0; c < 4; ++c) { dg[c] = f32x2{0.f, 0.f}; db[c] = f32x2{0.f, 0.f}; }
#pragma unroll
  for (int q = 0; q < OCM; ++q) {
    dbp[q] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dwp[q][c] = f32x2{0.f, 0.f};
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
  // every lane loads UNCONDITIONALLY from an address that is always valid (its row clamped into the block, a lane beyond the
  // head's width reads the head's first slice) and what it must not use is zeroed afterwards: no load under a partial EXEC mask
  const float* hdproj_c = a.dproj[hok ? head : 0];
  const float* hstats_c = hok ? hstats : stats;
  const T* zcol = z + n0 + (vok ? gl : 0) * 8;
  auto fetch = [&](long m0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const long m = m0 + (long)r * rpi + rl;
      const long mc = m < mend ? m : mbeg;
      const float2 st2 = *reinterpret_cast<const float2*>(hstats_c + 2 * mc);
      mean[r] = st2.x;
      rstd[r] = st2.y;
      dp[r] = *reinterpret_cast<const float4*>(hdproj_c + mc * 8);
      xr[r].load(zcol + mc * ldz);
    }
  };
  fetch(mbeg);
  for (long m0 = mbeg; m0 < mend; m0 += (long)RR * rpi) {
    float x[RR][8], mu[RR], rs[RR];
    float4 d4v[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const bool ok = hok && m0 + (long)r * rpi + rl < mend;  // the masks of the unconditional loads, applied on use
      mu[r] = ok ? mean[r] : 0.f;
      rs[r] = ok ? rstd[r] : 0.f;
      d4v[r] = ok ? dp[r] : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[r].keep_if(ok && vok);
      xr[r].unpack(x[r]);
    }
    if (m0 + (long)RR * rpi < mend) fetch(m0 + (long)RR * rpi);  // next pair: in flight behind this pair's arithmetic
    float gm[8], bt[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { gm[c] = 0.f; bt[c] = 0.f; }
    int lo = gl * 8;
    asm volatile("" : "+v"(lo));  // opaque per iteration: keeps the parameter loads in the loop (LICM would pin their VGPRs)
    load8(hp + lo, gm);
    load8(hp + GW + lo, bt);
    f32x2 g[RR][4], hh[RR][4];
    float s1[RR], s2[RR];
    f32x2 gm2[4], bt2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      gm2[c] = f32x2{gm[2 * c], gm[2 * c + 1]};
      bt2[c] = f32x2{bt[2 * c], bt[2 * c + 1]};
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const float d4[4] = {d4v[r].x, d4v[r].y, d4v[r].z, d4v[r].w};
      if (gl == 0) {
#pragma unroll
        for (int q = 0; q < OCM; ++q) dbp[q] += d4[q];
      }
      f32x2 act[4], gp[4], da[4];
      const f32x2 rs2 = pk_splat(rs[r]), nmr = pk_splat(-mu[r] * rs[r]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // (no validity masks in the sums: a pad channel has gamma = beta = Wproj = 0, so its d(act) and everything derived
        // from it is zero whatever h is; a row beyond the block has zero statistics and zero d(proj))
        const f32x2 h = pk_fma(f32x2{x[r][2 * c], x[r][2 * c + 1]}, rs2, nmr);
        const f32x2 u = pk_fma(h, gm2[c], bt2[c]);
        act[c] = gelu2_t<T>(u);
        gp[c] = dgelu2_t<T>(u);
        hh[r][c] = h;
        da[c] = f32x2{0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < OCM; ++q) {
        float wp[8];
        load8(hp + (2 + q) * GW + lo, wp);
        const f32x2 dq = pk_splat(d4[q]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          da[c] = pk_fma(dq, f32x2{wp[2 * c], wp[2 * c + 1]}, da[c]);
          dwp[q][c] = pk_fma(dq, act[c], dwp[q][c]);
        }
      }
      f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 gg = da[c] * gp[c];
        g[r][c] = gg;
        dg[c] = pk_fma(gg, hh[r][c], dg[c]);
        db[c] += gg;
        const f32x2 dxh = gg * gm2[c];
        t1 += dxh;
        t2 = pk_fma(dxh, hh[r][c], t2);
      }
      s1[r] = t1.x + t1.y;
      s2[r] = t2.x + t2.y;
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
      const f32x2 ns1 = pk_splat(-s1[r]), ns2 = pk_splat(-s2[r]), rs2 = pk_splat(rs[r]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 v = f32x2{cm[2 * c], cm[2 * c + 1]} * rs2 * pk_fma(hh[r][c], ns2, pk_fma(g[r][c], gm2[c], ns1));
        o[2 * c] = v.x;
        o[2 * c + 1] = v.y;
      }
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
  auto flat8 = [](const f32x2 (&v)[4], float (&f)[8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { f[2 * c] = v[c].x; f[2 * c + 1] = v[c].y; }
  };
  float f8[8];
  flat8(dg, f8);
  reduce_store(f8, 0);
  flat8(db, f8);
  reduce_store(f8, pw);
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < OCM) {
      flat8(dwp[q < OCM ? q : 0], f8);
      reduce_store(f8, (2 + q) * pw);
    } else {
      reduce_store(zero8, (2 + q) * pw);
    }
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
  // four rows per trip: their 16-byte pieces are requested together as raw registers before the first one is converted (round 4;
  // one row per trip left every row's memory round trip exposed: 1.30 ms for 573 k pixels x 4 heads of config #5)
  constexpr int RU = 4;
  const long stride = (long)gridDim.x * rpi;
  for (long m0 = (long)blockIdx.x * rpi + rl; m0 < M; m0 += RU * stride) {
    Raw8<T> raw[RU];
    // unconditional loads from clamped addresses, as in the backward kernel above: a lane beyond the head's width reads the
    // head's first slice (its channel mask cm is zero), a row beyond M is never used
    const T* zcol = z + n0 + (vok ? gl : 0) * 8;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long mu = m0 + u * stride;
      raw[u].load(zcol + (mu < M ? mu : M - 1) * ldz);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
    const long m = m0 + u * stride;
    if (m >= M) break;
    float x[8];
    raw[u].unpack(x);
    // element pairs: packed fp32 arithmetic (v_pk_fma_f32 ...) - the kernel is bound by its vector instructions, not by HBM
    f32x2 xp[4], s2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      xp[c] = f32x2{x[2 * c], x[2 * c + 1]};
      s2 = pk_fma(xp[c], f32x2{cm[2 * c], cm[2 * c + 1]}, s2);
    }
    const float mean = group_sum<G>(s2.x + s2.y) * invC;
    f32x2 q2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      xp[c] = (xp[c] - pk_splat(mean)) * f32x2{cm[2 * c], cm[2 * c + 1]};
      q2 = pk_fma(xp[c], xp[c], q2);
    }
    const float rstd = rsqrtf(group_sum<G>(q2.x + q2.y) * invC + 1e-6f);
    f32x2 pr2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // pad channels: gamma = beta = 0 and the mask
      const f32x2 g2 = f32x2{gm[2 * c], gm[2 * c + 1]} * pk_splat(rstd);
      const f32x2 act = gelu2_t<T>(pk_fma(xp[c], g2, f32x2{bt[2 * c], bt[2 * c + 1]})) * f32x2{cm[2 * c], cm[2 * c + 1]};
#pragma unroll
      for (int q = 0; q < 4; ++q) pr2[q] = pk_fma(act, f32x2{wp[q][2 * c], wp[q][2 * c + 1]}, pr2[q]);
    }
    float pr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pr[q] = group_sum<G>(pr2[q].x + pr2[q].y);
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
