// Plain LDS-tiled fp32-accumulate implicit GEMMs (VALU FMA).  These are the fp32-mode kernels
// (forward parity <= 1e-3 vs the reference needs exact fp32 products) and the cross-check for the
// bf16 MFMA kernels in gemm_mfma.hip.  Same gather / epilogue code as the MFMA path.
#include "gemm_parts.h"

namespace {

constexpr int TS = 64;   // output tile edge
constexpr int TK = 16;   // reduction step
constexpr int LDT = TS + 4;

// D[m][n] = epi(sum_k A(m,k) * Bw[n][k]);  A gathered through the conv geometry.
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_simple_kernel(const T* __restrict__ x, vkas_conv_geom g,
                                                             const T* __restrict__ Bw, int Np, long M, int K,
                                                             vkas_epilogue e) {
  __shared__ __attribute__((aligned(16))) float As[TK][LDT];
  __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
  const int tid = threadIdx.x;
  const long m0 = (long)blockIdx.x * TS;
  const int n0 = blockIdx.y * TS;
  const int lr = tid >> 2;        // tile row this thread stages (A row / B row)
  const int kq = (tid & 3) * 4;   // 4 consecutive k it stages
  const int ty = tid >> 4, tx = tid & 15;

  const RowCoord rc = decode_row(m0 + lr, M, g);
  const int nb = n0 + lr;
  const bool nb_ok = nb < Np;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += TK) {
    const int k = k0 + kq;
    float av[4] = {0.f, 0.f, 0.f, 0.f};
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      const int tap = k / g.Cp;
      const int c = k - tap * g.Cp;
      const int ky = tap / g.KW;
      const int kx = tap - ky * g.KW;
      const long off = tap_offset(rc, ky, kx, g);
      if (off >= 0) load4(x + off + c, av);
      if (nb_ok) load4(Bw + (long)nb * K + k, bv);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[kq + i][lr] = av[i];
      Bs[kq + i][lr] = bv[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
      const float aa[4] = {a.x, a.y, a.z, a.w};
      const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }
  const int n = n0 + tx * 4;
  if (n < Np) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + ty * 4 + i;
      if (m < M) epi_store4<T>(e, m, n, acc[i]);
    }
  }
}

// gw[n][k] += sum_{m in split} dy[m][n] * A(m,k)
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_simple_kernel(const T* __restrict__ x, vkas_conv_geom g,
                                                             const T* __restrict__ dy, long lddy, int Np, long M,
                                                             int K, long rows_per_split, float* __restrict__ gw,
                                                             float* __restrict__ gb, int x_gelu) {
  __shared__ __attribute__((aligned(16))) float Ds[TK][LDT];  // [m][n]
  __shared__ __attribute__((aligned(16))) float Xs[TK][LDT];  // [m][k]
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * TS;
  const int kb = blockIdx.y * TS;
  const long mbeg = (long)blockIdx.z * rows_per_split;
  const long mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  const int mi = tid >> 4;         // staged row within the chunk
  const int q = (tid & 15) * 4;    // 4 consecutive columns staged
  const int ty = tid >> 4, tx = tid & 15;

  const int k = kb + q;
  int ky = 0, kx = 0, c = 0;
  const bool k_ok = k < K;
  if (k_ok) {
    const int tap = k / g.Cp;
    c = k - tap * g.Cp;
    ky = tap / g.KW;
    kx = tap - ky * g.KW;
  }
  const int nn = n0 + q;
  const bool n_ok = nn < Np;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  const bool do_bias = gb != nullptr && blockIdx.y == 0 && tx == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (long mc = mbeg; mc < mend; mc += TK) {
    const long m = mc + mi;
    float dv[4] = {0.f, 0.f, 0.f, 0.f};
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < mend) {
      if (n_ok) load4(dy + m * lddy + nn, dv);
      if (k_ok) {
        const RowCoord rc = decode_row(m, M, g);
        const long off = tap_offset(rc, ky, kx, g);
        if (off >= 0) load4(x + off + c, xv);
        if (x_gelu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] = gelu_t<T>(xv[i]);
        }
      }
    }
    *reinterpret_cast<float4*>(&Ds[mi][q]) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    *reinterpret_cast<float4*>(&Xs[mi][q]) = make_float4(xv[0], xv[1], xv[2], xv[3]);
    __syncthreads();
#pragma unroll
    for (int mm = 0; mm < TK; ++mm) {
      const float4 a = *reinterpret_cast<const float4*>(&Ds[mm][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Xs[mm][tx * 4]);
      const float aa[4] = {a.x, a.y, a.z, a.w};
      const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[i] += aa[i];
      }
    }
    __syncthreads();
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n0 + ty * 4 + i < Np) atomicAdd(gb + n0 + ty * 4 + i, bsum[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty * 4 + i;
    if (n >= Np) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = kb + tx * 4 + j;
      if (kk < K) atomicAdd(gw + (long)n * K + kk, acc[i][j]);
    }
  }
}

}  // namespace

int vkas_gemm_nt_simple(const void* x, const vkas_conv_geom* g, const void* Bw, int Np, const vkas_epilogue* e,
                        int dtype, hipStream_t st) {
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (M == 0) return VKAS_OK;
  dim3 grid((unsigned)vkas_cdiv(M, TS), (unsigned)vkas_cdiv(Np, TS));
  VKAS_DISPATCH_DTYPE(dtype, "gemm_nt_simple", {
    gemm_nt_simple_kernel<T><<<grid, 256, 0, st>>>((const T*)x, *g, (const T*)Bw, Np, M, K, *e);
  })
  VKAS_LAUNCH_CHECK("gemm_nt_simple");
  return VKAS_OK;
}

int vkas_gemm_tn_simple(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, float* gw,
                        float* gb, int flags, int dtype, hipStream_t st) {
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (M == 0) return VKAS_OK;
  const int x_gelu = flags & 1;  // flags: bit 0 = gelu on load, bit 1 = one split (see vkas_conv_gemm_wgrad_ordered)
  const long tiles = vkas_cdiv(Np, TS) * vkas_cdiv(K, TS);
  long splits = vkas_cdiv(2048, tiles);
  const long max_splits = (flags & 2) ? 1 : vkas_cdiv(M, 4 * TK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  long rows = vkas_cdiv(M, splits);
  rows = vkas_cdiv(rows, TK) * TK;
  splits = vkas_cdiv(M, rows);
  dim3 grid((unsigned)vkas_cdiv(Np, TS), (unsigned)vkas_cdiv(K, TS), (unsigned)splits);
  VKAS_DISPATCH_DTYPE(dtype, "gemm_tn_simple", {
    gemm_tn_simple_kernel<T><<<grid, 256, 0, st>>>((const T*)x, *g, (const T*)dy, lddy, Np, M, K, rows, gw, gb, x_gelu);
  })
  VKAS_LAUNCH_CHECK("gemm_tn_simple");
  return VKAS_OK;
}
