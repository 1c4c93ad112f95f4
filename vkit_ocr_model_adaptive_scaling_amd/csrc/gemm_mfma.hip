// bf16 implicit-GEMM kernels on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16).
//
//  * gemm_nt_mfma: D[m][n] = epi(sum_k A(m,k) Bw[n][k]).  128x128 tile, BK = 64, 4 waves (2x2), each wave a
//    64x64 sub-tile = 4x4 MFMA tiles.  A (gathered through the conv geometry, zero padded) and Bw are
//    staged global -> registers -> LDS (16-byte chunks, XOR-swizzled so that ds_read_b128 fragment reads
//    are bank-conflict free), double buffered, one barrier per K tile.  The MFMA is issued with the
//    operands swapped (D^T = W X^T) so that each lane ends up with 4 consecutive output channels of one
//    pixel: the epilogue works on the same (row, 4 columns) unit as the fp32 kernels.
//  * gemm_tn_mfma (wgrad): gw[n][k] += sum_m dy[m][n] A(m,k).  Both operands are reduced along the slow
//    (pixel) axis, so the LDS tiles are kept [m][col] and the fragments are fetched with the gfx950
//    transposing LDS read (ds_read_b64_tr_b16).  Split over M, fp32 atomics into the (small) gw.
#include "gemm_parts.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT_LDS_TILE = BM * BK;  // elements per operand per buffer

__device__ __forceinline__ int swz_off(int row, int chunk) {  // element offset inside a [128][64] bf16 tile
  return row * BK + ((chunk ^ (row & 7)) << 3);
}

__global__ __launch_bounds__(256) void gemm_nt_mfma_kernel(const bf16_t* __restrict__ x, vkas_conv_geom g,
                                                           const bf16_t* __restrict__ Bw, int Np, long M, int K,
                                                           vkas_epilogue e) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * NT_LDS_TILE];  // [buf][A|B][128][64] = 64 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // staging role: chunk column cc (8 elements), rows sr + 32*i
  const int cc = tid & 7;
  const int sr = tid >> 3;
  int a_by[4], a_y[4], a_x[4];  // b*Hin, oy*stride-pad, ox*stride-pad ; a_by < 0 => row out of range
  const bf16_t* b_ptr[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const RowCoord rc = decode_row(m0 + sr + 32 * i, M, g);
    a_by[i] = rc.ok ? rc.b * g.Hin : -1;
    a_y[i] = rc.oy * g.stride - g.pad;
    a_x[i] = rc.ox * g.stride - g.pad;
    const int n = n0 + sr + 32 * i;
    b_ok[i] = n < Np;
    b_ptr[i] = Bw + (long)(b_ok[i] ? n : 0) * K;
  }
  // running decode of this thread's k chunk: k = kt*BK + cc*8 -> (ky, kx, c)
  int kcur = cc * 8;
  int c_in = kcur, ky = 0, kx = 0;
  while (c_in >= g.Cp) {
    c_in -= g.Cp;
    if (++kx == g.KW) { kx = 0; ++ky; }
  }

  bf16x8 ra[4], rb[4];
  auto load_tile = [&]() {
    const bool k_ok = kcur < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 va = {0, 0, 0, 0, 0, 0, 0, 0};
      bf16x8 vb = {0, 0, 0, 0, 0, 0, 0, 0};
      const int iy = a_y[i] + ky, ix = a_x[i] + kx;
      if (k_ok && a_by[i] >= 0 && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win) {
        const long off = ((long)(a_by[i] + iy) * g.Win + ix) * (long)g.ldx + c_in;
        va = *reinterpret_cast<const bf16x8*>(x + off);
      }
      if (k_ok && b_ok[i]) vb = *reinterpret_cast<const bf16x8*>(b_ptr[i] + kcur);
      ra[i] = va;
      rb[i] = vb;
    }
    // advance to the next K tile
    kcur += BK;
    c_in += BK;
    while (c_in >= g.Cp) {
      c_in -= g.Cp;
      if (++kx == g.KW) { kx = 0; ++ky; }
    }
  };
  auto store_tile = [&](int buf) {
    bf16_t* As = lds + buf * 2 * NT_LDS_TILE;
    bf16_t* Bs = As + NT_LDS_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = sr + 32 * i;
      *reinterpret_cast<bf16x8*>(As + swz_off(row, cc)) = ra[i];
      *reinterpret_cast<bf16x8*>(Bs + swz_off(row, cc)) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  load_tile();
  store_tile(0);
  __syncthreads();

  const int frow = lane & 15;
  const int fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile();
    const bf16_t* As = lds + buf * 2 * NT_LDS_TILE;
    const bf16_t* Bs = As + NT_LDS_TILE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + i * 16 + frow;
        const int rb_ = wn * 64 + i * 16 + frow;
        fa[i] = *reinterpret_cast<const bf16x8*>(As + swz_off(ra_, s * 4 + fchunk));
        fb[i] = *reinterpret_cast<const bf16x8*>(Bs + swz_off(rb_, s * 4 + fchunk));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds D^T rows n = (lane>>4)*4 + r, col m = lane&15 of each 16x16 tile
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= Np) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epi_store4<bf16_t>(e, m, n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
constexpr int TN_ROWS = 64;         // reduction rows per iteration
constexpr int TN_LD = 128 + 16;     // padded row (elements): 288 B, keeps the transposing reads conflict free
constexpr int TN_TILE = TN_ROWS * TN_LD;

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int mbase, int colbase, int lane) {
  // k-permuted fragment: elements 0..3 <- rows mbase + 4g + {0..3}, elements 4..7 <- rows mbase + 16 + 4g + {0..3}
  const int g4 = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
  const bf16_t* a0 = tile + (mbase + 4 * g4 + q) * TN_LD + colbase + 4 * p;
  const bf16_t* a1 = a0 + 16 * TN_LD;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

__global__ __launch_bounds__(256) void gemm_tn_mfma_kernel(const bf16_t* __restrict__ x, vkas_conv_geom g,
                                                           const bf16_t* __restrict__ dy, long lddy, int Np, long M,
                                                           int K, long rows_per_split, float* __restrict__ gw) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * TN_TILE];  // [buf][dy|x][64][144] = 72 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.x * 128;
  const int kb = blockIdx.y * 128;
  const long mbeg = (long)blockIdx.z * rows_per_split;
  const long mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;

  // staging role: chunk column cc (0..15), rows sr + 16*i (i = 0..3)
  const int cc = tid & 15;
  const int sr = tid >> 4;
  const int nn = n0 + cc * 8;
  const bool n_ok = nn < Np;
  const int k = kb + cc * 8;
  const bool k_ok = k < K;
  int ky = 0, kx = 0, c_in = 0;
  if (k_ok) {
    const int tap = k / g.Cp;
    c_in = k - tap * g.Cp;
    ky = tap / g.KW;
    kx = tap - ky * g.KW;
  }

  // running (b, oy, ox) of the 4 rows this thread stages; advanced by TN_ROWS per iteration without divisions
  int r_b[4], r_y[4], r_x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const RowCoord rc = decode_row(mbeg + sr + 16 * i, M, g);
    r_b[i] = rc.b;
    r_y[i] = rc.oy;
    r_x[i] = rc.ox;
  }
  const int kyo = ky - g.pad, kxo = kx - g.pad;

  bf16x8 rd[4], rx[4];
  auto load_tile = [&](long mc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = mc + sr + 16 * i;
      bf16x8 vd = {0, 0, 0, 0, 0, 0, 0, 0};
      bf16x8 vx = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < mend) {
        if (n_ok) vd = *reinterpret_cast<const bf16x8*>(dy + m * lddy + nn);
        const int iy = r_y[i] * g.stride + kyo, ix = r_x[i] * g.stride + kxo;
        if (k_ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win)
          vx = *reinterpret_cast<const bf16x8*>(x + (((long)r_b[i] * g.Hin + iy) * g.Win + ix) * (long)g.ldx + c_in);
      }
      rd[i] = vd;
      rx[i] = vx;
      // advance this row by TN_ROWS output pixels
      r_x[i] += TN_ROWS;
      while (r_x[i] >= g.Wout) {
        r_x[i] -= g.Wout;
        if (++r_y[i] == g.Hout) {
          r_y[i] = 0;
          ++r_b[i];
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    bf16_t* Ds = lds + buf * 2 * TN_TILE;
    bf16_t* Xs = Ds + TN_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = sr + 16 * i;
      *reinterpret_cast<bf16x8*>(Ds + row * TN_LD + cc * 8) = rd[i];
      *reinterpret_cast<bf16x8*>(Xs + row * TN_LD + cc * 8) = rx[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nrows = mend - mbeg;
  const int nit = (int)((nrows + TN_ROWS - 1) / TN_ROWS);
  if (nit > 0) {
    load_tile(mbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) load_tile(mbeg + (long)(it + 1) * TN_ROWS);
    const bf16_t* Ds = lds + buf * 2 * TN_TILE;
    const bf16_t* Xs = Ds + TN_TILE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fd[4], fx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fd[i] = tr_frag(Ds, s * 32, wn * 64 + i * 16, lane);
        fx[i] = tr_frag(Xs, s * 32, wk * 64 + i * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], fx[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < nit) store_tile(buf ^ 1);
    __syncthreads();
  }
  // D[row = n_local][col = k_local]: lane holds col = lane&15, rows (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = kb + wk * 64 + j * 16 + (lane & 15);
      if (kk >= K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (n < Np) atomicAdd(gw + (long)n * K + kk, acc[i][j][r]);
      }
    }
  }
}

}  // namespace

int vkas_gemm_nt_mfma_bf16(const void* x, const vkas_conv_geom* g, const void* Bw, int Np, const vkas_epilogue* e,
                           hipStream_t st) {
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (M == 0) return VKAS_OK;
  dim3 grid((unsigned)vkas_cdiv(M, BM), (unsigned)vkas_cdiv(Np, BN));
  gemm_nt_mfma_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, *g, (const bf16_t*)Bw, Np, M, K, *e);
  VKAS_LAUNCH_CHECK("gemm_nt_mfma");
  return VKAS_OK;
}

int vkas_gemm_tn_mfma_bf16(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, float* gw,
                           hipStream_t st) {
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (M == 0) return VKAS_OK;
  const long tiles = vkas_cdiv(Np, 128) * vkas_cdiv(K, 128);
  long splits = vkas_cdiv(1024, tiles);
  const long max_splits = vkas_cdiv(M, 8 * TN_ROWS);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  long rows = vkas_cdiv(M, splits);
  rows = vkas_cdiv(rows, TN_ROWS) * TN_ROWS;
  splits = vkas_cdiv(M, rows);
  dim3 grid((unsigned)vkas_cdiv(Np, 128), (unsigned)vkas_cdiv(K, 128), (unsigned)splits);
  gemm_tn_mfma_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, *g, (const bf16_t*)dy, lddy, Np, M, K, rows, gw);
  VKAS_LAUNCH_CHECK("gemm_tn_mfma");
  return VKAS_OK;
}
