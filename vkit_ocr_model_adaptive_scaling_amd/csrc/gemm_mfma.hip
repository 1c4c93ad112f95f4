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
#include <stdlib.h>

#include <type_traits>

#include "gemm_parts.h"

// This file is compiled twice (csrc/Makefile): for bf16 storage and, with -DVKAS_MFMA_F16, for fp16 storage (config #5 of
// BASELINE.json).  The two differ in the element type of the operands and in the MFMA opcode (same rate, fp32 accumulate);
// data movement (LDS-DMA, swizzles, transposing reads) only sees 16-bit elements.
#ifdef VKAS_MFMA_F16
typedef f16_t elem_t;
typedef f16x8 elem8;
typedef f16x4 elem4;
#define VKAS_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define VKAS_MFMA_FN(name) name##_f16
#else
typedef bf16_t elem_t;
typedef bf16x8 elem8;
typedef bf16x4 elem4;
#define VKAS_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define VKAS_MFMA_FN(name) name##_bf16
#endif

// tile / kernel choices shared by both builds (defined once, in the bf16 build)
int vkas_gemm_nt_tile_choice(long M, int Np);
int vkas_gemm_tn_tile_choice(long M, int Np, int K);
bool vkas_nt_slab_eligible(const vkas_conv_geom* g, int Np);
int vkas_gemm_nt_ring_stages(const vkas_conv_geom* g, int Np);
bool vkas_tn_slab_eligible(const vkas_conv_geom* g, int Np, long lddy);
bool vkas_tn_slab_n112(int Np);
bool vkas_tn_slab_n96(int Np);

namespace {

constexpr int BK = 64;

// Timing-only ablation switches for profiles/ablate_nt.py (never defined in the shipped build; results are wrong when
// set): 1 no global loads in the K loop, 2 no LDS staging writes, 4 no barrier, 8 fragments read once, 16 no MFMA,
// 256 no fused head tail (the staging passes and their barriers stay); slab kernel only: 32 every LDS-DMA request out of range
// (zero fill, no memory access), 64 every request reads the same 1 KB, 128 weight requests confined to a 1 MB window.
#ifndef VKAS_ABL
#define VKAS_ABL 0
#endif
constexpr int ABL = VKAS_ABL;
// Switches of conv3x3_slab_mfma_kernel (all give correct results; A/B builds through profiles/build_variant.sh, measured with
// profiles/bench_slab.py + profiles/pmc_fetch.sh in round 3 on the fused precise heads, 10.4 ms / 8.4 GB fetched per launch):
//   1  the slab (activation) requests carry the non-temporal hint                       10.6 ms, 13.5 GB with 4 (rows re-fetched)
//   2  the weight requests carry it                                                      10.7 ms
//   4  fused launch of four heads: XCDs 0-3 run heads 0, 1 and XCDs 4-7 heads 2, 3, so that an XCD's L2 (4 MB) holds the
//      2.7 MB of weights it cycles through instead of thrashing on 5.3 MB                10.4 ms, 7.05 GB   <- shipped
//   8  the LDS-DMA requests of a READ phase in front of its fragment reads               10.5 ms
//  32  z leaves with streaming (non-temporal) stores                                     10.5 ms, 8.4 GB (7.03 GB with 4)
#ifndef VKAS_TN_DEEP
#define VKAS_TN_DEEP 1
#endif
#ifndef VKAS_EXP
#define VKAS_EXP 4
#endif
constexpr int EXPS = VKAS_EXP;
// Round-3 schedule experiments on conv3x3_slab_mfma_kernel / conv3x3_wgrad_slab_kernel (both correct, both SLOWER; kept as a
// patch with their measurements in profiles/experiments/): a single instruction stream per wave with fragments prefetched one
// pipeline unit ahead and one barrier per sub-step (4.26 against 3.88 ms on the N = 384, K = 9 x 384 shape), and the LDS-DMA
// requests paced through the MFMA phases instead of issued back to back in the READ phases (4.80 against 3.95 ms).

// Phase timestamps of gemm_nt_mfma_kernel for profiles/trace_nt.py (-DVKAS_TRACE builds only, never shipped): 8 slots per
// workgroup - s_memtime at entry, after the prologue barrier, after the K loop, after the epilogue (stores issued), after
// the stores were acknowledged; slot 6 = HW_ID, slot 7 = XCC_ID.
#if defined(VKAS_TRACE) && !defined(VKAS_MFMA_F16)
__device__ unsigned long long vkas_trace_buf[65536 * 8];
#define VKAS_TR(slot)                                                                                      \
  if (threadIdx.x == 0 && blockIdx.x < 65536) vkas_trace_buf[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter()
#else
#define VKAS_TR(slot)
#endif

__device__ __forceinline__ int swz_off(int row, int chunk) {  // element offset inside a [rows][64] bf16 tile
  return row * BK + ((chunk ^ (row & 7)) << 3);
}

// Fused head tail on one staged pass of ER rows (fp32 accumulators in LDS, row pitch EP): per pixel
//   z = acc + bias -> stored (bf16, needed by backward); LayerNorm statistics over the head's C channels;
//   a = GELU(LN(z)); proj[q] = <a, Wproj[q]> + bproj[q], q < 4.
// LPR = NTHR / ER lanes share a row (8 for the 8-wave tiles): each keeps up to VPL 8-channel vectors in registers,
// row sums go through DPP shuffles inside the lane group.  The (M, C) activation is never written.
template <int NTHR, int ER, int EP, int VPR>
__device__ __forceinline__ void head_tail_rows(const vkas_epilogue& e, int head, const float* stage, const float* hp,
                                               long mrow0, long M, int n0, int width, int tid) {
  constexpr int LPR = NTHR / ER;
  constexpr int VPL = (VPR + LPR - 1) / LPR;
  static_assert(LPR == 4 || LPR == 8, "4 or 8 lanes per row");
  const int row = tid / LPR, j = tid % LPR;
  const long m = mrow0 + row;
  const int C = e.head.c[head];
  const int pw = e.head.pw;
  const int oc = e.head.oc[head];  // projection rows beyond it are zero: not computed (the projection is 8 of a tail's ~27
                                   // vector operations per element at four rows)
  // hp: this head's gamma | beta | Wproj[4] | bproj(8), copied to LDS once per tile by the caller - every lane re-reads its
  // 8-channel slices of them for every row (up to 12 16-byte loads per vector: from global memory that was 1.5 MB through
  // the L1 per tile and a quarter of the fused kernels' time)
  // The accumulators already hold conv + bias (the K loop starts from the bias), and a pad column (beyond the head's C
  // channels) is exactly zero - zero weight rows, zero bias - so sum and sum of squares need no masks; the variance comes from
  // E[z^2] - mean^2 in fp32 (this epilogue exists in the 16-bit storage modes only: the cancellation error, eps * mean^2 / var,
  // stays orders below the storage rounding).  3 of the tail's ~21 vector operations per element less.
  float v[VPL][8];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c0 = (j + LPR * i) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[i][c] = 0.f;
    if (c0 < width) {
      const float4 lo = *reinterpret_cast<const float4*>(stage + row * EP + c0);
      const float4 hi = *reinterpret_cast<const float4*>(stage + row * EP + c0 + 4);
      v[i][0] = lo.x; v[i][1] = lo.y; v[i][2] = lo.z; v[i][3] = lo.w;
      v[i][4] = hi.x; v[i][5] = hi.y; v[i][6] = hi.z; v[i][7] = hi.w;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      s += v[i][c];
      q = fmaf(v[i][c], v[i][c], q);
    }
  }
  s = group_sum<LPR>(s);
  q = group_sum<LPR>(q);
  const float mean = s / (float)C;
  q = fmaxf(q - s * mean, 0.f);  // sum of squared deviations
  const float rstd = rsqrtf(q / (float)C + 1e-6f);
  float pr[4] = {0.f, 0.f, 0.f, 0.f};
  elem_t* zout = reinterpret_cast<elem_t*>(e.out);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c0 = (j + LPR * i) * 8;
    if (c0 >= width) continue;
    if (m < M && zout) {  // zout == nullptr: inference, nothing kept for backward
      if constexpr ((EXPS & 32) != 0) {  // experiment: streaming store (z is next read by the backward pass, far away)
        elem8 a;
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = (elem_t)v[i][c];
        __builtin_nontemporal_store(a, reinterpret_cast<elem8*>(zout + m * e.ldo + n0 + c0));
      } else {
        store8(zout + m * e.ldo + n0 + c0, v[i]);
      }
    }
    float gm[8], bt[8], a[8];
    load8(hp + c0, gm);
    load8(hp + pw + c0, bt);
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = gelu_t<elem_t>((v[i][c] - mean) * rstd * gm[c] + bt[c]);  // pad: gamma = beta = 0
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if (qq >= oc) break;  // workgroup-uniform: the head's out_channels (1 for the rough heads and the probability head)
      float w[8];
      load8(hp + (2 + qq) * pw + c0, w);
#pragma unroll
      for (int c = 0; c < 8; ++c) pr[qq] = fmaf(a[c], w[c], pr[qq]);
    }
  }
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    if (qq >= oc) break;
    pr[qq] = group_sum<LPR>(pr[qq]);
  }
  if (j == 0 && m < M) {
    const float4 bp = *reinterpret_cast<const float4*>(hp + 6 * pw);
    float* po = e.head.proj + ((long)head * M + m) * 8;
    *reinterpret_cast<float4*>(po) = make_float4(pr[0] + bp.x, pr[1] + bp.y, pr[2] + bp.z, pr[3] + bp.w);
    *reinterpret_cast<float4*>(po + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.head.stats) {
      float* st = e.head.stats + ((long)head * M + m) * 2;
      st[0] = mean;
      st[1] = rstd;
    }
  }
}

// Epilogue shared by the NT kernels.  `stage` is the kernel's (idle) tile storage; the caller guarantees that every wave
// has finished reading it.  The accumulators (lane = 4 consecutive channels of one pixel) go through LDS so that every
// global access of the fused epilogue is a coalesced 16-byte piece of an output row.
//  * plain modes: ONE pass for the whole tile.  Every wave adds the bias and writes bf16(acc + bias) - exactly the value
//    the layer's output tensor holds, and what the reference's bf16 autocast feeds to GELU / the residual / the GELU
//    backward - so the whole 256 x BN tile fits the tile buffers, all waves stage at once and there is one barrier
//    instead of eight (the 4-pass fp32 staging was ~40% of a tile's time on the short-K, HBM-bound layer GEMMs).
//  * fused head tail: fp32 staging, one wave-row (TM*16 rows) per pass: LayerNorm statistics want the fp32 sums.
template <int WM, int WN, int TM, int TN, bool HEAD>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[TM][TN], float* stage, const vkas_epilogue& e, int tile_n,
                                            long m0, long M, int n0, int n_end, int tid) {
  constexpr int NTHR = WM * WN * 64;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int VPR = BN / 8;           // 8-channel vectors per row
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  if constexpr (!HEAD) {
    constexpr int PB = BN * 2 + 16;     // staged row pitch in bytes (16-byte aligned rows, 2-way at worst on the b64 writes)
    static_assert(BM * PB <= 2 * (BM + BN) * BK * 2, "single-pass staging must fit the tile buffers");
    char* st = reinterpret_cast<char*>(stage);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cl = wn * TN * 16 + j * 16 + (lane >> 4) * 4;   // column inside the tile
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e.bias && n0 + cl < n_end) b = *reinterpret_cast<const float4*>(e.bias + n0 + cl);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * TM * 16 + i * 16 + (lane & 15);
        float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
        store4(reinterpret_cast<elem_t*>(st + row * PB) + cl, v);
      }
    }
    // epilogues that read a second operand (the residual, the pre-activation of the GELU backward): all of this lane's row
    // pieces are requested here - the accumulators are dead, their registers free - so the requests are in flight together
    // behind the barrier instead of costing one global-load latency per pair of pieces
    constexpr int NIT = (BM * VPR + NTHR - 1) / NTHR;
    const bool pre = e.aux != nullptr && (e.mode == VKAS_EPI_SCALE_RES || e.mode == VKAS_EPI_DGELU || e.mode == VKAS_EPI_ADD);
    Raw8<elem_t> auxr[NIT];
    if (pre) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int t = tid + it * NTHR;
        const int row = t / VPR, c8 = t - row * VPR;
        const long m = m0 + row;
        const int n = n0 + c8 * 8;
        auxr[it].zero();
        if (t < BM * VPR && m < M && n < n_end) auxr[it].load(reinterpret_cast<const elem_t*>(e.aux) + m * e.ldaux + n);
      }
    }
    __syncthreads();
    vkas_epilogue e2 = e;
    e2.bias = nullptr;  // already added
    if (pre) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int t = tid + it * NTHR;
        const int row = t / VPR, c8 = t - row * VPR;
        const long m = m0 + row;
        const int n = n0 + c8 * 8;
        if (t < BM * VPR && m < M && n < n_end) {
          float v[8];
          load8(reinterpret_cast<const elem_t*>(st + row * PB) + c8 * 8, v);
          epi_store8<elem_t>(e2, m, n, v, &auxr[it]);
        }
      }
    } else {
#pragma unroll 2
      for (int t = tid; t < BM * VPR; t += NTHR) {
        const int row = t / VPR, c8 = t - row * VPR;
        const long m = m0 + row;
        const int n = n0 + c8 * 8;
        if (m < M && n < n_end) {
          float v[8];
          load8(reinterpret_cast<const elem_t*>(st + row * PB) + c8 * 8, v);
          epi_store8<elem_t>(e2, m, n, v);
        }
      }
    }
  } else {
    constexpr int EP = BN + 4;            // fp32 row pitch: pitch % 32 == 4 keeps the 16-byte writes conflict free
    constexpr int ER = TM * 16;           // rows per pass
    // the head's parameter block and its slice of the conv bias behind the staged rows (visible after the first barrier)
    float* hp_s = stage + ER * EP;
    {
      const int pw = e.head.pw, PSZ = 6 * pw + 8;
      const float* src = e.head.params + (long)tile_n * PSZ;
      for (int i = tid; i < PSZ; i += NTHR) hp_s[i] = src[i];  // (the conv bias is already in the accumulators)
    }
#pragma unroll 1
    for (int pass = 0; pass < WM; ++pass) {
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4*>(stage + (i * 16 + (lane & 15)) * EP + wn * TN * 16 + j * 16 + (lane >> 4) * 4) =
                acc[i][j];
      }
      __syncthreads();
      if constexpr ((ABL & 256) == 0)  // timing-only ablation: skip the head tail's arithmetic and stores
        head_tail_rows<NTHR, ER, EP, VPR>(e, tile_n, stage, hp_s, m0 + (long)pass * ER, M, n0, n_end - n0, tid);
      __syncthreads();
    }
  }
}

// WM x WN waves, each owning TM x TN 16x16 MFMA tiles: block tile = (WM*TM*16) x (WN*TN*16), BK = 64.
// Instantiated as 128x128 (4 waves), 256x128, 256x192 and 256x224 (8 waves, 2 per SIMD): the wide tiles cut
// the zero-padding waste on N = 192..200 (head convs) and raise the MFMA : LDS-traffic ratio.
//
// BUF = true (every operand < 4 GiB, the normal case): operands are fetched with raw buffer loads.  The hardware
// range check returns zeros for an out-of-range offset, so zero padding / tile tails cost one v_cndmask on a 32-bit
// byte offset instead of a divergent branch, and the per-K-tile address arithmetic shrinks to an add per row: the
// main loop had ~3.8 VALU instructions per MFMA and was issue-bound; this path has < 1.
template <int WM, int WN, int TM, int TN, bool BUF, bool HEAD>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_mfma_kernel(const elem_t* __restrict__ x, vkas_conv_geom g,
                                                                  const elem_t* __restrict__ Bw, int Np, long M, int K,
                                                                  vkas_epilogue e, unsigned a_bytes, unsigned b_bytes) {
  constexpr int NTHR = WM * WN * 64;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int RSTEP = NTHR / 8;                   // rows covered by one staging pass
  constexpr int ACH = BM / RSTEP;                   // A chunks per thread
  constexpr int BCH = (BN + RSTEP - 1) / RSTEP;     // B chunks per thread (last one guarded)
  static_assert(BM % RSTEP == 0, "A tile must be covered by whole staging passes");
  __shared__ __attribute__((aligned(16))) elem_t lds[2 * (BM + BN) * BK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so give every XCD a
  // contiguous run of tiles, N tiles fastest: the N tiles of one pixel block and its vertical neighbours (the 3x3
  // halo rows) then hit the same L2 instead of re-reading A from HBM.  Pure speed choice, any placement is correct.
  constexpr bool head_mode = HEAD;  // N tiles = heads (each <= BN wide, at its own column offset)
  const unsigned ntile_n = head_mode ? (unsigned)e.head.n_heads : (unsigned)((Np + BN - 1) / BN);
  VKAS_TR(0);
  const unsigned total = gridDim.x;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned q8 = total >> 3, r8 = total & 7u;
  const unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const long m0 = (long)(tile / ntile_n) * BM;
  const int tile_n = (int)(tile % ntile_n);
  const int n0 = head_mode ? e.head.n0[tile_n] : tile_n * BN;
  const int n_end = head_mode ? n0 + e.head.np[tile_n] : Np;  // first column this tile must not touch

  // staging role: chunk column cc (8 elements), rows sr + RSTEP*i
  const int cc = tid & 7;
  const int sr = tid >> 3;
  int a_by[ACH], a_y[ACH], a_x[ACH];  // b*Hin, oy*stride-pad, ox*stride-pad ; a_by < 0 => row out of range
  unsigned a_base[ACH];                // BUF: byte offset of pixel (b, a_y, a_x), modulo 2^32 (a_y / a_x may be -pad)
  const elem_t* b_ptr[BCH];
  unsigned b_base[BCH];
  bool b_ok[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const RowCoord rc = decode_row(m0 + sr + RSTEP * i, M, g);
    a_by[i] = rc.ok ? rc.b * g.Hin : -1;
    a_y[i] = rc.oy * g.stride - g.pad;
    a_x[i] = rc.ox * g.stride - g.pad;
    a_base[i] = (((unsigned)(rc.b * g.Hin + a_y[i]) * (unsigned)g.Win + (unsigned)a_x[i]) * (unsigned)g.ldx) << 1;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int r = sr + RSTEP * i;
    const int n = n0 + r;
    b_ok[i] = r < BN && n < n_end;
    b_ptr[i] = Bw + (long)(b_ok[i] ? n : 0) * K;
    b_base[i] = ((unsigned)(b_ok[i] ? n : 0) * (unsigned)K) << 1;
  }
  // running decode of this thread's k chunk: k = kt*BK + cc*8 -> (ky, kx, c)
  int kcur = cc * 8;
  int c_in = kcur, ky = 0, kx = 0;
  while (c_in >= g.Cp) {
    c_in -= g.Cp;
    if (++kx == g.KW) { kx = 0; ++ky; }
  }
  __amdgpu_buffer_rsrc_t rs_a, rs_b;
  if constexpr (BUF) {
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, (int)a_bytes, 0x00020000);
    rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bw, (short)0, (int)b_bytes, 0x00020000);
  }

  elem8 ra[ACH], rb[BCH];
  auto load_tile = [&]() {
    const bool k_ok = kcur < K;
    if constexpr (BUF) {
      constexpr unsigned OOB = 0xFFFFFFF0u;  // beyond any descriptor: the load returns zeros
      const unsigned t_off = (((unsigned)(ky * g.Win + kx) * (unsigned)g.ldx) + (unsigned)c_in) << 1;
#pragma unroll
      for (int i = 0; i < ACH; ++i) {
        const int iy = a_y[i] + ky, ix = a_x[i] + kx;
        const bool ok = k_ok && a_by[i] >= 0 && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? a_base[i] + t_off : OOB, 0, 0);
        ra[i] = __builtin_bit_cast(elem8, v);
      }
      const unsigned kb2 = (unsigned)kcur << 1;
#pragma unroll
      for (int i = 0; i < BCH; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (k_ok && b_ok[i]) ? b_base[i] + kb2 : OOB, 0, 0);
        rb[i] = __builtin_bit_cast(elem8, v);
      }
    } else {
#pragma unroll
      for (int i = 0; i < ACH; ++i) {
        elem8 va = {0, 0, 0, 0, 0, 0, 0, 0};
        const int iy = a_y[i] + ky, ix = a_x[i] + kx;
        if (k_ok && a_by[i] >= 0 && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win) {
          const long off = ((long)(a_by[i] + iy) * g.Win + ix) * (long)g.ldx + c_in;
          va = *reinterpret_cast<const elem8*>(x + off);
        }
        ra[i] = va;
      }
#pragma unroll
      for (int i = 0; i < BCH; ++i) {
        elem8 vb = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k_ok && b_ok[i]) vb = *reinterpret_cast<const elem8*>(b_ptr[i] + kcur);
        rb[i] = vb;
      }
    }
    // advance to the next K tile (branch-free when a tap holds at least one K tile of channels)
    kcur += BK;
    c_in += BK;
    if (g.Cp >= BK) {  // wave-uniform
      const bool wrap = c_in >= g.Cp;
      c_in -= wrap ? g.Cp : 0;
      kx += wrap ? 1 : 0;
      const bool wrap_x = kx == g.KW;
      kx = wrap_x ? 0 : kx;
      ky += wrap_x ? 1 : 0;
    } else {
      while (c_in >= g.Cp) {
        c_in -= g.Cp;
        if (++kx == g.KW) { kx = 0; ++ky; }
      }
    }
  };
  auto store_tile = [&](int buf) {
    elem_t* As = lds + buf * (BM + BN) * BK;
    elem_t* Bs = As + BM * BK;
#pragma unroll
    for (int i = 0; i < ACH; ++i) *reinterpret_cast<elem8*>(As + swz_off(sr + RSTEP * i, cc)) = ra[i];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int row = sr + RSTEP * i;
      if (BN % RSTEP == 0 || row < BN) *reinterpret_cast<elem8*>(Bs + swz_off(row, cc)) = rb[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (HEAD) {  // as in conv3x3_slab_body: the fused head tail starts from the bias
      const int col = n0 + wn * TN * 16 + j * 16 + (lane >> 4) * 4;
      if (e.bias && col < n_end) b4 = *reinterpret_cast<const f32x4*>(e.bias + col);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i][j] = b4;
  }

  const int nk = (K + BK - 1) / BK;
  load_tile();
  store_tile(0);
  if (nk > 1) load_tile();  // registers now hold tile 1
  __syncthreads();
  VKAS_TR(1);

  const int frow = lane & 15;
  const int fchunk = lane >> 4;
  // Per K tile: first MFMA half | registers (tile kt+1, loaded one and a half iterations ago) -> LDS, reissue the
  // global loads for tile kt+2 | second MFMA half | barrier.  The LDS write latency and the global-load latency
  // both sit behind MFMA work instead of in front of the barrier.
  elem8 fa_once[2][TM], fb_once[2][TN];
  if constexpr ((ABL & 8) != 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa_once[s][i] = *reinterpret_cast<const elem8*>(lds + swz_off(wm * TM * 16 + i * 16 + frow, s * 4 + fchunk));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb_once[s][j] = *reinterpret_cast<const elem8*>(lds + BM * BK + swz_off(wn * TN * 16 + j * 16 + frow, s * 4 + fchunk));
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const elem_t* As = lds + buf * (BM + BN) * BK;
    const elem_t* Bs = As + BM * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      elem8 fa[TM], fb[TN];
      if constexpr ((ABL & 8) != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = fa_once[s][i];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = fb_once[s][j];
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const elem8*>(As + swz_off(wm * TM * 16 + i * 16 + frow, s * 4 + fchunk));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const elem8*>(Bs + swz_off(wn * TN * 16 + j * 16 + frow, s * 4 + fchunk));
      }
      if constexpr ((ABL & 16) != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
      } else {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = VKAS_MFMA16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      if (s == 0 && kt + 1 < nk) {
        if constexpr ((ABL & 2) == 0) store_tile(buf ^ 1);
        else {
#pragma unroll
          for (int i = 0; i < ACH; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
          for (int i = 0; i < BCH; ++i) asm volatile("" ::"v"(rb[i]));
        }
        if constexpr ((ABL & 1) == 0) {
          if (kt + 2 < nk) load_tile();
        }
      }
    }
    if constexpr ((ABL & 4) == 0) __syncthreads();
  }

  static_assert(TM * 16 * (BN + 4) * 4 + (7 * 224 + 8) * 4 <= 2 * (BM + BN) * BK * 2, "epilogue staging must fit the tile buffers");
  VKAS_TR(2);
  nt_epilogue<WM, WN, TM, TN, HEAD>(acc, reinterpret_cast<float*>(lds), e, tile_n, m0, M, n0, n_end, tid);
  VKAS_TR(3);
#if defined(VKAS_TRACE) && !defined(VKAS_MFMA_F16)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  VKAS_TR(4);
  if (threadIdx.x == 0 && blockIdx.x < 65536) {
    vkas_trace_buf[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    vkas_trace_buf[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// gemm_nt_ring_kernel (round 4): the NT GEMM for launches of FEW tiles - the forward-only configurations of BASELINE.json
// (configs[1] / [4]: one to four pages per call, M = 400 ... 12 288 rows at stages 2 / 3 of the backbone, K up to 4 096).
// There a launch is one round of <= 256 workgroups and its duration is the length of ONE workgroup's K loop; the kernel above
// requests a K tile one and a half iterations ahead through registers, so every iteration costs most of a memory round trip
// (1.1 us per 64-deep step measured: 70 us for M = 1 792, N = 1 024, K = 4 096 - profiles/sweep_small.py).  This kernel keeps
// NST - 1 K tiles in flight instead: operands go global -> LDS by LDS-DMA into a ring of NST stages (no staging registers, no
// ds_write pass), a stage is waited for with a counted vmcnt (the younger stages stay in flight) in front of ONE raw barrier per
// K tile, and the slot read in the previous iteration is refilled right behind that barrier.  128 x 128 tile, 8 waves of 32 x 64
// (two per SIMD, so one wave's requests and fragment reads sit behind the other's MFMAs); same geometry decode, swizzle, MFMA
// operand order and epilogue as gemm_nt_mfma_kernel - the results are bit-identical to it.
// LDS image: a wave instruction fills 8 rows x 128 B, lane-linear; the XOR swizzle the fragment reads expect is applied to the
// per-lane SOURCE chunk (as in the slab kernel).  Out-of-range lanes (zero padding, M / N / K tails, the dummy stages behind the
// last K tile that keep the wait count uniform) use an offset beyond the descriptor: the hardware writes zeros.
template <int NST>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(const elem_t* __restrict__ x, vkas_conv_geom g,
                                                           const elem_t* __restrict__ Bw, int Np, long M, int K,
                                                           vkas_epilogue e, unsigned a_bytes, unsigned b_bytes) {
  constexpr int WM = 4, WN = 2, TM = 2, TN = 4;
  constexpr int NTHR = WM * WN * 64;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int RSTEP = NTHR / 8;                     // rows covered by one staging instruction of every wave
  constexpr int ACH = BM / RSTEP, BCH = BN / RSTEP;   // requests per lane and stage
  constexpr int STAGE = (BM + BN) * BK;               // elements per ring stage
  static_assert(BM % RSTEP == 0 && BN % RSTEP == 0 && RSTEP % 8 == 0, "whole 8-row instructions");
  static_assert(NST >= 2 && NST * STAGE * 2 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) elem_t lds[NST * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned ntile_n = (unsigned)((Np + BN - 1) / BN);
  const unsigned total = gridDim.x;
  const unsigned xcd = blockIdx.x & 7u, slot8 = blockIdx.x >> 3;
  const unsigned q8 = total >> 3, r8 = total & 7u;
  const unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
  const long m0 = (long)(tile / ntile_n) * BM;
  const int tile_n = (int)(tile % ntile_n);
  const int n0 = tile_n * BN;
  const int n_end = Np;

  // staging role: LDS position (tid & 7) of rows sr + RSTEP i; it receives source chunk cc = position ^ (row & 7)
  const int sr = tid >> 3;
  const int cc = (tid & 7) ^ (sr & 7);
  int a_by[ACH], a_y[ACH], a_x[ACH];
  unsigned a_base[ACH], b_base[BCH];
  bool b_ok[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const RowCoord rc = decode_row(m0 + sr + RSTEP * i, M, g);
    a_by[i] = rc.ok ? rc.b * g.Hin : -1;
    a_y[i] = rc.oy * g.stride - g.pad;
    a_x[i] = rc.ox * g.stride - g.pad;
    a_base[i] = (((unsigned)(rc.b * g.Hin + a_y[i]) * (unsigned)g.Win + (unsigned)a_x[i]) * (unsigned)g.ldx) << 1;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int n = n0 + sr + RSTEP * i;
    b_ok[i] = n < n_end;
    b_base[i] = ((unsigned)(b_ok[i] ? n : 0) * (unsigned)K) << 1;
  }
  int kcur = cc * 8;
  int c_in = kcur, ky = 0, kx = 0;
  while (c_in >= g.Cp) {
    c_in -= g.Cp;
    if (++kx == g.KW) { kx = 0; ++ky; }
  }
  const u32x4 rs_a = vkas_make_rsrc(x, a_bytes);
  const u32x4 rs_b = vkas_make_rsrc(Bw, b_bytes);
  const unsigned lds0 = vkas_lds_addr(lds) + (unsigned)wave * (8 * BK * 2);  // this wave's 8 rows of staging pass 0
  constexpr unsigned OOB = 0xFFFFFFF0u;

  auto issue = [&](int slot) __attribute__((always_inline)) {
    const bool k_ok = kcur < K;
    const unsigned t_off = (((unsigned)(ky * g.Win + kx) * (unsigned)g.ldx) + (unsigned)c_in) << 1;
    const unsigned dst = lds0 + (unsigned)slot * (STAGE * 2);
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int iy = a_y[i] + ky, ix = a_x[i] + kx;
      const bool ok = k_ok && a_by[i] >= 0 && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
      vkas_lds_dma16(rs_a, dst + i * (RSTEP * BK * 2), ok ? a_base[i] + t_off : OOB);
    }
    const unsigned kb2 = (unsigned)kcur << 1;
#pragma unroll
    for (int i = 0; i < BCH; ++i)
      vkas_lds_dma16(rs_b, dst + BM * BK * 2 + i * (RSTEP * BK * 2), (k_ok && b_ok[i]) ? b_base[i] + kb2 : OOB);
    kcur += BK;
    c_in += BK;
    if (g.Cp >= BK) {  // wave-uniform
      const bool wrap = c_in >= g.Cp;
      c_in -= wrap ? g.Cp : 0;
      kx += wrap ? 1 : 0;
      const bool wrap_x = kx == g.KW;
      kx = wrap_x ? 0 : kx;
      ky += wrap_x ? 1 : 0;
    } else {
      while (c_in >= g.Cp) {
        c_in -= g.Cp;
        if (++kx == g.KW) { kx = 0; ++ky; }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);

  const int frow = lane & 15;
  const int fchunk = lane >> 4;
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // K tile kt has landed once at most the NST - 2 younger stages of this wave are outstanding; behind the barrier every
    // wave's share of it is visible and every wave has finished reading tile kt - 1, whose slot is refilled at once
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (ACH + BCH)) : "memory");
    __builtin_amdgcn_s_barrier();
    // the two waves of a SIMD (w and w + 4) take turns: one issues its requests while the other one multiplies
    if (wave < 4) issue(slot == 0 ? NST - 1 : slot - 1);
    const elem_t* As = lds + slot * STAGE;
    const elem_t* Bs = As + BM * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      elem8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const elem8*>(As + swz_off(wm * TM * 16 + i * 16 + frow, s * 4 + fchunk));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const elem8*>(Bs + swz_off(wn * TN * 16 + j * 16 + frow, s * 4 + fchunk));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = VKAS_MFMA16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (wave >= 4) issue(slot == 0 ? NST - 1 : slot - 1);
    slot = slot + 1 == NST ? 0 : slot + 1;
  }
  // the dummy stages (zero fill) must have landed and every wave must be done reading before the ring becomes epilogue staging
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  nt_epilogue<WM, WN, TM, TN, false>(acc, reinterpret_cast<float*>(lds), e, tile_n, m0, M, n0, n_end, tid);
}

// ---------------------------------------------------------------------------------------------------
// Row-slab kernel for 3x3 / stride 1 / pad 1 convolutions whose rows are a multiple of 256 pixels wide (the head convs
// and their input gradients at 8x512x512: ~45% of the step).  The generic kernel above is bound by L2 -> CU operand
// traffic there (a 256x192 tile moves 57 KB per 64-deep K step, ~17 TB/s chip-wide, the measured L2 gather rate), so
// this kernel cuts the bytes instead of chasing issue slots:
//  * an M tile is 256 consecutive pixels of ONE image row.  For input row ky and a 64-channel block, the 258-pixel
//    slab (tile + one halo pixel each side) is staged once and serves the three kx taps as row-shifted fragment
//    reads: A traffic / 3.  K is walked as (channel block, ky, kx); the weights keep their [n][ky][kx][c] layout.
//  * operands go global -> LDS directly (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass.  The
//    LDS image of one wave instruction is 8 rows x 128 B, lane-linear; the XOR swizzle the fragment reads expect is
//    applied to the per-lane SOURCE chunk.  Out-of-range lanes (zero padding, channel / N tails) use an offset
//    beyond the descriptor: the hardware writes zeros for them.
//  * 2 slab buffers + a ring of 3 weight tiles; every sub-step issues the weight tile two sub-steps ahead and a
//    share of the next slab, then waits with a counted vmcnt (only this sub-step's own issues stay in flight) in
//    front of a raw s_barrier.  A staged buffer is read only after the barrier that follows the wait retiring it,
//    and re-filled only after the barrier that ends its last reading sub-step.
constexpr int SLAB = 264 * BK;  // rows 0..255 tile pixels, 256 / 257 left / right halo, 258..263 unused

// TN = MFMA column tiles per wave (block N extent 32 * TN); BNT = N extent the launch spaces its N tiles by (>= 32 * TN:
// a fused-head launch sized for its widest head runs the narrower heads with a smaller TN, see the kernel below).
typedef __attribute__((address_space(3))) elem_t lds_elem;
typedef __attribute__((address_space(3))) elem8 lds_elem8;

template <int TN, bool HEAD, int BNT>
__device__ __forceinline__ void conv3x3_slab_body(const elem_t* __restrict__ x, const vkas_conv_geom& g,
                                                  const elem_t* __restrict__ Bw, int Np, long M, int K,
                                                  const vkas_epilogue& e, unsigned a_bytes, unsigned b_bytes,
                                                  lds_elem* lds, unsigned tile, unsigned ntile_n) {
  constexpr int WM = 4, WN = 2, TM = 4;
  constexpr int BM = 256, BN = WN * TN * 16;
  constexpr int BT = BN * BK;
  constexpr int NWI = BN / 8;                // wave instructions (8 rows each) per weight tile
  constexpr int NBQ = (NWI + 7) / 8;         // ... per wave (waves >= RAG issue one less when NWI % 8 != 0)
  constexpr int RAG = NWI % 8;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  static_assert(TM * 16 * (BN + 4) * 4 + (7 * 224 + 8) * 4 <= (2 * SLAB + 3 * BT) * 2, "epilogue staging must fit");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const long m0 = (long)(tile / ntile_n) * BM;
  const int tile_n = (int)(tile % ntile_n);
  const int n0 = HEAD ? e.head.n0[tile_n] : tile_n * BNT;
  const int n_end = HEAD ? n0 + e.head.np[tile_n] : Np;
  const int H = g.Hin, W = g.Win, Cp = g.Cp;
  const int hw = H * W;
  const int bimg = (int)(m0 / hw);
  const int rem = (int)(m0 - (long)bimg * hw);
  const int oy = rem / W, ox0 = rem - oy * W;  // the tile is pixels ox0 .. ox0+255 of image row (bimg, oy)

  // per-lane staging constants: a wave instruction fills 8 rows x 8 chunk positions; the lane at (row lr, position
  // cpos) fetches logical chunk cpos ^ lr (all row bases are multiples of 8)
  const int lr = lane >> 3;
  const int cl = (lane & 7) ^ lr;
  const unsigned row_pitch = (unsigned)W * (unsigned)g.ldx * 2u;  // bytes between image rows
  // one base offset per operand; the per-instruction offsets are scalar multiples added at the point of use (the
  // kernel lives at the VGPR limit: keeping 4 + 4 hoisted copies spills, and a scratch reload drains the DMA queue)
  const unsigned a_base = ((unsigned)((bimg * H + oy - 1) * W + ox0 + wave * 8 + lr) * (unsigned)g.ldx + (unsigned)(cl * 8)) * 2u;
  const unsigned a_qstep = 64u * (unsigned)g.ldx * 2u;  // 64 slab rows further
  // halo: lanes 0, 1 of wave w carry chunks 2w, 2w+1 of the 16 halo chunks (row 256: pixel ox0-1, row 257: ox0+256)
  const int hc = wave * 2 + (lane & 1);
  const int hrow = hc >> 3;
  const int hcl = (hc & 7) ^ hrow;
  const int hpix = hrow == 0 ? ox0 - 1 : ox0 + BM;
  const bool h_ok = (unsigned)hpix < (unsigned)W;
  const unsigned h_off = ((unsigned)((bimg * H + oy - 1) * W + hpix) * (unsigned)g.ldx + (unsigned)(hcl * 8)) * 2u;
  const int b_row = wave * 8 + lr;                       // row of this lane in its first weight instruction
  const int b_rows = (n_end - n0 < BN) ? n_end - n0 : BN;  // valid rows of the weight tile
  const unsigned b_base = ((unsigned)(n0 + b_row) * (unsigned)K + (unsigned)(cl * 8)) * 2u;
  const unsigned b_qstep = 64u * (unsigned)K * 2u;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bw, (short)0, (int)b_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // slab part `part` (0, 1: wave instructions {0,1} / {2,3} of the tile rows; 2: halo) of step (ky, cb) -> slab sb
  auto issue_a = [&](int part, int ky, int cb, int sb) {
    const bool row_ok = (unsigned)(oy - 1 + ky) < (unsigned)H;
    const unsigned step_off = (unsigned)ky * row_pitch + (unsigned)cb * 128u;
    lds_elem* dst = lds + sb * SLAB;
    if (part < 2) {
      const bool ok = row_ok && cb * 64 + cl * 8 < Cp;
      unsigned base = a_base;
      asm volatile("" : "+v"(base));  // opaque: recomputed per use, never hoisted
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = part * 2 + qq;
        unsigned voff = ok ? base + (unsigned)q * a_qstep + step_off : OOB;
        if constexpr ((ABL & 32) != 0) voff = OOB;                      // timing only: zero fill, nothing fetched
        if constexpr ((ABL & 64) != 0) voff = (unsigned)lane * 16u;     // timing only: every request reads the same 1 KB
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(dst + (q * 8 + wave) * 512), 16, voff, 0, 0, (EXPS & 1) ? 2 : 0);
      }
    } else {
      const bool ok = row_ok && h_ok && cb * 64 + hcl * 8 < Cp;
      const unsigned voff = ok ? h_off + step_off : OOB;
      if (lane < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(dst + 256 * BK + wave * 16), 16, voff, 0, 0, 0);
    }
  };
  // weight tile of (ky, kx, cb) -> ring slot
  auto issue_b = [&](int ky, int kx, int cb, int slot) {
    const unsigned step_off = ((unsigned)((ky * 3 + kx) * Cp) + (unsigned)cb * 64u) * 2u;
    const bool c_ok = cb * 64 + cl * 8 < Cp;
    lds_elem* dst = lds + 2 * SLAB + slot * BT;
    unsigned base = b_base;
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int q = 0; q < NBQ; ++q) {
      if (RAG != 0 && q == NBQ - 1 && wave >= RAG) break;  // wave-uniform
      unsigned voff = (c_ok && q * 64 + b_row < b_rows) ? base + (unsigned)q * b_qstep + step_off : OOB;
      if constexpr ((ABL & 32) != 0) voff = OOB;
      if constexpr ((ABL & 64) != 0) voff = (unsigned)lane * 16u;
      if constexpr ((ABL & 128) != 0) voff = (unsigned)(q * 8 + wave) * 1024u + (unsigned)lane * 16u + step_off % (1u << 20);  // weights from a 1 MB window
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(dst + (q * 8 + wave) * 512), 16, voff, 0, 0, (EXPS & 2) ? 2 : 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (HEAD) {  // the fused head tail wants conv + bias: start from the bias instead of adding it per element later
      const int col = n0 + wn * TN * 16 + j * 16 + (lane >> 4) * 4;  // lane = 4 consecutive channels of one pixel
      if (e.bias && col < n_end) b4 = *reinterpret_cast<const f32x4*>(e.bias + col);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i][j] = b4;
  }

  const int NCB = (Cp + 63) >> 6;
  const int NS = 3 * NCB;  // steps (ky, cb); three kx sub-steps each
  issue_a(0, 0, 0, 0);
  issue_a(1, 0, 0, 0);
  issue_a(2, 0, 0, 0);
  issue_b(0, 0, 0, 0);
  issue_b(0, 1, 0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int frow = lane & 15;
  const int fchunk = lane >> 4;
  // Two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run half a sub-step apart: every sub-step is a READ
  // phase (issue the LDS-DMA, fetch both K halves of the fragments) and an MFMA phase (48-56 back-to-back MFMAs), each
  // closed by a barrier; group 1 starts one barrier late, so on every SIMD one wave feeds the matrix core while the
  // other one reads.  Buffer life times with the lag (intervals between barriers, sub-step j: group 0 reads in interval
  // 2j, group 1 in 2j+1): tile j's slot is re-filled from interval 2j+2 on; a share issued in READ(j-1) is retired by
  // group 1 at the end of READ(j) and by group 0 at the end of MFMA(j), both in front of the barrier that ends interval
  // 2j+1, i.e. before the first read of tile j+1 in interval 2j+2.
  const int grp = wave >> 2;
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int ky = 0, cb = 0;
  // one step = (ky, cb) with its three kx sub-steps; the slab parity is a compile-time constant (two steps per loop
  // trip) so that every LDS address is a per-lane base plus an immediate
  auto step = [&](auto sbc, int s) {
    constexpr int sb = decltype(sbc)::value;
    const bool more = s + 1 < NS;
    // K order (channel block, ky, kx): the three input rows of a channel block are staged in consecutive steps, so row y
    // of block cb is fetched by the tiles of rows y+1, y, y-1 within ~3 steps of each other and the later two hit L2.
    // With ky outermost (round 1) those uses were a third of the K loop apart and the row had left the 4 MB L2 in
    // between: rocprofv3 showed 13.3 GB per dgrad launch against ~5 GB of operands.
    int ky1 = ky + 1, cb1 = cb;
    if (ky1 == 3) { ky1 = 0; cb1 = cb + 1; }
    const lds_elem* As = lds + sb * SLAB;
    auto sub = [&](auto kxc) {
      constexpr int kx = decltype(kxc)::value;
      constexpr int NA = kx == 0 ? 3 : (kx == 1 ? 2 : 0);
      auto retire = [&]() {  // all but this sub-step's own issues have landed
        if constexpr ((ABL & 1) != 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (more) {
          if (RAG != 0 && wave >= RAG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NBQ - (RAG != 0 ? 1 : 0)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NBQ) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      };
      // ---- READ phase: share of the next slab, weight tile two sub-steps ahead, this sub-step's fragments
      const lds_elem* Bs = lds + 2 * SLAB + kx * BT;
      elem8 fa[2][TM], fb[2][TN];
      auto issue_all = [&]() {
        if constexpr ((ABL & 1) == 0) {
          if (more) {
            if constexpr (kx == 0) { issue_a(0, ky1, cb1, sb ^ 1); issue_a(2, ky1, cb1, sb ^ 1); }
            if constexpr (kx == 1) issue_a(1, ky1, cb1, sb ^ 1);
          }
          if constexpr (kx == 0) issue_b(ky, 2, cb, 2);
          else if (more) issue_b(ky1, kx - 1, cb1, kx - 1);
        }
      };
      if constexpr ((EXPS & 8) != 0) {  // experiment: requests in front of the fragment reads
        issue_all();
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = wm * 64 + i * 16 + frow;      // tile pixel
          int sr = r + kx - 1;                         // slab row of the tap's pixel
          if (kx == 0 && i == 0) sr = (sr < 0) ? 256 : sr;
          if (kx == 2 && i == TM - 1) sr = (sr > 255) ? 257 : sr;
          fa[h][i] = *(const lds_elem8*)(As + swz_off(sr, h * 4 + fchunk));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[h][j] = *(const lds_elem8*)(Bs + swz_off(wn * TN * 16 + j * 16 + frow, h * 4 + fchunk));
      }
      asm volatile("" ::: "memory");  // fragment reads first: their latency hides behind the DMA issue
      if constexpr ((EXPS & 8) == 0) issue_all();
      if (grp == 1) retire();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr ((ABL & 4) == 0) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // ---- MFMA phase
      if constexpr ((ABL & 16) != 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[h][i]));
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[h][j]));
        }
      } else {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = VKAS_MFMA16(fb[h][j], fa[h][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      if (grp == 0) retire();
      if constexpr ((ABL & 4) == 0) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    sub(std::integral_constant<int, 0>{});
    sub(std::integral_constant<int, 1>{});
    sub(std::integral_constant<int, 2>{});
    ky = ky1;
    cb = cb1;
  };
  int s = 0;
  for (; s + 1 < NS; s += 2) {
    step(std::integral_constant<int, 0>{}, s);
    step(std::integral_constant<int, 1>{}, s + 1);
  }
  if (s < NS) step(std::integral_constant<int, 0>{}, s);
  if (grp == 0) __builtin_amdgcn_s_barrier();  // group 1's last MFMA phase: every wave passes the same number of barriers
  nt_epilogue<WM, WN, TM, TN, HEAD>(acc, (float*)lds, e, tile_n, m0, M, n0, n_end, tid);
}

template <int TN, bool HEAD>
__global__ __launch_bounds__(512) void conv3x3_slab_mfma_kernel(const elem_t* __restrict__ x, vkas_conv_geom g,
                                                                const elem_t* __restrict__ Bw, int Np, long M, int K,
                                                                vkas_epilogue e, unsigned a_bytes, unsigned b_bytes) {
  constexpr int BN = 32 * TN;
  __shared__ __attribute__((aligned(1024))) elem_t lds[2 * SLAB + 3 * BN * BK];
  // tile order: as in gemm_nt_mfma_kernel (contiguous runs per XCD, N tiles fastest)
  const unsigned ntile_n = HEAD ? (unsigned)e.head.n_heads : (unsigned)((Np + BN - 1) / BN);
  const unsigned total = gridDim.x;
  const unsigned xcd = blockIdx.x & 7u, slot8 = blockIdx.x >> 3;
  const unsigned q8 = total >> 3, r8 = total & 7u;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
  if constexpr (HEAD && (EXPS & 4) != 0) {
    if (ntile_n == 4 && (total & 31u) == 0) {  // whole M tiles per XCD quarter
      const unsigned mt = total >> 2;                    // M tiles
      const unsigned m_tile = (xcd & 3u) * (mt >> 2) + (slot8 >> 1);
      tile = m_tile * 4u + (xcd >> 2) * 2u + (slot8 & 1u);
    }
  }
  if constexpr (HEAD && TN == 7) {
    // the launch is sized for its widest head (<= 224 columns); heads of <= 192 columns run the 6-tile body: 1/7 fewer
    // MFMAs and weight bytes for three of the four precise heads
    if (e.head.np[tile % ntile_n] <= 192) {
      conv3x3_slab_body<6, true, BN>(x, g, Bw, Np, M, K, e, a_bytes, b_bytes, (lds_elem*)lds, tile, ntile_n);
      return;
    }
  }
  conv3x3_slab_body<TN, HEAD, BN>(x, g, Bw, Np, M, K, e, a_bytes, b_bytes, (lds_elem*)lds, tile, ntile_n);
}

// ---------------------------------------------------------------------------------------------------
constexpr int TN_ROWS = 64;  // reduction rows per iteration

// acc + lo + hi of a packed pair of 16-bit elements (one v_dot2c_f32_{bf16,f16} against (1, 1); products with 1 are exact)
__device__ __forceinline__ float dot2_ones(unsigned pair, float acc) {
#ifdef VKAS_MFMA_F16
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pair), one, acc, false);
#else
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 one = {(__bf16)1.0f, (__bf16)1.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, pair), one, acc, false);
#endif
}

// k-permuted transposed fragment: elements 0..3 <- rows mbase + 4g + {0..3}, elements 4..7 <- rows mbase + 16 + 4g + {0..3}
// (the same permutation is used for both MFMA operands, so the sum over k is unchanged)
template <int LD>
__device__ __forceinline__ elem8 tr_frag(const elem_t* tile, int mbase, int colbase, int lane) {
  const int g4 = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
  const elem_t* a0 = tile + (mbase + 4 * g4 + q) * LD + colbase + 4 * p;
  const elem_t* a1 = a0 + 16 * LD;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
  union { struct { s16x4 l, h; } s; elem8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

// gw[n][k] += sum_m dy[m][n] * A(m,k).  WN x WK waves, each TNn x TK MFMA tiles: the block owns (WN*TNn*16) output
// channels x (WK*TK*16) K columns and walks its M split 64 rows at a time.  LDS tiles stay [row][col] with rows
// padded by 16 elements (row pitch = odd multiple of 32 B => the 8 rows a half-wave touches in one transposing
// read sit on distinct bank groups).
// XG: the x operand is gelu(x) (the W2 weight gradient of the fused ConvNeXt MLP, whose forward keeps only the
// pre-activation): applied to the staged registers on their way into LDS, once per element and N tile.
// PW (pointwise, BUF only): KH = KW = 1, stride 1, no padding - row m of x starts at m * ldx, so the per-chunk tap / row
// decode (and the registers it keeps: three per staged x row) reduce to one running byte offset per staged row.
// NOBIAS: compiled without the bias-gradient sums (gb must be null): the second staging set only fits beside 96 accumulator
// registers without them.
template <int WN, int WK, int TNn, int TK, bool BUF, bool XG, bool PW = false, bool NOBIAS = false>
__global__ __launch_bounds__(WN* WK * 64) void gemm_tn_mfma_kernel(const elem_t* __restrict__ x, vkas_conv_geom g,
                                                                  const elem_t* __restrict__ dy, long lddy, int Np,
                                                                  long M, int K, long rows_per_split,
                                                                  float* __restrict__ gw, float* __restrict__ gb,
                                                                  unsigned x_bytes, unsigned dy_bytes) {
  constexpr int NTHR = WN * WK * 64;
  constexpr int BNn = WN * TNn * 16, BKc = WK * TK * 16;
  constexpr int LDD = BNn + 16, LDX = BKc + 16;
  static_assert((BNn / 16) % 2 == 0 && (BKc / 16) % 2 == 0, "row pitch must be an odd multiple of 32 bytes");
  constexpr int CPRD = BNn / 8, CPRX = BKc / 8;           // 16-byte chunks per row
  constexpr int DCH = (TN_ROWS * CPRD + NTHR - 1) / NTHR;  // dy chunks per thread (last guarded)
  constexpr int XSTEP = NTHR / CPRX;                       // rows between a thread's x chunks
  constexpr int XCH = TN_ROWS / XSTEP;                     // x chunks per thread
  static_assert(NTHR % CPRX == 0 && TN_ROWS % XSTEP == 0, "x tile must be covered by whole staging passes");
  constexpr int TILE = TN_ROWS * (LDD + LDX);
  __shared__ __attribute__((aligned(16))) elem_t lds[2 * TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;
  // Work order: every XCD (private L2; workgroups are dealt round-robin over the 8 XCDs) gets a contiguous run of
  // work items, all (n, k) tiles of one M split before the next split.  The tiles of a split walk the same dy / x
  // rows at the same time, so each row is fetched from HBM about once per XCD and re-used from L2 by the other tiles
  // (dy by every K tile, x by every N tile and by the K tiles of the other taps).  Speed only, never correctness.
  const unsigned ntn = (unsigned)((Np + BNn - 1) / BNn), ntk = (unsigned)((K + BKc - 1) / BKc);
  const unsigned total = gridDim.x;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned q8 = total >> 3, r8 = total & 7u;
  const unsigned work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const unsigned tile = work % (ntn * ntk);
  const long mbeg = (long)(work / (ntn * ntk)) * rows_per_split;
  const long mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  const bool first_k_tile = tile / ntn == 0;
  const int n0 = (int)(tile % ntn) * BNn;
  const int kb = (int)(tile / ntn) * BKc;

  // dy staging: chunk index tid + NTHR*i -> (row, col)
  int d_row[DCH], d_lds[DCH];
  const elem_t* d_ptr[DCH];
  unsigned d_off[DCH];  // BUF: running byte offset of this chunk
  bool d_ok[DCH];
#pragma unroll
  for (int i = 0; i < DCH; ++i) {
    const int idx = tid + NTHR * i;
    const int row = idx / CPRD, col = idx - row * CPRD;
    d_row[i] = row;
    d_ok[i] = row < TN_ROWS && n0 + col * 8 < Np;
    d_lds[i] = (row < TN_ROWS ? row : 0) * LDD + col * 8;
    d_ptr[i] = dy + (mbeg + (row < TN_ROWS ? row : 0)) * lddy + n0 + col * 8;
    d_off[i] = (unsigned)(((mbeg + (row < TN_ROWS ? row : 0)) * lddy + n0 + col * 8) << 1);
  }
  // x staging: fixed chunk column -> fixed tap (ky, kx) and channel offset; rows xr + XSTEP*i
  const int xc = tid % CPRX;
  const int xr = tid / CPRX;
  const int k = kb + xc * 8;
  const bool k_ok = k < K;
  int kyo = 0, kxo = 0, c_in = 0;
  if (k_ok) {
    const int tap = k / g.Cp;
    c_in = k - tap * g.Cp;
    const int ky = tap / g.KW;
    kyo = ky - g.pad;
    kxo = tap - ky * g.KW - g.pad;
  }
  static_assert(!PW || BUF, "the pointwise variant addresses x with buffer offsets");
  int r_b[PW ? 1 : XCH], r_y[PW ? 1 : XCH], r_x[PW ? 1 : XCH];
  unsigned x_off[PW ? XCH : 1];  // PW: running byte offset of this thread's chunk in staged row i
  if constexpr (PW) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) x_off[i] = (unsigned)(((mbeg + xr + XSTEP * i) * (long)g.ldx + k) << 1);
  } else {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const RowCoord rc = decode_row(mbeg + xr + XSTEP * i, M, g);
      r_b[i] = rc.b;
      r_y[i] = rc.oy;
      r_x[i] = rc.ox;
    }
  }
  const unsigned x_step = (unsigned)((long)TN_ROWS * g.ldx * 2);
  __amdgpu_buffer_rsrc_t rs_x, rs_d;
  if constexpr (BUF) {
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, (int)x_bytes, 0x00020000);
    rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dy, (short)0, (int)dy_bytes, 0x00020000);
  }
  const unsigned d_step = (unsigned)((long)TN_ROWS * lddy * 2);
  const unsigned c_in2 = (unsigned)c_in << 1;

  // Staging registers.  DEEP (tiles whose register budget allows it): two sets, i.e. the loads of chunk it+2 AND it+3 are in
  // flight behind chunk it's products - with one set a 64-row iteration lasted as long as a global load takes to return
  // (~4 200 cycles against 1 536 on the matrix cores)
  // (no room for the second set beside the im2col decode's registers, nor beside the GELU polynomial's temporaries at TNn = 6)
  // (an 8-wave 128 x 256 tile has room for both sets beside the bias sums or the GELU polynomial - built, measured, slower than
  // the 192-wide tile with one set: profiles/experiments/README.md)
  // (four 512-register waves of 96 x 128 - 14 fragments per 48 products instead of 10 per 24 - were compiled: the 192 accumulators
  // go to AGPRs, but staging sets + fragments exceed the 256 architectural registers and spill, with or without the second set)
  // (the 384 x 128 tile stages 6 + 2 chunks per thread: its second set spills 16 dwords)
  constexpr bool DEEP = VKAS_TN_DEEP && PW && TNn <= 6 && !XG && NOBIAS && WN * WK >= (VKAS_TN_DEEP > 1 ? 4 : 8) && DCH + XCH <= 7;
  elem8 rdA[DCH], rxA[XCH], rdB[DEEP ? DCH : 1], rxB[DEEP ? XCH : 1];
  long mcur = mbeg;
  auto load_tile = [&](elem8 (&rd)[DCH], elem8 (&rx)[XCH]) {
    constexpr unsigned OOB = 0xFFFFFFF0u;  // beyond any descriptor: the buffer load returns zeros
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const bool ok = d_ok[i] && mcur + d_row[i] < mend;
      if constexpr (BUF) {
        rd[i] = __builtin_bit_cast(elem8, __builtin_amdgcn_raw_buffer_load_b128(rs_d, ok ? d_off[i] : OOB, 0, 0));
        d_off[i] += d_step;
      } else {
        elem8 vd = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) vd = *reinterpret_cast<const elem8*>(d_ptr[i]);
        rd[i] = vd;
        d_ptr[i] += (long)TN_ROWS * lddy;
      }
    }
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      if constexpr (PW) {
        const bool ok = k_ok && mcur + xr + XSTEP * i < mend;
        rx[i] = __builtin_bit_cast(elem8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? x_off[i] : OOB, 0, 0));
        x_off[i] += x_step;
        continue;
      }
      const int iy = r_y[i] * g.stride + kyo, ix = r_x[i] * g.stride + kxo;
      const bool ok = k_ok && mcur + xr + XSTEP * i < mend && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
      if constexpr (BUF) {
        const unsigned off = ((((unsigned)(r_b[i] * g.Hin + iy) * (unsigned)g.Win + (unsigned)ix) * (unsigned)g.ldx) << 1) + c_in2;
        rx[i] = __builtin_bit_cast(elem8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, 0, 0));
      } else {
        elem8 vx = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) vx = *reinterpret_cast<const elem8*>(x + (((long)r_b[i] * g.Hin + iy) * g.Win + ix) * (long)g.ldx + c_in);
        rx[i] = vx;
      }
      r_x[i] += TN_ROWS;  // advance this row by TN_ROWS output pixels, no divisions
      while (r_x[i] >= g.Wout) {
        r_x[i] -= g.Wout;
        if (++r_y[i] == g.Hout) {
          r_y[i] = 0;
          ++r_b[i];
        }
      }
    }
    mcur += TN_ROWS;
  };
  auto store_tile = [&](int buf, elem8 (&rd)[DCH], elem8 (&rx)[XCH]) {
    elem_t* Ds = lds + buf * TILE;
    elem_t* Xs = Ds + TN_ROWS * LDD;
#pragma unroll
    for (int i = 0; i < DCH; ++i)
      if ((TN_ROWS * CPRD) % NTHR == 0 || tid + NTHR * i < TN_ROWS * CPRD) *reinterpret_cast<elem8*>(Ds + d_lds[i]) = rd[i];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      elem8 v = rx[i];
      if constexpr (XG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // two elements per packed instruction; the same values as gelu_t
          const f32x2 gq = gelu2_t<elem_t>(f32x2{(float)v[2 * q], (float)v[2 * q + 1]});
          v[2 * q] = (elem_t)gq.x;
          v[2 * q + 1] = (elem_t)gq.y;
        }
      }
      *reinterpret_cast<elem8*>(Xs + (xr + XSTEP * i) * LDX + xc * 8) = v;
      if constexpr (XG && DEEP) __builtin_amdgcn_sched_barrier(0);  // one chunk's polynomial temporaries at a time
    }
  };

  f32x4 acc[TNn][TK];
#pragma unroll
  for (int i = 0; i < TNn; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dy: the first K block's wk == 0 waves add up the dy fragments they already hold
  // bias gradient = column sums of dy: the first K block's wk == 0 waves add up the dy fragments they already hold, two
  // elements per v_dot2c_f32_{bf16,f16} against (1, 1) - 4 instructions per fragment.  (Until round 4: one conversion and one
  // add per element, ~390 vector instructions per two iterations in the loop of every wave - the branch is wave uniform, the
  // registers are not -, which pushed the 8-wave tiles into scratch once both staging sets were really live.)
  const bool do_bias = !NOBIAS && gb != nullptr && first_k_tile && wk == 0;
  float bsum[TNn];
#pragma unroll
  for (int i = 0; i < TNn; ++i) bsum[i] = 0.f;

  const long nrows = mend - mbeg;
  const int nit = (int)((nrows + TN_ROWS - 1) / TN_ROWS);
  // Every load_tile / store_tile below is UNCONDITIONAL (a chunk behind the last one is all out-of-range requests, which
  // return zeros): the compiler derives its s_waitcnt vmcnt(N) from the number of loads it can prove to be outstanding, and
  // with `if (it + 3 < nit) load_tile(...)` it had to assume none were - store_tile of the older set then waited vmcnt(0),
  // i.e. for the set issued half an iteration earlier as well, and the second set bought nothing (round 4, ISA census).
  if (nit > 0) {
    load_tile(rdA, rxA);
    store_tile(0, rdA, rxA);
    load_tile(rdA, rxA);  // set A now holds chunk 1
    if constexpr (DEEP) load_tile(reinterpret_cast<elem8(&)[DCH]>(rdB), reinterpret_cast<elem8(&)[XCH]>(rxB));  // set B: chunk 2
  }
  __syncthreads();
  // one iteration; rdS / rxS hold chunk it+1 (DEEP: chunk c >= 1 travels in set A when c is odd, in set B when even)
  auto iteration = [&](int it, elem8 (&rdS)[DCH], elem8 (&rxS)[XCH]) {
    const int buf = it & 1;
    const elem_t* Ds = lds + buf * TILE;
    const elem_t* Xs = Ds + TN_ROWS * LDD;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // the second half's fragments are not read ahead of the first half's products: with both halves' fragments, two staging
      // sets and the accumulators live at once the 8-wave tiles spill, and a scratch reload waits for EVERY outstanding load
      if (DEEP && s == 1) __builtin_amdgcn_sched_barrier(0);
      elem8 fd[TNn], fx[TK];
#pragma unroll
      for (int i = 0; i < TNn; ++i) fd[i] = tr_frag<LDD>(Ds, s * 32, wn * TNn * 16 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < TK; ++j) fx[j] = tr_frag<LDX>(Xs, s * 32, wk * TK * 16 + j * 16, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TNn; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
          acc[i][j] = VKAS_MFMA16(fd[i], fx[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < TNn; ++i) {
          const u32x4 w = __builtin_bit_cast(u32x4, fd[i]);
#pragma unroll
          for (int q = 0; q < 4; ++q) bsum[i] = dot2_ones(w[q], bsum[i]);
        }
      }
      if (s == 0) {  // stage chunk it+1 and reissue loads into the set it leaves behind the MFMAs
        // (XG: the polynomial's temporaries only fit once the first half's fragments are dead - keep it behind those MFMAs)
        if constexpr (XG && DEEP) __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1, rdS, rxS);
        load_tile(rdS, rxS);
      }
    }
    __syncthreads();
  };
  if constexpr (DEEP) {
    int it = 0;
    for (; it + 1 < nit; it += 2) {
      iteration(it, rdA, rxA);
      iteration(it + 1, reinterpret_cast<elem8(&)[DCH]>(rdB), reinterpret_cast<elem8(&)[XCH]>(rxB));
    }
    if (it < nit) iteration(it, rdA, rxA);
  } else {
    for (int it = 0; it < nit; ++it) iteration(it, rdA, rxA);
  }
  // D[row = n_local][col = k_local]: lane holds col = lane&15, rows (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < TNn; ++i) {
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int kk = kb + wk * TK * 16 + j * 16 + (lane & 15);
      if (kk >= K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * TNn * 16 + i * 16 + (lane >> 4) * 4 + r;
#ifdef TN_NOATOMIC  // timing-only ablation (profiles/build_variant.sh): the reduction tail is skipped
        if (n < Np && acc[i][j][r] == 1.2345f) atomicAdd(gw + (long)n * K + kk, acc[i][j][r]);
#else
        if (n < Np) atomicAdd(gw + (long)n * K + kk, acc[i][j][r]);
#endif
      }
    }
  }
  if (do_bias) {  // lanes l, l^16, l^32, l^48 hold different rows of the same column
#pragma unroll
    for (int i = 0; i < TNn; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wn * TNn * 16 + i * 16 + (lane & 15);
      if (lane < 16 && n < Np) atomicAdd(gb + n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of the same row-aligned 3x3 / stride 1 / pad 1 convolutions (rows a multiple of 64 pixels):
//   gw[n][ky][kx][c] += sum over the block's pixels of dy[pixel][n] * x[pixel + (ky-1, kx-1)][c].
// A block owns 128 output channels x (3 kx taps x 128 input channels) for ONE ky and walks its pixel split 64 pixels
// (a piece of one image row) at a time: the 66-pixel x slab of input row oy+ky-1 serves the three kx taps as
// row-shifted transposing reads, so a chunk stages 16 KB of dy + 17 KB of x for 2*64*128*384 flops (the generic TN
// kernel: 61 KB for 2*64*224*256).  Staging is LDS-DMA into a ring of 4 chunk buffers (2 chunks in flight behind
// the one being read); both operands stay [pixel][channel] with 256-byte rows, the 32-byte block index XOR-ed with
// (row & 7) - applied to the DMA source chunk and to the reads - so the 8 rows a half-wave touches in one
// ds_read_b64_tr_b16 sit on distinct banks.  8 waves: every wave holds all 8 n tiles x 3 of the 24 K-column tiles.
// Same two-group READ / MFMA phase schedule as conv3x3_slab_mfma_kernel (chunk t: group 0 reads in interval 2t,
// group 1 in 2t+1; its buffer is re-filled from interval 2t+2 on; a wave's issues for chunk t+3 happen in its
// READ(t) phase and are retired - all but the two youngest chunks' worth - in front of the barrier ending interval
// 2t+1, while chunk t+1 is first read in interval 2t+2).
constexpr int WG_ROWS = 64, WG_N = 128, WG_C = 128;
constexpr int WG_DY = WG_ROWS * WG_N;           // elements
constexpr int WG_X = 68 * WG_C;                 // rows 0..63 pixels, 64 / 65 left / right halo, 66 / 67 unused
constexpr int WG_BUF = WG_DY + WG_X;

typedef s16x4 __attribute__((address_space(3))) * wg_lds_ptr;
// k-permuted transposed fragment (see tr_frag) from two byte addresses inside the LDS window: elements 0..3 <- a0,
// elements 4..7 <- a1 (16 rows further)
__device__ __forceinline__ elem8 wg_frag2(unsigned a0, unsigned a1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_ptr)(uintptr_t)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_ptr)(uintptr_t)a1);
  union { struct { s16x4 l, h; } s; elem8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

template <int TNn>
__global__ __launch_bounds__(512) void conv3x3_wgrad_slab_kernel(const elem_t* __restrict__ x, vkas_conv_geom g,
                                                                 const elem_t* __restrict__ dy, long lddy, int Np,
                                                                 long M, int K, int chunks_per_split,
                                                                 float* __restrict__ gw, float* __restrict__ gb,
                                                                 unsigned x_bytes, unsigned dy_bytes) {
  // TNn = 8: 128 output channels per block, dy rows of 256 B with the XOR swizzle.  TNn = 7: 112 channels, rows of
  // 224 B = 56 banks - consecutive rows already start 8 banks apart, the image stays linear (N = 776: 7 tiles, 1% padding).
  // TNn = 6: 96 channels, rows of 192 B with rotated blocks (N = 192, the probability head: two tiles, no padding).
  constexpr int TK = 3;
  constexpr int BNn = TNn * 16, PD = BNn * 2;  // dy row pitch in bytes
  constexpr int DWI = WG_ROWS * PD / 1024;     // dy wave instructions per chunk (16 or 14)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  __shared__ __attribute__((aligned(1024))) elem_t lds[4 * WG_BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = g.Hin, W = g.Win, Cp = g.Cp;
  // work order: as in gemm_tn_mfma_kernel (contiguous runs per XCD, all tiles of one pixel split before the next)
  const unsigned ntn = (unsigned)((Np + BNn - 1) / BNn), ncb = (unsigned)((Cp + WG_C - 1) / WG_C);
  const unsigned tiles = ntn * 3u * ncb;
  const unsigned total = gridDim.x;
  const unsigned xcd = blockIdx.x & 7u, slot8 = blockIdx.x >> 3;
  const unsigned q8 = total >> 3, r8 = total & 7u;
  const unsigned work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
  const unsigned tile = work % tiles;
  // (n tiles fastest within a split; the 9 (ky, channel block) tiles of an n tile adjacent instead - "n tile slowest" -
  // was measured in round 2 and is 2-3 % slower although it halves the dy re-reads: the kernel is not HBM bound)
  const int n0 = (int)(tile % ntn) * BNn;
  const int ky = (int)((tile / ntn) % 3u);
  const int cb = (int)(tile / (ntn * 3u));
  const long total_chunks = M / WG_ROWS;
  const long c_beg = (long)(work / tiles) * chunks_per_split;
  const int nchunks = (int)((c_beg + chunks_per_split <= total_chunks ? chunks_per_split : total_chunks - c_beg));

  // DMA roles.  A wave instruction fills 4 rows x 16 chunk positions (1 KB); wave w takes instructions w and w + 8 of
  // the 16 of each operand: rows 4w + (lane>>4) and 32 + 4w + (lane>>4).  Position c' of row r holds logical 16-byte
  // chunk (((c'>>1) ^ (r & 7)) << 1) | (c' & 1); r & 7 = 4*(w&1) + (lane>>4) for both instructions.
  const int lrow = lane >> 4, cpos = lane & 15;
  const int rkey = 4 * (wave & 1) + lrow;
  const int lchunk = (((cpos >> 1) ^ rkey) << 1) | (cpos & 1);
  const bool x_ok = cb * WG_C + lchunk * 8 < Cp;
  // dy: instruction q of this wave is instruction w + 8q of the chunk; TNn = 8: row 4w + 32q + (lane>>4), swizzled chunk;
  // TNn = 7: lane-linear over [64][14] chunks (instructions 14, 15 do not exist: waves 6, 7 issue one)
  unsigned d_lane[2];
  bool d_ok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int row, c;
    if constexpr (TNn == 8) {
      row = 4 * wave + 32 * q + lrow;
      c = lchunk;
    } else if constexpr (TNn == 7) {
      const int ci = (wave + 8 * q) * 64 + lane;
      row = ci / 14;
      c = ci - row * 14;
    } else {
      // TNn = 6: rows of 192 B = 48 banks - rows r and r + 4 would start on the same bank, so the 32-byte blocks of a row are
      // rotated by (r >> 2) & 3 positions: the 16 rows of a transposing read then sit two per bank group, the minimum
      static_assert(TNn == 6, "dy staging knows 6, 7 and 8 column tiles");
      const int ci = (wave + 8 * q) * 64 + lane;
      row = ci / 12;
      const int cp = ci - row * 12;                       // position inside the LDS row
      const int lb = ((cp >> 1) + 6 - ((row >> 2) & 3)) % 6;  // logical block held by that position
      c = lb * 2 + (cp & 1);
    }
    d_ok[q] = n0 + c * 8 < Np && c * 8 < BNn;
    d_lane[q] = (unsigned)(row * lddy + n0 + c * 8) * 2u;   // + chunk pixel base
  }
  const int d_count = (DWI == 16 || wave + 8 < DWI) ? 2 : 1;
  const unsigned x_lane = (unsigned)((4 * wave + lrow) * g.ldx + cb * WG_C + lchunk * 8) * 2u;
  // halo: lanes 0..3 of wave w carry positions 4w .. 4w+3 of the 32 halo chunks (row 64: pixel -1, row 65: pixel 64)
  const int hc = wave * 4 + (lane & 3);
  const int hrow = hc >> 4;
  const int hpos = hc & 15;
  const int hl = (((hpos >> 1) ^ hrow) << 1) | (hpos & 1);   // (64 + hrow) & 7 = hrow
  const bool hx_ok = cb * WG_C + hl * 8 < Cp;
  // LDS-DMA through vkas_lds_dma16 (inline assembly, vkas_common.h), not the compiler's builtin (round 4): with the builtin,
  // SIInsertWaitcnts put an `s_waitcnt vmcnt(0)` in front of the first transposing read of EVERY chunk (the ds_read_tr
  // builtins carry memory operands it takes for aliases of the DMA destinations) - the two chunks this kernel keeps in flight
  // behind the one being read (counted vmcnt(10) / vmcnt(8) below) were drained at the top of every iteration.
  const u32x4 rs_x = vkas_make_rsrc(x, x_bytes);
  const u32x4 rs_d = vkas_make_rsrc(dy, dy_bytes);
  const unsigned lds_dma0 = vkas_lds_addr(lds);
  // issue cursor: chunk -> (image, row, first pixel); 64 | W keeps a chunk inside one image row
  long ic = c_beg;
  int i_b, i_y, i_x;
  {
    const long pix = c_beg * WG_ROWS;
    const int hw = H * W;
    i_b = (int)(pix / hw);
    const int rem = (int)(pix - (long)i_b * hw);
    i_y = rem / W;
    i_x = rem - i_y * W;
  }
  auto issue_chunk = [&](int buf) {  // 5 instructions per wave
    const unsigned Ds = lds_dma0 + (unsigned)(buf * WG_BUF) * 2u;  // LDS byte addresses of the chunk buffer's dy / x parts
    const unsigned Xs = Ds + (unsigned)WG_DY * 2u;
    const unsigned d_base = (unsigned)(ic * WG_ROWS * lddy) * 2u;
    const int iy = i_y + ky - 1;
    const bool row_ok = (unsigned)iy < (unsigned)H;
    const unsigned x_base = (unsigned)(((i_b * H + iy) * W + i_x) * g.ldx) * 2u;
    unsigned xl = x_lane;
    asm volatile("" : "+v"(xl));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && d_count == 1) break;  // wave-uniform
      const unsigned vd = d_ok[q] ? d_base + d_lane[q] : OOB;
      vkas_lds_dma16(rs_d, Ds + (unsigned)((q * 8 + wave) * 1024), vd);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned vx = (row_ok && x_ok) ? x_base + xl + (unsigned)(q * 32 * g.ldx * 2) : OOB;
      vkas_lds_dma16(rs_x, Xs + (unsigned)((q * 8 + wave) * 1024), vx);
    }
    {
      const int hpix = hrow == 0 ? i_x - 1 : i_x + WG_ROWS;
      const bool ok = row_ok && hx_ok && (unsigned)hpix < (unsigned)W;
      const unsigned vh = ok ? (unsigned)(((i_b * H + iy) * W + hpix) * g.ldx + cb * WG_C + hl * 8) * 2u : OOB;
      if (lane < 4)
        vkas_lds_dma16(rs_x, Xs + (unsigned)((64 * WG_C + wave * 32) * 2), vh);
    }
    ++ic;
    i_x += WG_ROWS;
    if (i_x == W) {
      i_x = 0;
      if (++i_y == H) { i_y = 0; ++i_b; }
    }
  };

  f32x4 acc[TNn][TK];
#pragma unroll
  for (int i = 0; i < TNn; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dy: in the (ky = 0, cb = 0) blocks wave w adds up the dy fragments of n tile w (every wave
  // holds all of them anyway).  One wave summing all eight tiles spent ~250 vector operations per chunk in its MFMA phase,
  // which - behind the per-chunk barriers - made every ninth workgroup of the launch ~25 % slower than the rest.
  const bool do_bias = gb != nullptr && ky == 0 && cb == 0 && wave < TNn;
  float bsum = 0.f;
  // this wave's K-column tiles: columns wave*48 + 16 j of [kx][128 channels]
  int x_shift[TK], x_blk[TK];
#pragma unroll
  for (int j = 0; j < TK; ++j) {
    const int col = wave * 48 + j * 16;
    x_shift[j] = (col >> 7) - 1;
    x_blk[j] = (col & 127) >> 4;
  }
  // Fragment addresses (bytes from the start of a chunk buffer) of the (K half 0, rows +0) read; the other three reads
  // of a fragment are +4096 (16 rows) / +8192 (K half 1): the XOR key (row & 7) is the same for all four.  The x slab
  // is read row-shifted by kx - 1: only pixel -1 (half 0, first read, shift -1) and pixel 64 (half 1, second read,
  // shift +1) leave the pattern and go to the halo rows 64 / 65.
  const int fr = 4 * (lane >> 4) + ((lane & 15) >> 2);  // row of this lane inside a 16-row group
  const int fp = (lane & 3) * 8;                         // byte offset of its 4 columns inside the 32-byte block
  unsigned d_addr[TNn], x_addr[TK], x_first[TK], x_last[TK];
#pragma unroll
  for (int i = 0; i < TNn; ++i)
    d_addr[i] = (unsigned)(fr * PD + ((TNn == 8 ? (i ^ (fr & 7)) : (TNn == 6 ? (i + (fr >> 2)) % 6 : i)) << 5) + fp);
#pragma unroll
  for (int j = 0; j < TK; ++j) {
    const int r = fr + x_shift[j];
    x_addr[j] = (unsigned)(WG_DY * 2 + r * 256 + ((x_blk[j] ^ (r & 7)) << 5) + fp);
    x_first[j] = r < 0 ? (unsigned)(WG_DY * 2 + 64 * 256 + ((x_blk[j] ^ 0) << 5) + fp) : x_addr[j];
    x_last[j] = r + 48 > 63 ? (unsigned)(WG_DY * 2 + 65 * 256 + ((x_blk[j] ^ 1) << 5) + fp) : x_addr[j] + 12288u;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(wg_lds_ptr)lds;

  const int grp = wave >> 2;
#if defined(VKAS_TRACE) && !defined(VKAS_MFMA_F16)
  // per (workgroup, wave group) sums of s_memtime cycles: 0 fragment reads (+ lgkmcnt), 1 DMA issue, 2 retire (vmcnt),
  // 3 barrier after READ, 4 MFMA phase, 5 retire of group 0, 6 barrier after MFMA, 7 chunks   (profiles/trace_wgrad.py)
  unsigned long long tw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tw_last = __builtin_readcyclecounter();
#define WG_TR(i)                                                  \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    tw[i] += now_ - tw_last;                                      \
    tw_last = now_;                                               \
  } while (0)
#else
#define WG_TR(i)
#endif
  if (nchunks > 0) issue_chunk(0);
  if (nchunks > 1) issue_chunk(1);
  if (nchunks > 2) issue_chunk(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nchunks; ++t) {
    const unsigned bufb = lds_base + (unsigned)((t & 3) * WG_BUF * 2);
    // ---- READ phase
    elem8 fd[2][TNn], fx[2][TK];
#pragma unroll
    for (int i = 0; i < TNn; ++i) {
      const unsigned a = bufb + d_addr[i];
      fd[0][i] = wg_frag2(a, a + 16u * PD);
      fd[1][i] = wg_frag2(a + 32u * PD, a + 48u * PD);
    }
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const unsigned a = bufb + x_addr[j];
      fx[0][j] = wg_frag2(bufb + x_first[j], a + 4096u);
      fx[1][j] = wg_frag2(a + 8192u, bufb + x_last[j]);
    }
    asm volatile("" ::: "memory");
#if defined(VKAS_TRACE) && !defined(VKAS_MFMA_F16)
    if (t == 0) tw_last = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    WG_TR(0);
    const bool issued = t + 3 < nchunks;
    if constexpr ((ABL & 1) == 0) {
      if (issued) issue_chunk((t + 3) & 3);
    }
    WG_TR(1);
    // chunks t+2 and t+3 (2 x 5 instructions, 2 x 4 for the waves without a second dy instruction) may stay in flight;
    // near the end of the split fewer were issued.  (Issuing chunk t+3 between the two K halves of the MFMA phase instead
    // - the READ phase is the longer one: fragment reads 860 + DMA issue 400 cycles against 850, profiles/trace_wgrad.py -
    // was measured in round 2: the issue cost moves with it, the MFMA phase grows to 1190 cycles, no gain.)
    auto retire = [&]() {
      if (t + 3 >= nchunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (d_count == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };
    if (grp == 1) retire();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WG_TR(2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WG_TR(3);
    // ---- MFMA phase
    if constexpr ((ABL & 16) != 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TNn; ++i) asm volatile("" ::"v"(fd[h][i]));
#pragma unroll
        for (int j = 0; j < TK; ++j) asm volatile("" ::"v"(fx[h][j]));
      }
    } else {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < TNn; ++i)
#pragma unroll
          for (int j = 0; j < TK; ++j)
            acc[i][j] = VKAS_MFMA16(fd[h][i], fx[h][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < TNn; ++i) {
        if (wave != i) continue;  // wave-uniform
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 8; ++q) bsum += (float)fd[h][i][q];
      }
    }
    WG_TR(4);
    if (grp == 0) retire();
    WG_TR(5);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WG_TR(6);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // group 1's last MFMA phase: every wave passes the same number of barriers
#if defined(VKAS_TRACE) && !defined(VKAS_MFMA_F16)
  if ((tid & 255) == 0 && blockIdx.x < 32768) {
    tw[7] = (unsigned long long)nchunks;
    for (int i = 0; i < 8; ++i) vkas_trace_buf[(blockIdx.x * 2 + grp) * 8 + i] = tw[i];
  }
#endif

  // D[row = n_local][col = k_local]: lane holds col = lane&15, rows (lane>>4)*4 + r
#pragma unroll
  for (int j = 0; j < TK; ++j) {
    const int c = cb * WG_C + x_blk[j] * 16 + (lane & 15);
    if (c >= Cp) continue;
    const long kk = (long)(ky * 3 + x_shift[j] + 1) * Cp + c;
#pragma unroll
    for (int i = 0; i < TNn; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + i * 16 + (lane >> 4) * 4 + r;
        if (n < Np) atomicAdd(gw + (long)n * K + kk, acc[i][j][r]);
      }
    }
  }
  if (do_bias) {  // lanes l, l^16, l^32, l^48 hold different rows of the same column
    float v = bsum;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int n = n0 + wave * 16 + (lane & 15);
    if (lane < 16 && n < Np) atomicAdd(gb + n, v);
  }
}

}  // namespace

template <int WM, int WN, int TM, int TN>
static void launch_nt(const void* x, const vkas_conv_geom* g, const void* Bw, int Np, long M, int K,
                      const vkas_epilogue* e, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  const long ntn = e->mode == VKAS_EPI_HEAD ? e->head.n_heads : vkas_cdiv(Np, BN);
  dim3 grid((unsigned)(vkas_cdiv(M, BM) * ntn));
  // bytes spanned by the operands (x may be a channel slice: last pixel ends after Cp of its ld channels)
  const long a_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  const long b_bytes = (long)Np * K * 2;
  static const bool no_buf = getenv("VKAS_NT_NOBUF") != nullptr;
  const bool buf = !no_buf && a_bytes < 0xFFFFFFF0L && b_bytes < 0xFFFFFFF0L;
  const elem_t* xp = (const elem_t*)x;
  const elem_t* bp = (const elem_t*)Bw;
  const unsigned ab = buf ? (unsigned)a_bytes : 0u, bb = buf ? (unsigned)b_bytes : 0u;
  if (e->mode == VKAS_EPI_HEAD) {
    if constexpr (WM * WN == 8) {  // the fused head epilogue is instantiated for the 8-wave tiles only
      if (buf) gemm_nt_mfma_kernel<WM, WN, TM, TN, true, true><<<grid, WM * WN * 64, 0, st>>>(xp, *g, bp, Np, M, K, *e, ab, bb);
      else gemm_nt_mfma_kernel<WM, WN, TM, TN, false, true><<<grid, WM * WN * 64, 0, st>>>(xp, *g, bp, Np, M, K, *e, ab, bb);
    }
  } else if (buf) {
    gemm_nt_mfma_kernel<WM, WN, TM, TN, true, false><<<grid, WM * WN * 64, 0, st>>>(xp, *g, bp, Np, M, K, *e, ab, bb);
  } else {
    gemm_nt_mfma_kernel<WM, WN, TM, TN, false, false><<<grid, WM * WN * 64, 0, st>>>(xp, *g, bp, Np, M, K, *e, ab, bb);
  }
}

#ifndef VKAS_MFMA_F16
// Tile choice of the NT kernel: returns 1 for the 4-wave 128x128 tile, else the N extent (128 / 192 / 224) of the
// 8-wave 256-row tile.  256-row tiles once there is enough work to fill the chip with them; N extent = the candidate
// with the least zero padding (ties -> wider tile, fewer re-reads of A).
int vkas_gemm_nt_tile_choice(long M, int Np) {
  static const int force = getenv("VKAS_NT_TILE") ? atoi(getenv("VKAS_NT_TILE")) : 0;
  if (force) return force;
  if (M < 16384) return 1;
  long best = -1;
  int bn = 128;
  const int cand[3] = {224, 192, 128};
  for (int c = 0; c < 3; ++c) {
    const long padded = vkas_cdiv(Np, cand[c]) * cand[c];
    if (best < 0 || padded < best) {
      best = padded;
      bn = cand[c];
    }
  }
  return bn;
}

// Ring depth of gemm_nt_ring_kernel for a launch the tile choice above gives the 128 x 128 tile (0 = the register-staged
// kernel).  At most one round of workgroups (<= 256 tiles): four stages (128 KB of LDS, one workgroup per CU, three K tiles in
// flight) - the launch lasts as long as one workgroup's K loop; more tiles: two stages, so that two workgroups share a CU and
// one's prologue / epilogue sits behind the other's K loop (profiles/sweep_small.py: 4 stages 26.6 / 37.1 us against 31.7 /
// 52.1 at M = 7 168, N = 512, K = 2 048 / M = 1 792, N = 1 024, K = 4 096; 2 stages 34.2 against 44.3 at M = 7 168,
// N = 2 048, K = 512; the register-staged kernel: 42.2, 70.2 and 40.0).  VKAS_NT_RING = 0 keeps the register-staged kernel,
// 2 / 3 / 4 force a depth.
int vkas_gemm_nt_ring_stages(const vkas_conv_geom* g, int Np) {
  static const int ring_env = getenv("VKAS_NT_RING") ? atoi(getenv("VKAS_NT_RING")) : -1;
  const long M = (long)g->B * g->Hout * g->Wout;
  const long K = (long)g->KH * g->KW * g->Cp;
  const long a_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  if (a_bytes >= 0xFFFFFFF0L || (long)Np * K * 2 >= 0xFFFFFFF0L) return 0;  // operands addressed with 32-bit buffer offsets
  if (ring_env >= 0) return ring_env >= 2 && ring_env <= 4 ? ring_env : 0;
  return vkas_cdiv(M, 128) * vkas_cdiv(Np, 128) <= 256 ? 4 : 2;
}

static bool row_aligned_3x3(const vkas_conv_geom* g, int wmod) {
  return g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1 && g->Hout == g->Hin && g->Wout == g->Win &&
         g->Win % wmod == 0;
}

// conv3x3_slab_mfma_kernel: rows of whole 256-pixel tiles, operands addressable with 32-bit buffer offsets
bool vkas_nt_slab_eligible(const vkas_conv_geom* g, int Np) {
  static const bool no_slab = getenv("VKAS_NT_NOSLAB") != nullptr;
  const long K = (long)g->KH * g->KW * g->Cp;
  const long a_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  return !no_slab && row_aligned_3x3(g, 256) && a_bytes < 0xFFFFFFF0L && (long)Np * K * 2 < 0xFFFFFFF0L;
}

// conv3x3_wgrad_slab_kernel: rows of whole 64-pixel chunks, at least one full n tile and channel block
bool vkas_tn_slab_eligible(const vkas_conv_geom* g, int Np, long lddy) {
  static const bool no_slab = getenv("VKAS_TN_NOSLAB") != nullptr;
  const long M = (long)g->B * g->Hout * g->Wout;
  const long x_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  const long dy_bytes = ((M - 1) * lddy + Np) * 2;
  return !no_slab && row_aligned_3x3(g, 64) && M >= 65536 && Np >= 112 && g->Cp >= 128 && x_bytes < 0xFFFFFFF0L &&
         dy_bytes < 0xFFFFFFF0L;
}
#ifdef VKAS_TRACE
extern "C" int vkas_trace_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vkas_trace_buf), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
bool vkas_tn_slab_n112(int Np) { return vkas_cdiv(Np, 112) * 112 < vkas_cdiv(Np, 128) * 128; }  // 112-wide tiles pad less
// 96-wide tiles when they pad less than both wider ones (N = 192: no padding against 224 / 256 columns)
bool vkas_tn_slab_n96(int Np) {
  static const bool off = getenv("VKAS_TN_NO96") != nullptr;
  const long p96 = vkas_cdiv(Np, 96) * 96;
  return !off && p96 < vkas_cdiv(Np, 112) * 112 && p96 < vkas_cdiv(Np, 128) * 128;
}

#endif  // VKAS_MFMA_F16

int VKAS_MFMA_FN(vkas_gemm_nt_mfma)(const void* x, const vkas_conv_geom* g, const void* Bw, int Np, const vkas_epilogue* e,
                           hipStream_t st) {
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (M == 0) return VKAS_OK;
  int choice = vkas_gemm_nt_tile_choice(M, Np);
  if (e->mode == VKAS_EPI_HEAD) {  // one 256-row tile per head: the narrowest N extent that holds the widest head
    int wmax = 0;
    for (int h = 0; h < e->head.n_heads; ++h) wmax = e->head.np[h] > wmax ? e->head.np[h] : wmax;
    choice = wmax <= 128 ? 128 : (wmax <= 192 ? 192 : 224);
  }
  const bool big = choice != 1;
  const int bn = big ? choice : 128;
  // 3x3 / stride 1 / pad 1 with rows of whole 256-pixel tiles: the row-slab kernel
  const long a_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  const long b_bytes = (long)Np * K * 2;
  if (big && vkas_nt_slab_eligible(g, Np)) {
    const bool head = e->mode == VKAS_EPI_HEAD;
    const long ntn = head ? e->head.n_heads : vkas_cdiv(Np, bn);
    dim3 grid((unsigned)((M / 256) * ntn));
    const elem_t* xp = (const elem_t*)x;
    const elem_t* bp = (const elem_t*)Bw;
#define VKAS_SLAB(TNV)                                                                                                  \
  if (head) conv3x3_slab_mfma_kernel<TNV, true><<<grid, 512, 0, st>>>(xp, *g, bp, Np, M, K, *e, (unsigned)a_bytes, (unsigned)b_bytes); \
  else conv3x3_slab_mfma_kernel<TNV, false><<<grid, 512, 0, st>>>(xp, *g, bp, Np, M, K, *e, (unsigned)a_bytes, (unsigned)b_bytes);
    if (bn == 224) { VKAS_SLAB(7) }
    else if (bn == 192) { VKAS_SLAB(6) }
    else { VKAS_SLAB(4) }
#undef VKAS_SLAB
    VKAS_LAUNCH_CHECK("conv3x3_slab_mfma");
    return VKAS_OK;
  }
  if (!big) {
    const int nst = e->mode == VKAS_EPI_HEAD ? 0 : vkas_gemm_nt_ring_stages(g, Np);
    if (nst >= 2) {
      dim3 grid((unsigned)(vkas_cdiv(M, 128) * vkas_cdiv(Np, 128)));
      const elem_t* xp = (const elem_t*)x;
      const elem_t* bp = (const elem_t*)Bw;
      if (nst == 2) gemm_nt_ring_kernel<2><<<grid, 512, 0, st>>>(xp, *g, bp, Np, M, K, *e, (unsigned)a_bytes, (unsigned)b_bytes);
      else if (nst == 3) gemm_nt_ring_kernel<3><<<grid, 512, 0, st>>>(xp, *g, bp, Np, M, K, *e, (unsigned)a_bytes, (unsigned)b_bytes);
      else gemm_nt_ring_kernel<4><<<grid, 512, 0, st>>>(xp, *g, bp, Np, M, K, *e, (unsigned)a_bytes, (unsigned)b_bytes);
      VKAS_LAUNCH_CHECK("gemm_nt_ring");
      return VKAS_OK;
    }
    launch_nt<2, 2, 4, 4>(x, g, Bw, Np, M, K, e, st);
  }
  else if (bn == 224) launch_nt<4, 2, 4, 7>(x, g, Bw, Np, M, K, e, st);
  else if (bn == 192) launch_nt<4, 2, 4, 6>(x, g, Bw, Np, M, K, e, st);
  else launch_nt<4, 2, 4, 4>(x, g, Bw, Np, M, K, e, st);
  VKAS_LAUNCH_CHECK("gemm_nt_mfma");
  return VKAS_OK;
}

template <int WN, int WK, int TNn, int TK, bool XG>
static void launch_tn(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, long M, int K, float* gw,
                      float* gb, bool one_split, hipStream_t st) {
  constexpr int BNn = WN * TNn * 16, BKc = WK * TK * 16;
  const long tiles = vkas_cdiv(Np, BNn) * vkas_cdiv(K, BKc);
  // Splits over M: pick the count that minimises a two-term cost model.
  //   main loop: rounds of resident workgroups (256 CUs x 1 block of 8 waves | 2 blocks of 4) x 64-row iterations of a
  //     split x time per iteration (measured: ~1.2 us for the 8-wave tiles, ~0.5 us for the 4-wave tile);
  //   reduction: every split adds a full copy of its tile with fp32 atomics, ~1.3 TB/s chip-wide (MI355X_MICROARCH.md).
  // On the stage-3/4 weight gradients (M = 16-64 K rows) the atomic tail was half the launch with the old "3 rounds"
  // rule.  Ties go to multiples of 8 (whole splits per XCD, see the kernel's work order).
  const long resident = 256L * (WN * WK >= 8 ? 1 : 2);
  const double t_iter = WN * WK >= 8 ? 1.2e-6 : 0.5e-6;
  const double tile_bytes = (double)BNn * BKc * 4.0;
  // one_split: every output tile is reduced over all M rows by one workgroup, in row order, and added once to the zeroed
  // gw: the result does not depend on the order workgroups run in (vkas_conv_gemm_wgrad_ordered)
  const long max_splits = one_split ? 1 : vkas_cdiv(M, 4 * TN_ROWS);
  long splits = 1;
  double best_t = 1e30;
  for (long sp = 1; sp <= max_splits && sp * tiles <= 8 * resident; ++sp) {
    const long blocks = sp * tiles;
    const double t = (double)vkas_cdiv(blocks, resident) * (double)vkas_cdiv(vkas_cdiv(M, sp), TN_ROWS) * t_iter +
                     (double)blocks * tile_bytes / 1.3e12;
    if (t < best_t * (sp % 8 == 0 ? 1.02 : 0.999)) {
      best_t = t;
      splits = sp;
    }
  }
  if (splits > 65535) splits = 65535;
  long rows = vkas_cdiv(M, splits);
  rows = vkas_cdiv(rows, TN_ROWS) * TN_ROWS;
  splits = vkas_cdiv(M, rows);
  dim3 grid((unsigned)(tiles * splits));
  const long x_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  const long dy_bytes = ((M - 1) * lddy + Np) * 2;
  static const bool no_buf = getenv("VKAS_TN_NOBUF") != nullptr;
  const bool pointwise = g->KH == 1 && g->KW == 1 && g->stride == 1 && g->pad == 0 && g->Hin == g->Hout && g->Win == g->Wout;
  if (!no_buf && x_bytes < 0xFFFFFFF0L && dy_bytes < 0xFFFFFFF0L && pointwise && gb == nullptr && !XG && TNn == 6)
    gemm_tn_mfma_kernel<WN, WK, TNn, TK, true, XG, true, !XG && TNn == 6><<<grid, WN * WK * 64, 0, st>>>(
        (const elem_t*)x, *g, (const elem_t*)dy, lddy, Np, M, K, rows, gw, gb, (unsigned)x_bytes, (unsigned)dy_bytes);
  else if (!no_buf && x_bytes < 0xFFFFFFF0L && dy_bytes < 0xFFFFFFF0L && pointwise)
    gemm_tn_mfma_kernel<WN, WK, TNn, TK, true, XG, true><<<grid, WN * WK * 64, 0, st>>>(
        (const elem_t*)x, *g, (const elem_t*)dy, lddy, Np, M, K, rows, gw, gb, (unsigned)x_bytes, (unsigned)dy_bytes);
  else if (!no_buf && x_bytes < 0xFFFFFFF0L && dy_bytes < 0xFFFFFFF0L)
    gemm_tn_mfma_kernel<WN, WK, TNn, TK, true, XG><<<grid, WN * WK * 64, 0, st>>>((const elem_t*)x, *g, (const elem_t*)dy, lddy, Np,
                                                                             M, K, rows, gw, gb, (unsigned)x_bytes,
                                                                             (unsigned)dy_bytes);
  else
    gemm_tn_mfma_kernel<WN, WK, TNn, TK, false, XG><<<grid, WN * WK * 64, 0, st>>>((const elem_t*)x, *g, (const elem_t*)dy, lddy,
                                                                              Np, M, K, rows, gw, gb, 0u, 0u);
}

#ifndef VKAS_MFMA_F16
// Tile choice of the TN (wgrad) kernel: N extent 128 (4 waves, 128 K columns), 192 / 224 (8 waves, 256 K columns) or 384 (8 waves, 128 K columns):
// 8-wave tiles when there is enough work and K is wide enough; N extent = least zero padding.
int vkas_gemm_tn_tile_choice(long M, int Np, int K) {
  static const int force = getenv("VKAS_TN_TILE") ? atoi(getenv("VKAS_TN_TILE")) : 0;
  if (force) return force;
  int bn = 128;
  if (M >= 16384 && K >= 192) {
    long best = vkas_cdiv(Np, 128) * 128;
    const int cand[2] = {192, 224};
    for (int c = 0; c < 2; ++c) {
      const long padded = vkas_cdiv(Np, cand[c]) * cand[c];
      if (padded <= best) {
        best = padded;
        bn = cand[c];
      }
    }
    // 384 (N) x 128 (K) instead of 192 x 256 - the same 96 x 64 per wave - where the 256-wide K tiles would be padded and the
    // 128-wide ones are not (K = 384: the W1 weight gradient of a C = 384 ConvNeXt MLP ran a quarter of its products on zeros)
    if (bn == 192 && Np % 384 == 0 && vkas_cdiv(K, 128) * 128 < vkas_cdiv(K, 256) * 256) bn = 384;
  }
  return bn;
}

#endif  // VKAS_MFMA_F16

int VKAS_MFMA_FN(vkas_gemm_tn_mfma)(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, float* gw,
                           float* gb, int flags, hipStream_t st) {
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (M == 0) return VKAS_OK;
  const bool x_gelu = (flags & 1) != 0, one_split = (flags & 2) != 0;
  // row-aligned 3x3 / stride 1 / pad 1 with wide operands: the slab kernel
  const long x_bytes = (((long)g->B * g->Hin * g->Win - 1) * g->ldx + g->Cp) * 2;
  const long dy_bytes = ((M - 1) * lddy + Np) * 2;
  if (!x_gelu && !one_split && vkas_tn_slab_eligible(g, Np, lddy)) {
    const bool n96 = vkas_tn_slab_n96(Np);
    const bool n112 = !n96 && vkas_tn_slab_n112(Np);
    const long tiles = vkas_cdiv(Np, n96 ? 96 : (n112 ? 112 : 128)) * 3 * vkas_cdiv(g->Cp, 128);
    const long chunks = M / 64;
    // Pixel splits: whole splits per XCD (multiple of 8).  The tiles of one split walk the same dy / x chunks at the same
    // time and share them through that XCD's L2 (every operand byte is used by 9 tiles): keeping a split's tiles
    // together matters more than filling the last round of workgroups (a round-balanced, XCD-straddling split count
    // measured 5% slower).  About 3 rounds of the 256 resident workgroups, at least 16 chunks per split.
    long splits = vkas_cdiv(3 * 256, tiles);
    splits = vkas_cdiv(splits, 8) * 8;
    if (splits > chunks / 16) splits = chunks / 16 > 0 ? chunks / 16 : 1;
    const long cps = vkas_cdiv(chunks, splits);
    splits = vkas_cdiv(chunks, cps);
    if (n96)
      conv3x3_wgrad_slab_kernel<6><<<(unsigned)(tiles * splits), 512, 0, st>>>((const elem_t*)x, *g, (const elem_t*)dy, lddy, Np, M,
                                                                               K, (int)cps, gw, gb, (unsigned)x_bytes,
                                                                               (unsigned)dy_bytes);
    else if (n112)
      conv3x3_wgrad_slab_kernel<7><<<(unsigned)(tiles * splits), 512, 0, st>>>((const elem_t*)x, *g, (const elem_t*)dy, lddy, Np, M,
                                                                               K, (int)cps, gw, gb, (unsigned)x_bytes,
                                                                               (unsigned)dy_bytes);
    else
      conv3x3_wgrad_slab_kernel<8><<<(unsigned)(tiles * splits), 512, 0, st>>>((const elem_t*)x, *g, (const elem_t*)dy, lddy, Np, M,
                                                                               K, (int)cps, gw, gb, (unsigned)x_bytes,
                                                                               (unsigned)dy_bytes);
    VKAS_LAUNCH_CHECK("conv3x3_wgrad_slab");
    return VKAS_OK;
  }
  const int bn = vkas_gemm_tn_tile_choice(M, Np, K);
  if (bn == 384) {
    if (x_gelu) launch_tn<4, 2, 6, 4, true>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
    else launch_tn<4, 2, 6, 4, false>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
  } else if (x_gelu) {
    if (bn == 224) launch_tn<2, 4, 7, 4, true>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
    else if (bn == 192) launch_tn<2, 4, 6, 4, true>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
    else launch_tn<2, 2, 4, 4, true>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
  } else if (bn == 224) launch_tn<2, 4, 7, 4, false>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
  else if (bn == 192) launch_tn<2, 4, 6, 4, false>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
  else launch_tn<2, 2, 4, 4, false>(x, g, dy, lddy, Np, M, K, gw, gb, one_split, st);
  VKAS_LAUNCH_CHECK("gemm_tn_mfma");
  return VKAS_OK;
}
