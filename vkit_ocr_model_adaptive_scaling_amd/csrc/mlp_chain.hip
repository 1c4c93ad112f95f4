// ConvNeXt MLP as ONE kernel per direction (model/convnext.py:33-35,54-58: Linear(C,4C) -> GELU -> Linear(4C,C) ->
// layer scale -> stochastic depth -> residual), for C <= 512: the (pixels x 4C) intermediate never round-trips HBM
// between the two GEMMs.
//
//   forward : h = yn W1^T + b1 (stored once, bf16, for backward); g = gelu(h) stays in registers;
//             z = g W2^T + b2 (stored, C wide); out = x + rowscale[b] * colscale * z.
//   backward: dg = dz W2; dh = dg * gelu'(h) (stored once for the W1 weight gradient); dyn = dh W1.
//
// Structure (both directions are the same chain  GEMM-a (K = C) -> elementwise -> GEMM-b (K = 4C, N = C)):
//  * a wave owns TM x 16 pixel rows and ALL columns: its input rows live in registers as MFMA B fragments for the whole
//    kernel (read from HBM once, straight into the fragment layout), the C-wide output tile in fp32 accumulators.
//  * the 4C hidden units are walked in chunks of 32.  Per chunk GEMM-a produces two 16x16 D tiles per row group; with
//    v_mfma_f32_16x16x32 (operands swapped, D^T = W X^T) a lane then holds hidden units {4g..4g+3} and {16+4g..16+4g+3}
//    of one pixel - exactly one K = 32 B fragment of GEMM-b if the weight side uses the same k permutation.  So the
//    activation goes from the first matrix product into the second without touching LDS; the permutation is baked into
//    the packed weight image.
//  * weights: per chunk one pre-swizzled LDS image (GEMM-a tile [32][K] + GEMM-b tile [C][32]) that the pack kernel
//    lays out in global memory byte for byte as it must sit in LDS, so staging is a linear LDS-DMA copy
//    (buffer_load_dwordx4 ... lds), double buffered, one barrier per chunk.  All workgroups stream the same
//    147 KB - 1 MB of weights from L2.
//  * 256 threads, two workgroups per CU: one workgroup's epilogue / prologue overlaps the other's main loop.
#include <stdlib.h>

#include <type_traits>

#include "vkas_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int MODE_FWD = 0, MODE_BWD = 1;

// Timing-only ablation switches for profiles/bench_chain.py (never set in the shipped build; results are wrong when set):
// 1 no h / dh stores, 2 no epilogue traffic, 4 no GELU arithmetic, 8 no h loads (backward); pair kernel only: 16 no weight
// requests inside the loop, 32 one fragment read per MFMA phase.
#ifndef CHAIN_ABL
#define CHAIN_ABL 0
#endif
constexpr int ABL = CHAIN_ABL;

__host__ __device__ constexpr int chain_ka(int KS) { return ((KS + 1) / 2) * 64; }          // K extent of the stored GEMM-a tile
// per chunk: GEMM-a tile + GEMM-b tile + one 1-KB piece whose first 128 bytes hold the chunk's 32 GEMM-a biases (fp32)
__host__ __device__ constexpr int chain_img_elems(int KS) { return 32 * chain_ka(KS) + KS * 32 * 32 + 512; }

// GELU / d * GELU' of four stored values, two per packed instruction (vkas_common.h: gelu2_t / dgelu2_t - the same operations per
// component as gelu_t / dgelu_t, so nothing changes in the results)
template <typename T, typename V4>
__device__ __forceinline__ V4 chain_gelu4(const V4& h) {
  const f32x2 a = gelu2_t<T>(f32x2{(float)h[0], (float)h[1]}), b = gelu2_t<T>(f32x2{(float)h[2], (float)h[3]});
  V4 o;
  o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)b.x; o[3] = (T)b.y;
  return o;
}
template <typename T, typename V4>
__device__ __forceinline__ V4 chain_dgelu4(const V4& h, const f32x4& d) {
  const f32x2 a = f32x2{d[0], d[1]} * dgelu2_t<T>(f32x2{(float)h[0], (float)h[1]});
  const f32x2 b = f32x2{d[2], d[3]} * dgelu2_t<T>(f32x2{(float)h[2], (float)h[3]});
  V4 o;
  o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)b.x; o[3] = (T)b.y;
  return o;
}

// ---- packed weight image -------------------------------------------------------------------------------------------
// chunk j (hidden units 32 j .. 32 j + 31):
//   A region: sub-tiles s = 0 .. KA/64 - 1 of [32 hidden][64 k], 16-byte chunk c of row h stored at position c ^ (h & 7);
//   B region: [CB = 32 KS output rows][32 k slots] viewed as [CB / 2][64]: row n, slot group g -> row R = n >> 1, chunk
//             ((n & 1) * 4 + g) ^ (R & 7); k slot q of group g is hidden unit (q >> 2) * 16 + 4 g + (q & 3);
//   bias piece: 32 fp32 (b1 of the chunk's hidden units; zeros in the backward image), rest of the 1-KB piece unused.
// mode 0 (forward): A = W1 rows, B = W2; mode 1 (backward): A[h][c] = W2[c][h], B[c][h] = W1[h][c].
template <typename T>
__global__ void mlp_chain_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                      const float* __restrict__ b1, int C, int HID, int KS, int mode, T* __restrict__ img) {
  const int KA = chain_ka(KS);
  const int pieces = chain_img_elems(KS) / 8;  // 16-byte pieces per chunk image
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nchunks = HID / 32;
  if (idx >= (long)nchunks * pieces) return;
  const int j = (int)(idx / pieces);
  int p = (int)(idx - (long)j * pieces);
  float v[8];
  const int a_pieces = 32 * KA / 8;
  const int b_pieces = KS * 32 * 32 / 8;
  if (p >= a_pieces + b_pieces) {  // bias piece: fp32 values, 4 per 16-byte piece
    const int q = p - a_pieces - b_pieces;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < 8 && mode == 0 && b1) bv = *reinterpret_cast<const float4*>(b1 + j * 32 + q * 4);
    *reinterpret_cast<float4*>(img + idx * 8) = bv;
    return;
  }
  if (p < a_pieces) {
    const int s = p / 256, rem = p - s * 256;
    const int h = rem >> 3, cpos = rem & 7;
    const int c = cpos ^ (h & 7);
    const int k0 = s * 64 + c * 8;
    const int hid = j * 32 + h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + i;
      v[i] = k < C ? (mode == 0 ? w1[(long)hid * C + k] : w2[(long)k * HID + hid]) : 0.f;
    }
  } else {
    p -= a_pieces;
    const int R = p >> 3, cpos = p & 7;
    const int c = cpos ^ (R & 7);
    const int n = R * 2 + (c >> 2), g = c & 3;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int hid = j * 32 + (q >> 2) * 16 + g * 4 + (q & 3);
      v[q] = n < C ? (mode == 0 ? w2[(long)n * HID + hid] : w1[(long)hid * C + n]) : 0.f;
    }
  }
  store8(img + idx * 8, v);
}

struct ChainArgs {
  const void* a;        // [M][lda] input rows: yn (fwd) / dz (bwd)
  long lda;
  const void* img;      // packed weight chunks
  void* mid_out;        // fwd: h, bwd: dh   [M][ldm]
  long ldm;
  const void* mid_in;   // bwd: h            [M][ldmi]
  long ldmi;
  const float* bias_b;  // fwd: b2 (C)
  const void* res;      // fwd: x (residual) [M][ldres]
  long ldres;
  const float* colscale;  // fwd: block_scale (C)
  const float* rowscale;  // fwd: per-image keep mask / keep probability, or null
  int rows_per_image;
  void* z;              // fwd: pre-scale MLP output [M][ldz]
  long ldz;
  void* out;            // fwd: layer output, bwd: dyn [M][ldo]
  long ldo;
  long M;
  int C, HID;
  // fwd, optional: `a` holds the depthwise output and LayerNorm (helper.py:96-101 as used in convnext.py:32) is applied to
  // the rows on their way into the B fragments; ln_out / ln_stats (normalised rows, mean | rstd per row: what the standalone
  // kernel of norm.hip writes for backward) are written when given
  const float* ln_gamma;
  const float* ln_beta;
  void* ln_out;
  long ldln;
  float* ln_stats;
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct Frag<f16_t> { typedef f16x8 v8; typedef f16x4 v4; };

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// LayerNorm of the resident input rows (fused forward, round 4): a row's C channels sit in the KS fragments of the four
// lanes fr, fr + 16, fr + 32, fr + 48 (k = ks*32 + g*8 .. +7), so the statistics are per-lane sums + two cross-lane steps;
// two-pass in fp32 (mean, then centred variance, eps = 1e-6) and the same expression per element as layernorm_fwd_kernel
// (norm.hip), rounded to the storage type once.  mrow(i) = global row of this lane's row in group i.
// gamma | beta -> LDS (2 C floats at lnp), requested in FRONT of the row loads so that the two travel together: read from
// global memory inside chain_layernorm_rows they were a second, dependent round trip in every workgroup's prologue.
template <int NTHR>
__device__ __forceinline__ void chain_layernorm_stage(const ChainArgs& p, float* lnp, int tid) {
  const int cq = p.C >> 2;  // float4 pieces per vector (C % 8 == 0)
  for (int i = tid; i < 2 * cq; i += NTHR) {
    const bool hi = i >= cq;
    const int idx = hi ? i - cq : i;
    *reinterpret_cast<float4*>(lnp + (hi ? p.C : 0) + idx * 4) = *reinterpret_cast<const float4*>((hi ? p.ln_beta : p.ln_gamma) + idx * 4);
  }
}

template <typename T, int TM, int KS, typename V8, typename RowFn>
__device__ __forceinline__ void chain_layernorm_rows(V8 (&xf)[TM][KS], const ChainArgs& p, RowFn mrow, int g, const float* lnp) {
  const int C = p.C;
  float mean[TM], rstd[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)xf[i][ks][e];  // channels beyond C were loaded as zeros
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    mean[i] = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks * 32 + g * 8 < C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)xf[i][ks][e] - mean[i];
          q += d * d;
        }
      }
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    rstd[i] = rsqrtf(q / (float)C + 1e-6f);
  }
  T* Y = reinterpret_cast<T*>(p.ln_out);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = ks * 32 + g * 8;
    const bool k_ok = k < C;
    float gv[8], bv[8];
    load8(lnp + (k_ok ? k : 0), gv);
    load8(lnp + C + (k_ok ? k : 0), bv);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long m = mrow(i);
      const bool ok = k_ok && m < p.M;
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = ok ? (T)(((float)xf[i][ks][e] - mean[i]) * rstd[i] * gv[e] + bv[e]) : (T)0.f;
      xf[i][ks] = o;
      if (Y && ok) *reinterpret_cast<V8*>(Y + m * p.ldln + k) = o;
    }
  }
  if (p.ln_stats && g == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long m = mrow(i);
      if (m < p.M) {
        p.ln_stats[2 * m] = mean[i];
        p.ln_stats[2 * m + 1] = rstd[i];
      }
    }
  }
}

template <typename T, int KS, int TM, int MODE, int MINB>
__global__ __launch_bounds__(256, MINB) void mlp_chain_kernel(ChainArgs p) {
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  constexpr int NT2 = 2 * KS;                // 16-wide output column tiles of GEMM-b
  // fragment reads of a chunk's products hoisted in front of them and the 4C-wide stores batched (round 4): -5 ... -23 % from
  // 128 channels on; at <= 96 channels (TM = 4: 256 registers) the extra live fragments spill and it is 3 % slower
#ifdef CHAIN_OLD_SCHED
  constexpr bool HOIST = false;
#else
  constexpr bool HOIST = KS >= 4;
#endif
  constexpr int KA = chain_ka(KS);
  constexpr int IMG = chain_img_elems(KS);   // elements per chunk image
  constexpr int NI = IMG * 2 / 1024;         // 1-KB DMA instructions per chunk image
  constexpr int ROWS = TM * 16;              // pixel rows per wave
  constexpr int STG = ROWS * 64;             // per-wave staging of one chunk PAIR of the 4C-wide tensor: ROWS x 128 B
  constexpr int NF = ROWS / 8;               // 1-KB pieces (8 rows x 128 B) of that staging buffer
  constexpr int PE = KS * 32 + 8;            // epilogue staging row pitch (elements): odd multiple of 16 B
  constexpr int LDS_MAIN = 2 * IMG + 4 * STG, LDS_EPI = 4 * ROWS * PE;
  constexpr int LDS_ELEMS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert((IMG * 2) % 1024 == 0, "chunk image must be whole 1-KB DMA pieces");
  __shared__ __attribute__((aligned(1024))) T lds[LDS_ELEMS];
  typedef __attribute__((address_space(3))) T lds_T;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int C = p.C;
  const long row0 = ((long)blockIdx.x * 4 + wave) * ROWS;
  const int nchunks = p.HID / 32;
  const bool full_rows = row0 + ROWS <= p.M;  // wave-uniform
  const int n_img = (NI - wave + 3) / 4;      // image DMA instructions this wave issues per chunk (q*4 + wave < NI)
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // LDS-DMA through vkas_lds_dma16 (vkas_common.h), not the compiler's builtin: with the builtin SIInsertWaitcnts put an
  // s_waitcnt vmcnt(0) in front of the first LDS read it considered aliasing (the fp32 bias read of the image), i.e. the
  // "prefetched" next chunk was waited for at the top of every chunk (round 4, profiles/isa_loops.py)
#ifdef CHAIN_BUILTIN_DMA
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.img, (short)0, (int)((long)nchunks * IMG * 2), 0x00020000);
#else
  const u32x4 rs_w = vkas_make_rsrc(p.img, (unsigned)((long)nchunks * IMG * 2));
  const unsigned lds0 = vkas_lds_addr(lds);
#endif
  auto issue_chunk = [&](int j, int buf) {
#pragma unroll
    for (int q = 0; q < (NI + 3) / 4; ++q) {
      const int inst = q * 4 + wave;
      if (inst >= NI) break;  // wave-uniform
      const unsigned voff = (unsigned)j * (unsigned)(IMG * 2) + (unsigned)inst * 1024u + (unsigned)lane * 16u;
#ifdef CHAIN_BUILTIN_DMA
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)((lds_T*)lds + buf * IMG + inst * 512), 16, voff, 0, 0, 0);
#else
      vkas_lds_dma16(rs_w, lds0 + (unsigned)(buf * IMG + inst * 512) * 2u, voff);
#endif
    }
  };
  issue_chunk(0, 0);

  // Pair staging buffer of this wave (the 4C-wide tensor moves in 128-byte row pieces = two chunks): row r, logical
  // 16-byte piece c sits at position c ^ ((r >> 1) & 7) (the 16 rows x 2 halves a b64 access touches hit 64 distinct
  // banks); a 1-KB DMA / flush instruction covers 8 rows, lane = (row & 7) * 8 + position.
  T* stg = lds + 2 * IMG + wave * STG;
  auto stg_off = [&](int r, int e) { return r * 64 + ((((e >> 3) ^ (r >> 1)) & 7) << 3) + (e & 7); };
  const int f_r = lane >> 3, f_pp = lane & 7;  // flush / DMA role: row f_r of a piece, position f_pp
  // backward: saved pre-activations of chunk pair P -> staging buffer (LDS-DMA, whole 128-byte row pieces)
#ifdef CHAIN_BUILTIN_DMA
  __amdgpu_buffer_rsrc_t rs_h;
  if constexpr (MODE == MODE_BWD)
    rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)p.mid_in, (short)0, (int)(((p.M - 1) * p.ldmi + p.HID) * 2), 0x00020000);
#else
  u32x4 rs_h = {0, 0, 0, 0};
  if constexpr (MODE == MODE_BWD) rs_h = vkas_make_rsrc(p.mid_in, (unsigned)(((p.M - 1) * p.ldmi + p.HID) * 2));
#endif
  auto issue_h = [&](int P) {
#pragma unroll
    for (int it = 0; it < NF; ++it) {
      const int r = it * 8 + f_r;
      const int c = (f_pp ^ (r >> 1)) & 7;
      const long m = row0 + r;
      const bool ok = m < p.M && P * 64 + c * 8 < p.HID && !(ABL & 8);
      const unsigned voff = ok ? (unsigned)((m * p.ldmi + P * 64 + c * 8) * 2) : OOB;
#ifdef CHAIN_BUILTIN_DMA
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (lds_void_ptr)((lds_T*)stg + it * 512), 16, voff, 0, 0, 0);
#else
      vkas_lds_dma16(rs_h, vkas_lds_addr(stg) + (unsigned)(it * 1024), voff);
#endif
    }
  };
  // staged pair -> global (h forward, dh backward): 8 rows x 128 B per instruction, whole cache lines
  auto flush_pair = [&](int P, int width) {
    T* G = reinterpret_cast<T*>(p.mid_out);
    if constexpr (HOIST) {
    // all pieces read first (one LDS round trip instead of NF: each read stood in front of its own store)
    v8 v[NF];
#pragma unroll
    for (int it = 0; it < NF; ++it) v[it] = *reinterpret_cast<const v8*>(stg + (it * 8 + f_r) * 64 + f_pp * 8);
#pragma unroll
    for (int it = 0; it < NF; ++it) {
      const int r = it * 8 + f_r;
      const int c = (f_pp ^ (r >> 1)) & 7;
      const long m = row0 + r;
      if (m < p.M && c * 8 < width && !(ABL & 1)) *reinterpret_cast<v8*>(G + m * p.ldm + P * 64 + c * 8) = v[it];
    }
    } else {
#pragma unroll 2
    for (int it = 0; it < NF; ++it) {
      const int r = it * 8 + f_r;
      const int c = (f_pp ^ (r >> 1)) & 7;
      const long m = row0 + r;
      const v8 v = *reinterpret_cast<const v8*>(stg + r * 64 + f_pp * 8);
      if (m < p.M && c * 8 < width && !(ABL & 1)) *reinterpret_cast<v8*>(G + m * p.ldm + P * 64 + c * 8) = v;
    }
    }
  };

  // input rows -> B fragments (row fr of group i, k = ks*32 + g*8 .. +7), resident for the whole kernel
  v8 xf[TM][KS];
  const T* A = reinterpret_cast<const T*>(p.a);
  float* lnp = reinterpret_cast<float*>(lds + 2 * IMG);  // the staging region is idle until the first chunk
  static_assert(4 * STG * (int)sizeof(T) >= 2 * 512 * 4, "LayerNorm parameters fit the staging region");
  const bool fused_ln = MODE == MODE_FWD && p.ln_gamma != nullptr;  // workgroup-uniform
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = row0 + i * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + g * 8;
      v8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < p.M && k < C) v = *reinterpret_cast<const v8*>(A + m * p.lda + k);
      xf[i][ks] = v;
    }
  }
  if constexpr (MODE == MODE_FWD) {
    if (fused_ln) {
      chain_layernorm_stage<256>(p, lnp, tid);
      __syncthreads();
      chain_layernorm_rows<T, TM, KS>(xf, p, [&](int i) { return row0 + i * 16 + fr; }, g, lnp);
    }
  }
  f32x4 acc[TM][NT2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int n = 0; n < NT2; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE == MODE_BWD) issue_h(0);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int j = 0; j < nchunks; ++j) {
    const int buf = j & 1, par = j & 1;
    const bool more = j + 1 < nchunks;
    if (more) issue_chunk(j + 1, buf ^ 1);
    const T* Wa = lds + buf * IMG;
    const T* Wb = Wa + 32 * KA;
    const float* Ba = reinterpret_cast<const float*>(Wb + KS * 1024);
    // ---- GEMM-a: d[i][t] (lane: pixel fr of group i, hidden units t*16 + 4g .. +3 of this chunk)
    f32x4 d[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      d[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      d[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (HOIST) {
    {
      // every weight fragment of the chunk's GEMM-a is requested before the first product (the scheduler otherwise pairs each
      // read with its use: one LDS latency per k step in front of 2 TM products)
      constexpr int KB = KS < 8 ? KS : KS / 2;   // k steps per batch (KS >= 8: two batches keep the fragments within 64 registers)
#pragma unroll
      for (int k0 = 0; k0 < KS; k0 += KB) {
        v8 wa[KB][2];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
          const int ks = k0 + kk;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int h = t * 16 + fr;
            const int c = (ks & 1) * 4 + g;
            wa[kk][t] = *reinterpret_cast<const v8*>(Wa + (ks >> 1) * 2048 + h * 64 + ((c ^ (h & 7)) << 3));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            d[i][0] = mfma16(wa[kk][0], xf[i][k0 + kk], d[i][0]);
            d[i][1] = mfma16(wa[kk][1], xf[i][k0 + kk], d[i][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      v8 wa[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int h = t * 16 + fr;
        const int c = (ks & 1) * 4 + g;
        wa[t] = *reinterpret_cast<const v8*>(Wa + (ks >> 1) * 2048 + h * 64 + ((c ^ (h & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        d[i][0] = mfma16(wa[0], xf[i][ks], d[i][0]);
        d[i][1] = mfma16(wa[1], xf[i][ks], d[i][1]);
      }
    }
    }
    // ---- elementwise middle: the GEMM-b operand of this chunk, built in registers; the 4C-wide values go through the
    //      pair staging buffer (this lane: hidden units e0 .. e0+3 and e0+16 .. e0+19 of the pair's 64)
    v8 gf[TM];
    const int e0 = par * 32 + g * 4;
    if constexpr (MODE == MODE_FWD) {
      const float4 b0 = *reinterpret_cast<const float4*>(Ba + g * 4);       // bias from the LDS image: a VMEM load here
      const float4 b1 = *reinterpret_cast<const float4*>(Ba + g * 4 + 16);  // would make the compiler wait for the stores
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = i * 16 + fr;
        float v[8] = {d[i][0][0] + b0.x, d[i][0][1] + b0.y, d[i][0][2] + b0.z, d[i][0][3] + b0.w,
                      d[i][1][0] + b1.x, d[i][1][1] + b1.y, d[i][1][2] + b1.z, d[i][1][3] + b1.w};
        v4 h0, h1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h0[q] = (T)v[q];
          h1[q] = (T)v[4 + q];
        }
        *reinterpret_cast<v4*>(stg + stg_off(r, e0)) = h0;
        *reinterpret_cast<v4*>(stg + stg_off(r, e0 + 16)) = h1;
        // GELU of the stored (rounded) pre-activation: what backward will differentiate
        const v4 g0 = (ABL & 4) ? h0 : chain_gelu4<T, v4>(h0), g1 = (ABL & 4) ? h1 : chain_gelu4<T, v4>(h1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gf[i][q] = g0[q];
          gf[i][4 + q] = g1[q];
        }
      }
    } else {
      if (par == 0 && j > 0) {
        // the pair's h (DMA issued in the previous chunk, before its GEMM-b) must have landed; only this chunk's image
        // DMA is younger
        if (!full_rows || !more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n_img == (NI + 3) / 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI + 3) / 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI + 3) / 4 - 1) : "memory");
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = i * 16 + fr;
        const v4 h0 = *reinterpret_cast<const v4*>(stg + stg_off(r, e0));
        const v4 h1 = *reinterpret_cast<const v4*>(stg + stg_off(r, e0 + 16));
        v4 o0, o1;
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o0[q] = (T)(d[i][0][q] * (float)h0[q]);
            o1[q] = (T)(d[i][1][q] * (float)h1[q]);
          }
        } else {
          o0 = chain_dgelu4<T, v4>(h0, d[i][0]);
          o1 = chain_dgelu4<T, v4>(h1, d[i][1]);
        }
        *reinterpret_cast<v4*>(stg + stg_off(r, e0)) = o0;  // in place: dh over h
        *reinterpret_cast<v4*>(stg + stg_off(r, e0 + 16)) = o1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gf[i][q] = o0[q];
          gf[i][4 + q] = o1[q];
        }
      }
    }
    // (forward without a destination for h = inference: nothing is kept for a backward pass)
    const bool flush_now = (par == 1 || !more) && (MODE == MODE_BWD || p.mid_out != nullptr);
    if (flush_now) {
      flush_pair(j >> 1, par == 1 ? 64 : 32);
      if constexpr (MODE == MODE_BWD) {
        if (more) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the flush has read the buffer
          issue_h((j + 1) >> 1);
        }
      }
    }
    // ---- GEMM-b: acc[i][n] += Wb[n rows][32 k slots] . gf[i]
    if constexpr (HOIST) {
    {
      constexpr int NBB = NT2 <= 12 ? NT2 : 8;  // fragments per batch
#pragma unroll
      for (int n0 = 0; n0 < NT2; n0 += NBB) {
        v8 wb[NBB];
#pragma unroll
        for (int u = 0; u < NBB; ++u) {
          const int n = n0 + u;
          if (n >= NT2) break;
          const int row = n * 16 + fr;
          const int R = row >> 1, c = (row & 1) * 4 + g;
          wb[u] = *reinterpret_cast<const v8*>(Wb + R * 64 + ((c ^ (R & 7)) << 3));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NBB; ++u) {
          const int n = n0 + u;
          if (n >= NT2) break;
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][n] = mfma16(wb[u], gf[i], acc[i][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    } else {
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
      const int row = n * 16 + fr;
      const int R = row >> 1, c = (row & 1) * 4 + g;
      const v8 wb = *reinterpret_cast<const v8*>(Wb + R * 64 + ((c ^ (R & 7)) << 3));
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][n] = mfma16(wb, gf[i], acc[i][n]);
    }
    }
    // The next chunk's image (issued at the top of this iteration) must have landed.  What this chunk issued after it
    // - the flush stores and, backward, the next pair's h DMA - is younger and stays in flight: a full drain (which
    // __syncthreads() implies) would put the HBM store latency on the critical path of every chunk.  A wave with rows
    // beyond M may have skipped stores, so its count of younger operations is unknown: it drains.
    if (more) {
      if (!full_rows || !flush_now || ABL != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (MODE == MODE_FWD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NF) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue.  The accumulators (lane: pixel fr of group i, channels n*16 + 4g .. +3) go through a per-wave LDS tile
  // (the weight images are dead) so that every global access is a 16-byte piece of a whole row: lane = piece of row
  // (it * RPI + lane / PCS), PCS = C / 8 pieces per row, RPI rows per instruction.
  if ((ABL & 2) && acc[0][0][0] != 1.2345f) return;
  T* ep = lds + wave * (ROWS * PE);
#pragma unroll
  for (int n = 0; n < NT2; ++n) {
    const int col = n * 16 + g * 4;
    float4 b2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == MODE_FWD) {
      if (col < C) b2 = *reinterpret_cast<const float4*>(p.bias_b + col);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      v4 zq;
      zq[0] = (T)(acc[i][n][0] + b2.x);
      zq[1] = (T)(acc[i][n][1] + b2.y);
      zq[2] = (T)(acc[i][n][2] + b2.z);
      zq[3] = (T)(acc[i][n][3] + b2.w);
      *reinterpret_cast<v4*>(ep + (i * 16 + fr) * PE + col) = zq;
    }
  }
  const int PCS = C >> 3;
  const int RPI = 64 / PCS;
  const int piece = lane % PCS, rsub = lane / PCS;
  float cs[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) cs[q] = 0.f;
  if constexpr (MODE == MODE_FWD) load8(p.colscale + piece * 8, cs);
  // image of this lane's current row (rowscale index), advanced without divisions
  long m = row0 + rsub;
  int img = 0, rim = 0;
  if constexpr (MODE == MODE_FWD) {
    img = (int)(m / p.rows_per_image);
    rim = (int)(m - (long)img * p.rows_per_image);
  }
  // Forward, C <= 256: at most 16 row pieces per lane.  Their residual rows are requested here, all at once (the accumulators
  // are dead: registers are free), instead of one global-load latency per piece inside the loop below.
  constexpr int NPRE = (MODE == MODE_FWD && KS <= 8) ? 16 : 1;
  Raw8<T> xres[NPRE];
  if constexpr (NPRE > 1) {
#pragma unroll
    for (int it = 0; it < NPRE; ++it) {
      const int r = rsub + it * RPI;
      const long mm = row0 + r;
      xres[it].zero();
      if (rsub < RPI && r < ROWS && mm < p.M) xres[it].load(reinterpret_cast<const T*>(p.res) + mm * p.ldres + piece * 8);
    }
#pragma unroll
    for (int it = 0; it < NPRE; ++it) {
      const int r = rsub + it * RPI;
      if (rsub < RPI && r < ROWS) {
        if (m < p.M) {
          const v8 zq = *reinterpret_cast<const v8*>(ep + r * PE + piece * 8);
          if (p.z) *reinterpret_cast<v8*>(reinterpret_cast<T*>(p.z) + m * p.ldz + piece * 8) = zq;
          const float rs = p.rowscale ? p.rowscale[img] : 1.0f;
          float xr[8];
          xres[it].unpack(xr);
#pragma unroll
          for (int q = 0; q < 8; ++q) xr[q] += rs * cs[q] * (float)zq[q];
          store8(reinterpret_cast<T*>(p.out) + m * p.ldo + piece * 8, xr);
        }
        m += RPI;
        rim += RPI;
        while (rim >= p.rows_per_image) {
          rim -= p.rows_per_image;
          ++img;
        }
      }
    }
    return;
  }
  if (rsub < RPI) {
    for (int r = rsub; r < ROWS; r += RPI) {
      if (m < p.M) {
        const v8 zq = *reinterpret_cast<const v8*>(ep + r * PE + piece * 8);
        if constexpr (MODE == MODE_FWD) {
          // z is stored in the activation type and the residual uses the rounded value (as the two-kernel path does)
          if (p.z) *reinterpret_cast<v8*>(reinterpret_cast<T*>(p.z) + m * p.ldz + piece * 8) = zq;
          const float rs = p.rowscale ? p.rowscale[img] : 1.0f;
          float xr[8];
          load8(reinterpret_cast<const T*>(p.res) + m * p.ldres + piece * 8, xr);
#pragma unroll
          for (int q = 0; q < 8; ++q) xr[q] += rs * cs[q] * (float)zq[q];
          store8(reinterpret_cast<T*>(p.out) + m * p.ldo + piece * 8, xr);
        } else {
          *reinterpret_cast<v8*>(reinterpret_cast<T*>(p.out) + m * p.ldo + piece * 8) = zq;
        }
      }
      m += RPI;
      if constexpr (MODE == MODE_FWD) {
        rim += RPI;
        while (rim >= p.rows_per_image) {
          rim -= p.rows_per_image;
          ++img;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair-split chain for 256 < C <= 384 (stage 2 of ConvNeXt-T / -S: 9 / 27 of the backbone's layers), round 4.
//
// At C = 384 the kernel above needs 192 accumulator + 96 input registers per wave, i.e. one 512-register wave per SIMD:
// nothing overlaps its GELU arithmetic, LDS fragment latency and staging with the matrix pipe, and it measured no faster than
// the two GEMMs it replaces (0.31 ms per layer and direction).  This variant keeps the chain's data flow but splits a
// 32-row group over a PAIR of waves so that a wave fits 256 registers and every SIMD holds two waves:
//   * wave t (0 / 1) of a pair computes GEMM-a for hidden units 16 t .. 16 t + 15 of every 32-unit chunk (its 32 input rows
//     stay in registers as B fragments: 96 registers) and GEMM-b for output columns 192 t .. 192 t + 191 (96 accumulator
//     registers).  GEMM-b's K = 32 fragment needs both halves of the chunk's activation: each wave hands its 4 values per
//     lane and row group to its partner through LDS (8 bytes per lane: the D layout of GEMM-a IS the B layout of GEMM-b under
//     the k permutation baked into the weight image, so the exchange is lane to lane).  No product is computed twice.
//   * 8 waves = 4 pairs = 128 rows per workgroup; the two wave groups (waves 0-3 / 4-7: one wave of each on every SIMD) run
//     half an iteration apart, as in conv3x3_slab_mfma_kernel: per chunk a wave has an MFMA phase (GEMM-b of the previous
//     chunk, GEMM-a of this one: 48 MFMAs, 24 fragment reads) and a VALU phase (bias, rounding, GELU, staging, exchange,
//     LDS-DMA requests), each closed by a barrier; while one group multiplies the other one does its vector work.
//   * weights: the packed chunk images of the kernel above, unchanged; GEMM-a tiles and GEMM-b tiles in two rings of two slots
//     (a slot is refilled by LDS-DMA as soon as both groups have read it, one and a half to two intervals before its next use).
// LDS: 2 x 24 KB + 2 x 24 KB weight rings, 32 KB staging of the 4C-wide tensor (128-byte row pieces, per pair, two chunk
// pairs), 16 KB exchange, bias pieces: 145 KB.
// Phase timestamps of mlp_chain_pair_kernel for profiles/trace_chain.py (-DCHAIN_TRACE builds only, never shipped): workgroup
// 0, waves 0 (group 0) and 4 (group 1), chunks 8..15: s_memtime at the top of the MFMA phase, behind its last product, behind
// its waits, behind its barrier, at the end of the VALU phase, behind that barrier.
// which phase of mlp_chain_pair_kernel runs at raised wave priority: 0 none, 1 the MFMA phase, 2 the VALU phase
#ifndef CHAIN_PRIO
#define CHAIN_PRIO 1
#endif
#ifdef CHAIN_TRACE
__device__ unsigned long long vkas_chain_trace_buf[2 * 8 * 8];
#define CHAIN_TR(slot)                                                                                          \
  if (blockIdx.x == 0 && (tid & 255) == 0 && j >= 8 && j < 16)                                                   \
  vkas_chain_trace_buf[((tid >> 8) * 8 + (j - 8)) * 8 + (slot)] = __builtin_readcyclecounter()
#define CHAIN_TR_SB(slot)                  \
  __builtin_amdgcn_sched_barrier(0);       \
  CHAIN_TR(slot);                          \
  __builtin_amdgcn_sched_barrier(0)
#else
#define CHAIN_TR(slot)
#define CHAIN_TR_SB(slot)
#endif

template <typename T, int KS, int TM, int MODE>
__global__ __launch_bounds__(512, 2) void mlp_chain_pair_kernel(ChainArgs p) {
  // TM = 16-row groups per wave in GEMM-a (a pair owns 2 TM groups = 32 TM rows, a workgroup 128 TM rows).  Shipped: KS = 12,
  // TM = 1 (256 < C <= 384).  KS = 6, TM = 2 (128 < C <= 192: the same 48 + 96 resident registers, every GEMM-a fragment
  // feeding two products and every GEMM-b fragment four) was built, is correct and measured SLOWER than mlp_chain_kernel<6, 2>
  // at stage 1 of config #3 (0.35 / 0.46 ms forward / backward against 0.33 / 0.32): profiles/experiments/README.md.
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  static_assert(KS % 2 == 0, "the pair splits the 2 KS output column tiles evenly");
  constexpr int RP = 32 * TM;                    // rows of a pair
  constexpr int RG = 2 * TM;                     // 16-row groups of a pair
  constexpr int KA = chain_ka(KS);
  constexpr int IMG = chain_img_elems(KS);       // elements per packed chunk image (GEMM-a tile | GEMM-b tile | bias piece)
  constexpr int AEL = 32 * KA, BEL = KS * 1024;  // elements of the two tiles
  constexpr int NIA = AEL * 2 / 1024, NIB = BEL * 2 / 1024;  // 1-KB DMA instructions per tile
  constexpr int OFF_A = 0, OFF_B = 2 * AEL, OFF_BIAS = OFF_B + 2 * BEL;       // element offsets (T) into the LDS array
  constexpr int OFF_EX = OFF_BIAS + 4 * 64;                                    // bias: 4 slots x 32 floats (a slot is refilled two chunks ahead, while the chunk before it is still being read)
  constexpr int OFF_STG = OFF_EX + 2 * 4 * 2 * TM * 64 * 8;                    // exchange: [parity][pair][t][row group][lane] x 8 elements
  constexpr int LDS_MAIN = OFF_STG + 4 * 2 * RP * 64;                          // staging: [pair][parity][RP rows][64]
  constexpr int PE = KS * 16 + 8;                                              // epilogue row pitch: this wave's 16 KS columns
  constexpr int LDS_EPI = 8 * RP * PE;
  constexpr int LDS_ELEMS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(LDS_ELEMS * 2 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) T lds[LDS_ELEMS];
  typedef __attribute__((address_space(3))) T lds_T;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;            // wave group: the two groups run half an iteration apart
  const int pair = wave >> 1;           // 0..3: RP rows each
  const int t = wave & 1;               // half of the pair: row groups t TM .. in GEMM-a, output columns 16 KS t .. in GEMM-b
  const int wq = wave & 3;              // index inside the group (DMA / store shares: pair wq)
  const int fr = lane & 15, g = lane >> 4;
  const int C = p.C;
  const long row0 = (long)blockIdx.x * (4 * RP) + pair * RP;
  const int nchunks = p.HID / 32;
  const bool keep_mid = MODE == MODE_BWD || p.mid_out != nullptr;
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // every LDS-DMA request of this kernel goes through vkas_lds_dma16 (vkas_common.h): the waits are the kernel's own
  const u32x4 rs_w = vkas_make_rsrc(p.img, (unsigned)((long)nchunks * IMG * 2));
  const unsigned lds0 = vkas_lds_addr(lds);  // LDS byte address of element 0
  // tile DMA: NI 1-KB instructions starting at byte `src` of the image buffer -> LDS element offset `dst`, dealt over `nw` waves
  auto issue_tile = [&](unsigned src, int dst, int ni, int w, int nw) {
    for (int inst = w; inst < ni; inst += nw) {  // wave-uniform trip count
      const unsigned voff = src + (unsigned)inst * 1024u + (unsigned)lane * 16u;
      vkas_lds_dma16(rs_w, lds0 + (unsigned)(dst + inst * 512) * 2u, voff);
    }
  };
  auto issue_a = [&](int j, int w, int nw) {  // GEMM-a tile + the chunk's 32 biases (first 128 bytes of the bias piece)
    const unsigned base = (unsigned)j * (unsigned)(IMG * 2);
    issue_tile(base, OFF_A + (j & 1) * AEL, NIA, w, nw);
    if (w == 0 && lane < 8)
      vkas_lds_dma16(rs_w, lds0 + (unsigned)(OFF_BIAS + (j & 3) * 64) * 2u, base + (unsigned)((AEL + BEL) * 2) + (unsigned)lane * 16u);
  };
  auto issue_b = [&](int j, int w, int nw) { issue_tile((unsigned)j * (unsigned)(IMG * 2) + AEL * 2, OFF_B + (j & 1) * BEL, NIB, w, nw); };

  // staging of the 4C-wide tensor, per pair and chunk-pair parity: RP rows x 128 bytes, 16-byte piece c of row r at position
  // c ^ ((r >> 1) & 7) (as in the kernel above)
  auto stg_base_of = [&](int pr, int par2) { return lds + OFF_STG + (pr * 2 + par2) * (RP * 64); };
  auto stg_base = [&](int par2) { return stg_base_of(pair, par2); };
  auto stg_off = [&](int r, int e) { return r * 64 + ((((e >> 3) ^ (r >> 1)) & 7) << 3) + (e & 7); };
  const int f_r = lane >> 3, f_pp = lane & 7;
  u32x4 rs_h = {0, 0, 0, 0};
  if constexpr (MODE == MODE_BWD) rs_h = vkas_make_rsrc(p.mid_in, (unsigned)(((p.M - 1) * p.ldmi + p.HID) * 2));
  // Division of the memory work: group 1 issues every LDS-DMA request (weight tiles, backward: h) at the start of its VALU
  // phase and waits for them at the end of its NEXT MFMA phase - a whole phase of slack, and still one barrier before the first
  // reader; group 0 issues every store of the 4C-wide tensor and never waits for vector memory inside the loop, so a store's
  // acknowledgement (microseconds) is on nobody's critical path.
  constexpr int NFP = RP / 8;  // 1-KB pieces (8 rows x 128 B) of a pair's staging buffer
  auto issue_h = [&](int P2) {  // backward: saved pre-activations of chunk pair P2 -> staging, all four pairs (wave wq: pair wq)
    T* dst = stg_base_of(wq, P2 & 1);
    const long prow0 = (long)blockIdx.x * (4 * RP) + wq * RP;
#pragma unroll
    for (int it = 0; it < NFP; ++it) {
      const int r = it * 8 + f_r;
      const int c = (f_pp ^ (r >> 1)) & 7;
      const long m = prow0 + r;
      const bool ok = m < p.M && P2 * 64 + c * 8 < p.HID;
      const unsigned voff = ok ? (unsigned)((m * p.ldmi + P2 * 64 + c * 8) * 2) : OOB;
      vkas_lds_dma16(rs_h, vkas_lds_addr(dst) + (unsigned)(it * 1024), voff);
    }
  };
  auto flush_pair = [&](int P2, int width) {  // staged chunk pair of pair wq -> global (h forward, dh backward)
    T* G = reinterpret_cast<T*>(p.mid_out);
    const T* src = stg_base_of(wq, P2 & 1);
    const long prow0 = (long)blockIdx.x * (4 * RP) + wq * RP;
#pragma unroll
    for (int i0 = 0; i0 < NFP; i0 += 4) {  // four pieces per LDS round trip
      v8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v8*>(src + ((i0 + u) * 8 + f_r) * 64 + f_pp * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = (i0 + u) * 8 + f_r;
        const int c = (f_pp ^ (r >> 1)) & 7;
        const long m = prow0 + r;
        if (m < p.M && c * 8 < width) *reinterpret_cast<v8*>(G + m * p.ldm + P2 * 64 + c * 8) = v[u];
      }
    }
  };

  // prologue: the first two GEMM-a tiles, the first GEMM-b tile, (backward) the first two chunk pairs of h
  issue_a(0, wave, 8);
  issue_b(0, wave, 8);
  if (nchunks > 1) issue_a(1, wave, 8);
  if constexpr (MODE == MODE_BWD) {
    if (grp == 1) {
      issue_h(0);
      if (nchunks > 2) issue_h(1);
    }
  }
  // this wave's 16 TM input rows (row groups t TM .. t TM + TM - 1 of the pair) -> B fragments (row fr of group i,
  // k = ks*32 + g*8 .. +7), resident for the whole kernel
  v8 xf[TM][KS];
  {
    const T* A = reinterpret_cast<const T*>(p.a);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long m = row0 + (t * TM + i) * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 32 + g * 8;
        v8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < p.M && k < C) v = *reinterpret_cast<const v8*>(A + m * p.lda + k);
        xf[i][ks] = v;
      }
    }
  }
  if constexpr (MODE == MODE_FWD) {
    if (p.ln_gamma) {  // workgroup-uniform; the staging region is idle until the first VALU phase
      float* lnp = reinterpret_cast<float*>(lds + OFF_STG);
      chain_layernorm_stage<512>(p, lnp, tid);
      __syncthreads();
      chain_layernorm_rows<T, TM, KS>(xf, p, [&](int i) { return row0 + (t * TM + i) * 16 + fr; }, g, lnp);
    }
  }
  f32x4 acc[RG][KS];
#pragma unroll
  for (int i = 0; i < RG; ++i)
#pragma unroll
    for (int n = 0; n < KS; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one interval behind group 0

  f32x4 d[TM][2];  // GEMM-a result of row group i: hidden units 4g .. 4g+3 (d[i][0]) and 16 + 4g .. (d[i][1]) of the chunk
  v8 own[TM];      // ... after the elementwise middle: exactly one K = 32 B fragment of GEMM-b (k permutation of the weight image)
  v4 hq[TM][2];    // backward: the saved pre-activations of this lane's 8 hidden units per row group, read in the MFMA phase
  T* exch = lds + OFF_EX;
  auto exch_at = [&](int par, int tt, int i) { return exch + ((((par * 4 + pair) * 2 + tt) * TM + i) * 512) + lane * 8; };
  // One MFMA phase = GEMM-b of chunk jb (KS weight fragments, 2 TM products each; skipped for jb < 0) and GEMM-a of chunk ja
  // (2 KS fragments, TM products each; skipped for ja < 0), fragments in batches of FB: the reads of a batch are issued before
  // the products of the batch in front of it, so only the first batch's LDS latency is exposed (measured: a batch whose reads
  // stand directly in front of its own products costs 200 - 300 cycles of LDS latency per batch with every wave of the group
  // reading at once).  Forward: the chunk's biases are read with the first batch and become GEMM-a's initial accumulators.
  constexpr int FB = TM == 1 ? 6 : (MODE == MODE_BWD ? 3 : 4);  // (32 rows per wave: 16 more registers hold GEMM-b operands - and, backward, the saved pre-activations -, so smaller batches)
  const int b_lane = (fr >> 1) * 64 + ((((fr & 1) * 4 + g) ^ ((fr >> 1) & 7)) << 3);
  const int a_lane[2] = {fr * 64 + ((g ^ (fr & 7)) << 3), fr * 64 + (((4 + g) ^ (fr & 7)) << 3)};
  constexpr int NB_B = (KS + FB - 1) / FB, NB_A = (2 * KS) / FB;  // batches of GEMM-b / GEMM-a
  static_assert((2 * KS) % FB == 0, "GEMM-a fragments in whole batches");
  auto mfma_phase = [&](auto has_b_c, auto has_a_c, int jb, int ja) __attribute__((always_inline)) {
    constexpr bool HAS_B = decltype(has_b_c)::value, HAS_A = decltype(has_a_c)::value;  // compile-time: the buffer slots are too
    v8 fr_[2][FB];
    v8 gf[RG];
    const T* Wb = lds + OFF_B + (jb & 1) * BEL;
    const T* Wa = lds + OFF_A + (ja & 1) * AEL;
    auto read_b = [&](int bb, int slot) {
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int n = bb * FB + u;
        if (n >= KS) break;
        // row (t KS + n) 16 + fr of the [C / 2][64] view: R = row >> 1 = (t KS + n) 8 + (fr >> 1), R & 7 = fr >> 1, so the
        // swizzled position is a per-lane constant and the fragment address is lane base + 512 n elements (an immediate)
        if ((ABL & 32) && n > 0) fr_[slot][u] = xf[0][0];  // timing only: one fragment read per phase
        else fr_[slot][u] = *reinterpret_cast<const v8*>(Wb + (t * KS + n) * 512 + b_lane);
      }
    };
    auto read_a = [&](int ab, int slot) {  // fragment f of GEMM-a: k step f >> 1, hidden half f & 1
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int f = ab * FB + u;
        const int ks = f >> 1, tt = f & 1;
        // hidden row h = 16 tt + fr of sub-tile ks >> 1: (h & 7) = fr & 7, so two per-lane constants (k-step parity) + immediates
        if ((ABL & 32) && f > 0) fr_[slot][u] = xf[0][0];
        else fr_[slot][u] = *reinterpret_cast<const v8*>(Wa + (ks >> 1) * 2048 + tt * 1024 + a_lane[ks & 1]);
      }
    };
    auto mul_b = [&](int bb, int slot) {
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int n = bb * FB + u;
        if (n >= KS) break;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg][n] = mfma16(fr_[slot][u], gf[rg], acc[rg][n]);
      }
    };
    auto mul_a = [&](int ab, int slot) {
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int f = ab * FB + u;
#pragma unroll
        for (int i = 0; i < TM; ++i) d[i][f & 1] = mfma16(fr_[slot][u], xf[i][f >> 1], d[i][f & 1]);
      }
    };
    if constexpr (HAS_B) {
      read_b(0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {  // row groups t TM + i are this wave's own, the others the partner's
        const v8 other = *reinterpret_cast<const v8*>(exch_at(jb & 1, 1 - t, i));
        gf[i] = t == 0 ? own[i] : other;
        gf[TM + i] = t == 0 ? other : own[i];
      }
    }
    if constexpr (HAS_A) {
      if constexpr (MODE == MODE_FWD) {
        const float* Ba = reinterpret_cast<const float*>(lds + OFF_BIAS + (ja & 3) * 64);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Ba + g * 4), b1 = *reinterpret_cast<const f32x4*>(Ba + 16 + g * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          d[i][0] = b0;
          d[i][1] = b1;
        }
      } else {
        const int e0 = (ja & 1) * 32 + g * 4;
        const T* stg = stg_base((ja >> 1) & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          d[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          d[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
          hq[i][0] = *reinterpret_cast<const v4*>(stg + stg_off((t * TM + i) * 16 + fr, e0));
          hq[i][1] = *reinterpret_cast<const v4*>(stg + stg_off((t * TM + i) * 16 + fr, e0 + 16));
        }
      }
      if constexpr (!HAS_B) read_a(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int S0 = HAS_B ? NB_B : 0;  // batches in front of GEMM-a's first one: its buffer slot is S0 & 1
    if constexpr (HAS_B) {
#pragma unroll
      for (int bb = 0; bb < NB_B; ++bb) {
        // the next batch's reads (GEMM-b's next batch, or GEMM-a's first one) in front of this batch's products
        if (bb + 1 < NB_B) read_b(bb + 1, (bb + 1) & 1);
        else if constexpr (HAS_A) read_a(0, S0 & 1);
        __builtin_amdgcn_sched_barrier(0);
        mul_b(bb, bb & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (HAS_A) {
#pragma unroll
      for (int ab = 0; ab < NB_A; ++ab) {
        if (ab + 1 < NB_A) read_a(ab + 1, (S0 + ab + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mul_a(ab, (S0 + ab) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // one chunk = an MFMA phase and a VALU phase, each closed by a barrier; the first chunk (no GEMM-b yet) is peeled off the
  // loop so that the loop body has one shape (with both shapes behind a branch inside the loop the register allocator kept
  // two copies of the accumulators: 135 spilled registers)
  auto chunk = [&](auto no_b_c, int j) __attribute__((always_inline)) {
    constexpr bool NO_B = decltype(no_b_c)::value;
    std::integral_constant<bool, !NO_B> first_c;  // "has GEMM-b"
    CHAIN_TR(0);
    // ---------------- MFMA phase: GEMM-b of chunk j - 1, GEMM-a of chunk j
    if constexpr (CHAIN_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    mfma_phase(first_c, std::true_type{}, j - 1, j);
    if constexpr (CHAIN_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    CHAIN_TR(1);
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests of its previous VALU phase have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CHAIN_TR(2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    CHAIN_TR(3);
    if constexpr (CHAIN_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    // ---------------- VALU phase of chunk j
    // Group 1 (this is interval 2 j + 2): GEMM-a tile of chunk j + 2 (its slot was last read by this group's MFMA phase that
    // just ended; first read by group 0 in interval 2 j + 4) and GEMM-b tile of chunk j + 1 (likewise); waited for at the end
    // of this group's next MFMA phase (interval 2 j + 3).
    if (grp == 1) {
      if (!(ABL & 16)) {
        if (j + 2 < nchunks) issue_a(j + 2, wq, 4);
        if (j + 1 < nchunks) issue_b(j + 1, wq, 4);
      }
      if constexpr (MODE == MODE_BWD) {
        // h of chunk pair Q + 2 (Q = j / 2 - 1) into the buffer of pair Q, which group 0 flushed in the interval that just
        // ended; landed (this group's wait at the end of its next MFMA phase) two intervals before its first reader
        if ((j & 1) == 0 && j >= 2 && (j >> 1) + 1 < (nchunks + 1) / 2 && !(ABL & 1)) issue_h((j >> 1) + 1);
      }
    } else if (keep_mid && !(ABL & 1) && (j & 1) == 0 && j >= 2) {
      // Group 0 (interval 2 j + 1): chunk pair (j >> 1) - 1 of all four pairs is complete (group 1 wrote its last part in
      // interval 2 j - 2); nothing waits for the stores: they stay in flight
      flush_pair((j >> 1) - 1, 64);
    }
    {
      const int e0 = (j & 1) * 32 + g * 4;   // this lane's hidden units e0 .. e0+3 and e0+16 .. e0+19 of the chunk pair's 64
      T* stg = stg_base((j >> 1) & 1);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (t * TM + i) * 16 + fr;  // its row inside the pair
        v4 h0, h1;
        if constexpr (MODE == MODE_FWD) {
          h0[0] = (T)d[i][0][0]; h0[1] = (T)d[i][0][1]; h0[2] = (T)d[i][0][2]; h0[3] = (T)d[i][0][3];  // (accumulators started from the bias)
          h1[0] = (T)d[i][1][0]; h1[1] = (T)d[i][1][1]; h1[2] = (T)d[i][1][2]; h1[3] = (T)d[i][1][3];
          if (keep_mid) {
            *reinterpret_cast<v4*>(stg + stg_off(r, e0)) = h0;
            *reinterpret_cast<v4*>(stg + stg_off(r, e0 + 16)) = h1;
          }
          // GELU of the stored (rounded) pre-activation: what backward will differentiate
          const v4 g0 = (ABL & 4) ? h0 : chain_gelu4<T, v4>(h0), g1 = (ABL & 4) ? h1 : chain_gelu4<T, v4>(h1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            own[i][q] = g0[q];
            own[i][4 + q] = g1[q];
          }
        } else {
          h0 = hq[i][0];
          h1 = hq[i][1];
          v4 o0, o1;
          if constexpr ((ABL & 4) != 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              o0[q] = (T)(d[i][0][q] * (float)h0[q]);
              o1[q] = (T)(d[i][1][q] * (float)h1[q]);
            }
          } else {
            o0 = chain_dgelu4<T, v4>(h0, d[i][0]);
            o1 = chain_dgelu4<T, v4>(h1, d[i][1]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            own[i][q] = o0[q];
            own[i][4 + q] = o1[q];
          }
          *reinterpret_cast<v4*>(stg + stg_off(r, e0)) = o0;  // in place: dh over h
          *reinterpret_cast<v4*>(stg + stg_off(r, e0 + 16)) = o1;
        }
        *reinterpret_cast<v8*>(exch_at(j & 1, t, i)) = own[i];
      }
    }
    if constexpr (CHAIN_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CHAIN_TR(4);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    CHAIN_TR(5);
  };
  chunk(std::true_type{}, 0);
  for (int j = 1; j < nchunks; ++j) chunk(std::false_type{}, j);
  // last GEMM-b; then group 0 (one barrier ahead: group 1's last VALU phase has ended by then) stores the last chunk pair
  mfma_phase(std::true_type{}, std::false_type{}, nchunks - 1, -1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();  // every wave passes the same number of barriers
    asm volatile("" ::: "memory");
    if (keep_mid && !(ABL & 1)) flush_pair((nchunks - 1) >> 1, (nchunks & 1) ? 32 : 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();    // nobody reads the weight rings or the staging buffers any more
  asm volatile("" ::: "memory");

  // ---- epilogue: the pair's RP rows x this wave's 16 KS columns through a per-wave LDS tile, then whole 16-byte row pieces
  T* ep = lds + wave * (RP * PE);
  const int col0 = t * KS * 16;
#pragma unroll
  for (int n = 0; n < KS; ++n) {
    const int col = n * 16 + g * 4;
    float4 b2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == MODE_FWD) {
      if (col0 + col < C) b2 = *reinterpret_cast<const float4*>(p.bias_b + col0 + col);
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      v4 zq;
      zq[0] = (T)(acc[i][n][0] + b2.x);
      zq[1] = (T)(acc[i][n][1] + b2.y);
      zq[2] = (T)(acc[i][n][2] + b2.z);
      zq[3] = (T)(acc[i][n][3] + b2.w);
      *reinterpret_cast<v4*>(ep + (i * 16 + fr) * PE + col) = zq;
    }
  }
  constexpr int PCS = KS * 2;        // 16-byte pieces of this wave's part of a row
  constexpr int RPI = 64 / PCS;      // rows per instruction
  const int piece = lane % PCS, rsub = lane / PCS;
  const int gcol = col0 + piece * 8;
  if (rsub < RPI && gcol < C) {
    float cs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cs[q] = 0.f;
    if constexpr (MODE == MODE_FWD) load8(p.colscale + gcol, cs);
#pragma unroll 4
    for (int r = rsub; r < RP; r += RPI) {
      const long m = row0 + r;
      if (m >= p.M) break;
      const v8 zq = *reinterpret_cast<const v8*>(ep + r * PE + piece * 8);
      if constexpr (MODE == MODE_FWD) {
        // z is stored in the activation type and the residual uses the rounded value (as the two-kernel path does)
        if (p.z) *reinterpret_cast<v8*>(reinterpret_cast<T*>(p.z) + m * p.ldz + gcol) = zq;
        const float rs = p.rowscale ? p.rowscale[(int)(m / p.rows_per_image)] : 1.0f;
        float xr[8];
        load8(reinterpret_cast<const T*>(p.res) + m * p.ldres + gcol, xr);
#pragma unroll
        for (int q = 0; q < 8; ++q) xr[q] += rs * cs[q] * (float)zq[q];
        store8(reinterpret_cast<T*>(p.out) + m * p.ldo + gcol, xr);
      } else {
        *reinterpret_cast<v8*>(reinterpret_cast<T*>(p.out) + m * p.ldo + gcol) = zq;
      }
    }
  }
}

// KS (32-wide K steps covering C) and rows per wave for a channel count, 0 when the chain kernels do not cover it
static int chain_ks(int C) {
  if (C <= 0 || C % 8 != 0 || C > 512) return 0;
  const int ks = (C + 31) / 32;
  return ks <= 4 ? ks : (ks <= 6 ? 6 : (ks <= 8 ? 8 : (ks <= 12 ? 12 : 16)));
}

// VKAS_CHAIN_PAIR=0: the one-wave-per-SIMD kernel at 256 < C <= 384 instead of the pair-split kernel (A/B switch)
static int chain_pair_mode() {
  static const int mode = [] {
    const char* v = getenv("VKAS_CHAIN_PAIR");
    return (v && v[0] == '0') ? 0 : 1;
  }();
  return mode;
}

template <typename T, int MODE>
static int launch_chain(const ChainArgs& a, hipStream_t st) {
  const int ks = chain_ks(a.C);
#define VKAS_CHAIN(KSV, TMV, MINB)                                                                \
  {                                                                                               \
    const long rows_per_block = 4L * TMV * 16;                                                    \
    const unsigned grid = (unsigned)vkas_cdiv(a.M, rows_per_block);                               \
    mlp_chain_kernel<T, KSV, TMV, MODE, MINB><<<grid, 256, 0, st>>>(a);                           \
  }
  switch (ks) {
    case 1: VKAS_CHAIN(1, 4, 2) break;
    case 2: VKAS_CHAIN(2, 4, 2) break;
    case 3: VKAS_CHAIN(3, 4, 2) break;
    case 4: VKAS_CHAIN(4, 2, 2) break;
    case 6: VKAS_CHAIN(6, 2, 2) break;
    case 8: VKAS_CHAIN(8, 2, 1) break;
    // 384 / 512 channels (stage 2 of Tiny / Base): one workgroup of four 512-register waves per CU; the weight images
    // (49 / 65 KB per chunk) stream from L2, 128 rows per workgroup keep that stream under the L2 -> LDS rate
    case 12:
      // 256 < C <= 384: the pair-split kernel (two 256-register waves per SIMD); VKAS_CHAIN_PAIR=0 keeps the one-wave form
      if (a.C % 16 == 0 && chain_pair_mode() != 0) {
        const unsigned grid = (unsigned)vkas_cdiv(a.M, 128);
        mlp_chain_pair_kernel<T, 12, 1, MODE><<<grid, 512, 0, st>>>(a);
      } else {
        VKAS_CHAIN(12, 2, 1)
      }
      break;
    case 16: VKAS_CHAIN(16, 2, 1) break;
    default:
      vkas_set_error("vkas_mlp_chain: C=%d is not covered (multiple of 8, <= 512)", a.C);
      return VKAS_E_ARG;
  }
#undef VKAS_CHAIN
  return VKAS_OK;
}

}  // namespace

extern "C" size_t vkas_mlp_chain_image_elems(int C) {
  const int ks = chain_ks(C);
  return ks ? (size_t)(4 * C / 32) * (size_t)chain_img_elems(ks) : 0;
}

extern "C" int vkas_mlp_chain_pack(const float* w1, const float* w2, const float* b1, int C, int mode, void* img, int dtype,
                                   void* stream) {
  VKAS_CHECK(w1 && w2 && img && vkas_aligned16(img) && (mode != 0 || (b1 && vkas_aligned16(b1))),
             "vkas_mlp_chain_pack: null / misaligned pointer");
  const int ks = chain_ks(C);
  VKAS_CHECK(ks > 0, "vkas_mlp_chain_pack: C=%d is not covered (multiple of 8, <= 512)", C);
  VKAS_CHECK(mode == 0 || mode == 1, "vkas_mlp_chain_pack: mode must be 0 (forward) or 1 (backward)");
  VKAS_CHECK(dtype == VKAS_BF16 || dtype == VKAS_F16, "vkas_mlp_chain_pack: 16-bit storage types only");
  const long pieces = (long)(4 * C / 32) * (chain_img_elems(ks) / 8);
  if (dtype == VKAS_BF16)
    mlp_chain_pack_kernel<bf16_t><<<(unsigned)vkas_cdiv(pieces, 256), 256, 0, vkas_stream(stream)>>>(w1, w2, b1, C, 4 * C, ks,
                                                                                                     mode, (bf16_t*)img);
  else
    mlp_chain_pack_kernel<f16_t><<<(unsigned)vkas_cdiv(pieces, 256), 256, 0, vkas_stream(stream)>>>(w1, w2, b1, C, 4 * C, ks,
                                                                                                    mode, (f16_t*)img);
  VKAS_LAUNCH_CHECK("mlp_chain_pack");
  return VKAS_OK;
}

static int chain_check_act(const char* who, const void* p, long ld, int width) {
  VKAS_CHECK(p && vkas_aligned16(p) && ld >= width && ld % 8 == 0, "%s: bad activation operand (ld=%ld, width=%d)", who, ld, width);
  return VKAS_OK;
}

static int chain_fwd_impl(const char* who, const void* yn, long ldyn, const float* ln_gamma, const float* ln_beta, void* ln_out, long ldln,
                          float* ln_stats, const void* img, const float* b2, const void* x, long ldx, const float* colscale,
                          const float* rowscale, int rows_per_image, void* h, long ldh, void* z, long ldz, void* out, long ldo,
                          long M, int C, int dtype, void* stream) {
  VKAS_CHECK(dtype == VKAS_BF16 || dtype == VKAS_F16, "%s: 16-bit storage types only", who);
  VKAS_CHECK(chain_ks(C) > 0, "%s: C=%d is not covered (multiple of 8, <= 512)", who, C);
  VKAS_CHECK(img && b2 && colscale && vkas_aligned16(img) && vkas_aligned16(b2) && vkas_aligned16(colscale),
             "%s: null / misaligned parameter", who);
  VKAS_CHECK(M >= 0 && rows_per_image > 0, "%s: bad sizes", who);
  int rc;
  if ((rc = chain_check_act(who, yn, ldyn, C)) || (rc = chain_check_act(who, x, ldx, C)) || (rc = chain_check_act(who, out, ldo, C)))
    return rc;
  // h and z exist for the backward pass only: both NULL = inference (nothing is written for them)
  VKAS_CHECK((h == nullptr) == (z == nullptr), "%s: h and z must be given (training) or omitted (inference) together", who);
  if (h && ((rc = chain_check_act(who, h, ldh, 4 * C)) || (rc = chain_check_act(who, z, ldz, C)))) return rc;
  if (M == 0) return VKAS_OK;
  ChainArgs a = {};
  a.a = yn; a.lda = ldyn; a.img = img; a.mid_out = h; a.ldm = ldh; a.bias_b = b2; a.res = x; a.ldres = ldx;
  a.colscale = colscale; a.rowscale = rowscale; a.rows_per_image = rows_per_image; a.z = z; a.ldz = ldz; a.out = out; a.ldo = ldo;
  a.M = M; a.C = C; a.HID = 4 * C;
  a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_out = ln_out; a.ldln = ldln; a.ln_stats = ln_stats;
  rc = dtype == VKAS_BF16 ? launch_chain<bf16_t, MODE_FWD>(a, vkas_stream(stream))
                          : launch_chain<f16_t, MODE_FWD>(a, vkas_stream(stream));
  if (rc) return rc;
  VKAS_LAUNCH_CHECK("mlp_chain_fwd");
  return VKAS_OK;
}

extern "C" int vkas_mlp_chain_fwd(const void* yn, long ldyn, const void* img, const float* b2, const void* x, long ldx, const float* colscale, const float* rowscale, int rows_per_image, void* h, long ldh,
                                  void* z, long ldz, void* out, long ldo, long M, int C, int dtype, void* stream) {
  return chain_fwd_impl("vkas_mlp_chain_fwd", yn, ldyn, nullptr, nullptr, nullptr, 0, nullptr, img, b2, x, ldx, colscale, rowscale,
                        rows_per_image, h, ldh, z, ldz, out, ldo, M, C, dtype, stream);
}

extern "C" int vkas_mlp_chain_ln_fwd(const void* y, long ldy, const float* ln_gamma, const float* ln_beta, void* yn, long ldyn,
                                     float* stats, const void* img, const float* b2, const void* x, long ldx,
                                     const float* colscale, const float* rowscale, int rows_per_image, void* h, long ldh, void* z,
                                     long ldz, void* out, long ldo, long M, int C, int dtype, void* stream) {
  const char* who = "vkas_mlp_chain_ln_fwd";
  VKAS_CHECK(ln_gamma && ln_beta && vkas_aligned16(ln_gamma) && vkas_aligned16(ln_beta), "%s: null / misaligned LayerNorm parameter", who);
  // the backward operands travel together: h, z, yn and stats all given (training) or all omitted (inference)
  VKAS_CHECK((yn == nullptr) == (h == nullptr) && (stats == nullptr) == (h == nullptr),
             "%s: yn and stats must be given (training) or omitted (inference) together with h and z", who);
  if (yn) {
    VKAS_CHECK(vkas_aligned16(yn) && ldyn >= C && ldyn % 8 == 0, "%s: bad activation operand yn (ld=%ld, width=%d)", who, ldyn, C);
  }
  return chain_fwd_impl(who, y, ldy, ln_gamma, ln_beta, yn, ldyn, stats, img, b2, x, ldx, colscale, rowscale, rows_per_image, h, ldh,
                        z, ldz, out, ldo, M, C, dtype, stream);
}

extern "C" int vkas_mlp_chain_bwd(const void* dz, long lddz, const void* img_t, const void* h, long ldh, void* dh, long lddh,
                                  void* dyn, long lddyn, long M, int C, int dtype, void* stream) {
  const char* who = "vkas_mlp_chain_bwd";
  VKAS_CHECK(dtype == VKAS_BF16 || dtype == VKAS_F16, "%s: 16-bit storage types only", who);
  VKAS_CHECK(chain_ks(C) > 0, "%s: C=%d is not covered (multiple of 8, <= 512)", who, C);
  VKAS_CHECK(img_t && vkas_aligned16(img_t) && M >= 0, "%s: bad arguments", who);
  int rc;
  if ((rc = chain_check_act(who, dz, lddz, C)) || (rc = chain_check_act(who, h, ldh, 4 * C)) || (rc = chain_check_act(who, dh, lddh, 4 * C)) ||
      (rc = chain_check_act(who, dyn, lddyn, C)))
    return rc;
  if (M == 0) return VKAS_OK;
  ChainArgs a = {};
  a.a = dz; a.lda = lddz; a.img = img_t; a.mid_out = dh; a.ldm = lddh; a.mid_in = h; a.ldmi = ldh; a.out = dyn; a.ldo = lddyn;
  a.M = M; a.C = C; a.HID = 4 * C;
  rc = dtype == VKAS_BF16 ? launch_chain<bf16_t, MODE_BWD>(a, vkas_stream(stream))
                          : launch_chain<f16_t, MODE_BWD>(a, vkas_stream(stream));
  if (rc) return rc;
  VKAS_LAUNCH_CHECK("mlp_chain_bwd");
  return VKAS_OK;
}

#ifdef CHAIN_TRACE
extern "C" int vkas_chain_trace_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vkas_chain_trace_buf), bytes < sizeof(vkas_chain_trace_buf) ? bytes : sizeof(vkas_chain_trace_buf));
}
#endif
