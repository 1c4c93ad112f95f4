// Depthwise 7x7 (pad 3, stride 1) on NHWC activations: helper.py:61-73 used at convnext.py:30.
//
// Tiling: one workgroup = 8 x 32 output pixels x one 64-byte channel slice (32 bf16 / 16 fp32 channels).
// The 14 x 38 input halo tile is staged once in LDS (80-byte pixel pitch: the +16 B pad makes the
// per-thread 16-byte chunk reads of a 4-pixel strip land on distinct LDS slots), the 49 x slice weights in
// fp32 next to it.  A thread owns one 16-byte channel chunk and a 1 x 4 strip of outputs: per kernel row it
// reads 10 input chunks and 7 weight chunks from LDS for 28 * VEC FMAs.  HBM sees each input once per
// tile plus halo (2.08x from L2), each output once.
#include <stdlib.h>

#include "vkas_common.h"

int vkas_colreduce_finalize(const float* partial, long P, int n, int ldp, float* out, int accumulate, hipStream_t st);

namespace {

constexpr int TY = 8, TX = 32, IY = TY + 6, IX = TX + 6;
constexpr int PIXB = 80;  // bytes per staged pixel: 64 payload + 16 pad
constexpr int YG = 4;     // y tiles per workgroup in wgrad

template <typename T> struct Vec { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void load_chunk(const char* p, float* v);
template <> __device__ __forceinline__ void load_chunk<bf16_t>(const char* p, float* v) {
  load8(reinterpret_cast<const bf16_t*>(p), v);
}
template <> __device__ __forceinline__ void load_chunk<f16_t>(const char* p, float* v) {
  load8(reinterpret_cast<const f16_t*>(p), v);
}
template <> __device__ __forceinline__ void load_chunk<float>(const char* p, float* v) {
  load4(reinterpret_cast<const float*>(p), v);
}
template <typename T> __device__ __forceinline__ void store_chunk(T* p, const float* v);
template <> __device__ __forceinline__ void store_chunk<bf16_t>(bf16_t* p, const float* v) { store8(p, v); }
template <> __device__ __forceinline__ void store_chunk<f16_t>(f16_t* p, const float* v) { store8(p, v); }
template <> __device__ __forceinline__ void store_chunk<float>(float* p, const float* v) { store4(p, v); }

// stage the (IY x IX) halo tile around output tile origin (y0, x0), channel slice starting at c0
template <typename T>
__device__ __forceinline__ void stage_halo(char* tile, const T* __restrict__ x, long ldx, int b, int H, int W, int Cp,
                                           int y0, int x0, int c0) {
  constexpr int VEC = Vec<T>::N;
  for (int i = threadIdx.x; i < IY * IX * 4; i += 256) {
    const int ch = i & 3;
    const int p = i >> 2;
    const int py = p / IX, px = p - py * IX;
    const int gy = y0 + py - 3, gx = x0 + px - 3;
    const int c = c0 + ch * VEC;
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cp)
      v = *reinterpret_cast<const uint4*>(x + (((long)b * H + gy) * W + gx) * ldx + c);
    *reinterpret_cast<uint4*>(tile + p * PIXB + ch * 16) = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv7x7_fwd_kernel(const T* __restrict__ x, long ldx,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            const T* __restrict__ addend, long ldadd, T* __restrict__ y,
                                                            long ldy, int H, int W, int Cp, int cslices) {
  constexpr int VEC = Vec<T>::N;
  constexpr int CT = 4 * VEC;
  __shared__ __attribute__((aligned(16))) char tile[IY * IX * PIXB];
  __shared__ __attribute__((aligned(16))) float wl[49 * CT];
  const int tid = threadIdx.x;
  const int b = blockIdx.z / cslices;
  const int c0 = (blockIdx.z - b * cslices) * CT;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;

  for (int i = tid; i < 49 * CT; i += 256) {
    const int tap = i / CT, c = i - tap * CT;
    wl[i] = (c0 + c < Cp) ? w[(long)tap * Cp + c0 + c] : 0.f;
  }
  stage_halo<T>(tile, x, ldx, b, H, W, Cp, y0, x0, c0);
  __syncthreads();

  const int ch = tid & 3, xs = (tid >> 2) & 7, ty = tid >> 5;
  float acc[4][VEC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[j][c] = 0.f;

#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    float in[10][VEC];
    const char* row = tile + ((ty + ky) * IX + xs * 4) * PIXB + ch * 16;
#pragma unroll
    for (int j = 0; j < 10; ++j) load_chunk<T>(row + j * PIXB, in[j]);
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      float wv[VEC];
#pragma unroll
      for (int c = 0; c < VEC; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(&wl[(ky * 7 + kx) * CT + ch * VEC + c]);
        wv[c] = t.x; wv[c + 1] = t.y; wv[c + 2] = t.z; wv[c + 3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[j][c] = fmaf(in[j + kx][c], wv[c], acc[j][c]);
    }
  }

  const int cb = c0 + ch * VEC;
  const int oy = y0 + ty;
  if (cb >= Cp || oy >= H) return;
  float bv[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) bv[c] = bias ? bias[cb + c] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = x0 + xs * 4 + j;
    if (ox >= W) continue;
    const long pix = ((long)b * H + oy) * W + ox;
    float o[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) o[c] = acc[j][c] + bv[c];
    if (addend) {
      float a[VEC];
      load_chunk<T>(reinterpret_cast<const char*>(addend + pix * ldadd + cb), a);
#pragma unroll
      for (int c = 0; c < VEC; ++c) o[c] += a[c];
    }
    store_chunk<T>(y + pix * ldy + cb, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 forward / dgrad, "planar pair" formulation.  The kernel above spends as many VALU slots on bf16 -> fp32
// conversions and LDS operand reads as on FMAs (a lane re-reads and re-converts every input for each 1 x 4 strip and
// fetches its 8-channel weight vectors from LDS).  Here
//  * the 22 x 38 halo tile of a 16 x 32 output tile x 32 channels is staged ONCE into LDS as 16 channel-PAIR planes of
//    32-bit words (word = channels 2p, 2p+1 of one pixel; global side: 4 lanes x 16 B per pixel, plane stride = 2 mod 32
//    words so the 4 transposing ds_write_b32 of a wave hit 32 distinct banks);
//  * a wave works on ONE channel pair at a time: the 49 x 2 weights are wave-uniform and come from SGPRs (scalar
//    loads), a lane owns a 1 x 8 output strip, reads 14 consecutive words per kernel row (3 ds_read_b128 + 1 b64),
//    converts each once (2 ALU ops) and feeds 7 x 8 v_pk_fma_f32 (both channels of the pair per instruction):
//    ~31 VALU operations per output element instead of ~55, no weight traffic in LDS, 16 accumulator registers;
//  * results leave through LDS too (fp32, two halves of 8 planes) so that bias / the fused residual add of the dgrad
//    and the single rounding happen on whole 16-byte channel chunks with 4 lanes x 16 B per pixel on the global side.
constexpr int PTY = 16, PTX = 32, PIY = PTY + 6, PIX = PTX + 6;
constexpr int PPITCH = 40;                      // words per staged row (16-byte aligned rows)
constexpr int PPW = 1026;                       // words per plane: >= PIY * PPITCH (input) and >= 2 * PTY * PTX (fp32 results), = 2 (mod 32)
static_assert(PPW >= PIY * PPITCH && PPW >= 2 * PTY * PTX && PPW % 32 == 2, "plane stride");

// Timing-only ablation switches (never set in the shipped build): 1 no FMAs, 2 no LDS row reads, 4 no weight loads,
// 8 no halo loads, 16 no output phase.
// -DDW_TRACE (profiling builds only): per-workgroup sums of s_memtime cycles spent in the phases of dwconv7x7_mfma_kernel's
// tile loop - 0 fetch issue, 1 matrix products, 2 barrier, 3 result planes -> y, 4 staging, 5 barrier, 6 tiles -, read with
// vkas_dw_trace_read (profiles/trace_dw.py)
#ifdef DW_TRACE
__device__ unsigned long long vkas_dw_trace_buf[8192 * 8];
#define DW_TR(i)                                             \
  do {                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    tr_acc[i] += now_ - tr_last;                             \
    tr_last = now_;                                          \
  } while (0)
#else
#define DW_TR(i)
#endif
#ifndef DW_ABL
#define DW_ABL 0
#endif

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Persistent workgroups: a workgroup walks tiles (stride gridDim.x) and fetches the NEXT tile's halo into registers before
// it computes the current one - with one tile per workgroup the ~2 us global-load latency in front of every tile and the
// store tail behind it were as long as the arithmetic (2 workgroups per CU cannot hide them).
__global__ __launch_bounds__(256, 2) void dwconv7x7_planar_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const bf16_t* __restrict__ addend, long ldadd,
                                                                   bf16_t* __restrict__ y, long ldy, int B, int H, int W,
                                                                   int Cp, int cslices, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) unsigned planes[16 * PPW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 2, s = lane & 3;
  const float* wrows = w + 49L * Cp;
  const int ntiles = tiles_x * tiles_y * B * cslices;
  constexpr int NITEM = PIY * PIX * 4, NSTG = (NITEM + 255) / 256;

  // tile index -> (channel slice, image, tile row, tile column); x fastest: neighbours share halo lines in L2
  auto decode = [&](int t, int& b, int& c0, int& y0, int& x0) {
    const int tx = t % tiles_x;
    int q = t / tiles_x;
    const int ty = q % tiles_y;
    q /= tiles_y;
    const int cs = q % cslices;
    b = q / cslices;
    c0 = cs * 32;
    y0 = ty * PTY;
    x0 = tx * PTX;
  };
  // halo tile: item = (pixel, 16-byte chunk); all loads of a thread are issued together
  uint4 sv[NSTG];
  auto fetch = [&](int t) {
    int b, c0, y0, x0;
    decode(t, b, c0, y0, x0);
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
      const int it = tid + k * 256;
      const int chunk = it & 3, q = it >> 2;
      const int iy = q / PIX, ix = q - iy * PIX;
      const int gy = y0 + iy - 3, gx = x0 + ix - 3;
      const int c = c0 + chunk * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (it < NITEM && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cp && !(DW_ABL & 8))
        v = *reinterpret_cast<const uint4*>(x + (((long)b * H + gy) * W + gx) * ldx + c);
      sv[k] = v;
    }
  };
  // 4 words of a chunk -> planes 4*chunk .. 4*chunk+3 (plane stride = 2 mod 32: the wave's writes hit 32 distinct banks)
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
      const int it = tid + k * 256;
      if (it < NITEM) {
        const int chunk = it & 3, q = it >> 2;
        const int iy = q / PIX, ix = q - iy * PIX;
        unsigned* dst = planes + (chunk * 4) * PPW + iy * PPITCH + ix;
        dst[0] = sv[k].x;
        dst[PPW] = sv[k].y;
        dst[2 * PPW] = sv[k].z;
        dst[3 * PPW] = sv[k].w;
      }
    }
  };
  struct Row { uint4 a0, a1, a2; uint2 a3; };
  auto lrow = [&](int P, int ky) {
    const unsigned* pl = planes + P * PPW + (r + ky) * PPITCH + s * 8;
    Row t;
    if (DW_ABL & 2) {
      t.a0 = make_uint4(P, ky, r, s); t.a1 = t.a0; t.a2 = t.a0; t.a3 = make_uint2(P, ky);
      return t;
    }
    t.a0 = *reinterpret_cast<const uint4*>(pl);
    t.a1 = *reinterpret_cast<const uint4*>(pl + 4);
    t.a2 = *reinterpret_cast<const uint4*>(pl + 8);
    t.a3 = *reinterpret_cast<const uint2*>(pl + 12);
    return t;
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  fetch(t);
  stage();
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    int b, c0, y0, x0;
    decode(t, b, c0, y0, x0);
    const bool more = t + (int)gridDim.x < ntiles;
    if (more) fetch(t + gridDim.x);  // in flight behind this tile's arithmetic

    // ---- compute: wave = chunk (its 4 pair planes, which no other wave touches), lane = row r, strip s (8 pixels).
    // Pair-row weights (see pack_dw_weight_kernel): 16 floats per (pair, ky), fetched with ONE scalar load one kernel row
    // ahead of their use, like the 14 LDS words of the row.  The fp32 results of a pair overwrite the pair's own plane.
    auto wrow = [&](int P, int ky) {
      const int Pc = (c0 + 2 * P < Cp) ? (c0 >> 1) + P : 0;  // wave-uniform; an out-of-range pair reads pair 0 (unused)
      if (DW_ABL & 4) {
        f32x16 c;
        for (int i = 0; i < 16; ++i) c[i] = 0.01f * (float)(i + ky);
        return c;
      }
      return *reinterpret_cast<const f32x16*>(__builtin_assume_aligned(wrows + ((long)Pc * 7 + ky) * 16, 64));
    };
    f32x16 wn = wrow(wave * 4, 0);
    Row rn = lrow(wave * 4, 0);
#pragma unroll 1
    for (int pp = 0; pp < 4; ++pp) {
      const int P = wave * 4 + pp;          // plane = channel pair, wave-uniform
      const int ch = c0 + 2 * P;            // first channel of the pair
      f32x2 acc[8];
      f32x2 bv = {0.f, 0.f};
      if (bias && ch < Cp) bv = *reinterpret_cast<const f32x2*>(bias + ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bv;
#pragma unroll 1
      for (int ky = 0; ky < 7; ++ky) {
        const f32x16 wc = wn;
        const Row rc = rn;
        {  // next kernel row (of this pair, or row 0 of the next one): one address computation, no divergence
          const int Pn = ky < 6 ? P : (pp < 3 ? P + 1 : P);
          const int kn = ky < 6 ? ky + 1 : 0;
          wn = wrow(Pn, kn);
          rn = lrow(Pn, kn);
        }
        const unsigned wd[14] = {rc.a0.x, rc.a0.y, rc.a0.z, rc.a0.w, rc.a1.x, rc.a1.y, rc.a1.z, rc.a1.w,
                                 rc.a2.x, rc.a2.y, rc.a2.z, rc.a2.w, rc.a3.x, rc.a3.y};
        f32x2 in[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) {
          in[j][0] = __builtin_bit_cast(float, wd[j] << 16);
          in[j][1] = __builtin_bit_cast(float, wd[j] & 0xffff0000u);
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x2 wk = {wc[2 * kx], wc[2 * kx + 1]};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (DW_ABL & 1) acc[j] += (kx == 0 && j == 0) ? in[j + kx] * wk : (f32x2){0.f, 0.f};
            else acc[j] = __builtin_elementwise_fma(in[j + kx], wk, acc[j]);
          }
        }
      }
      float* dst = reinterpret_cast<float*>(planes + P * PPW) + (r * PTX + s * 8) * 2;
#pragma unroll
      for (int j = 0; j < 8; j += 2)
        *reinterpret_cast<float4*>(dst + j * 2) = make_float4(acc[j][0], acc[j][1], acc[j + 1][0], acc[j + 1][1]);
    }
    __syncthreads();

    // ---- fp32 result planes -> 16-byte channel chunks of y (+ the residual): item = (pixel, chunk)
    const float* oplanes = reinterpret_cast<const float*>(planes);
    constexpr int NOUT = (DW_ABL & 16) ? 0 : PTY * PTX * 4 / 256;
#pragma unroll 1
    for (int k0 = 0; k0 < NOUT; k0 += 2) {
      float av[2][8];
      if (addend) {  // residual loads first, in flight together
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int it = tid + (k0 + k) * 256;
          const int chunk = it & 3, q = it >> 2;
          const int oy = q / PTX, ox = q - oy * PTX;
          const int gy = y0 + oy, gx = x0 + ox;
          const int c = c0 + chunk * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) av[k][e] = 0.f;
          if (gy < H && gx < W && c < Cp) load8(addend + (((long)b * H + gy) * W + gx) * ldadd + c, av[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int it = tid + (k0 + k) * 256;
        const int chunk = it & 3, q = it >> 2;
        const int oy = q / PTX, ox = q - oy * PTX;
        const int gy = y0 + oy, gx = x0 + ox;
        const int c = c0 + chunk * 8;
        if (gy < H && gx < W && c < Cp) {
          const float* src = oplanes + (chunk * 4) * PPW + q * 2;
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float2 tt = *reinterpret_cast<const float2*>(src + e * PPW);
            o[2 * e] = tt.x;
            o[2 * e + 1] = tt.y;
          }
          if (addend) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += av[k][e];
          }
          store8(y + (((long)b * H + gy) * W + gx) * ldy + c, o);
        }
      }
    }
    if (!more) break;
    __syncthreads();  // the result planes have been read: the next halo tile may overwrite them
    stage();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-bit forward / dgrad on the MATRIX cores.  49 fp32 FMAs per output element put the VALU formulations above on the
// vector-ALU roof (157 TFLOP/s: 63 us for a stage-0 launch before a single conversion or LDS read is issued, HBM needs
// 73 us), so this kernel moves the arithmetic to MFMA: per channel and kernel row ky, the 7 taps along x are a banded
// (Toeplitz) 16 x 32 matrix A_ky[i][k] = w[ky][k - i - 5] that maps 32 input positions to 16 outputs, and
//      D[i = output x][j = image row] += A_ky[i][k] * B[k = input x][j],  B = the input of 16 image rows,
// is one v_mfma_f32_16x16x32 (7 useful of 32 k per row: 22 % of the matrix peak = 550 TFLOP/s effective, no operand
// conversions, fp32 accumulation).  B needs 8 consecutive x of ONE channel per lane, so the halo tile is staged channel-
// planar ([channel][row][x], 16-bit), transposed on the way in with ds_write_b16; results leave through fp32 planes and
// are re-assembled into 16-byte channel chunks.  The band matrices (28 registers per channel) are built once per workgroup:
// workgroups are persistent (8 waves), own a 16-channel slice (a wave 2 channels) and walk the spatial tiles of that slice with the
// next tile's halo prefetched into registers.
constexpr int MTY = 16, MTX = 32, MIY = MTY + 6, MIX = MTX + 6;
constexpr int MIP = 48;                         // elements per staged row: x0-8 .. x0+39 (16-byte aligned 8-element windows)
constexpr int MPB_IN = 2144 / 2;                // elements per input plane (>= 22 * 48; 536 words = 24 mod 32)
constexpr int MOP = 36;                         // floats per result row (144 B: the float4 writes of 16 rows hit 64 banks)
constexpr int MPB_OUT = 2336 / 4;               // floats per result plane (>= 16 * 36)
static_assert(MPB_IN >= MIY * MIP && MPB_OUT >= MTY * MOP, "plane sizes");

template <typename T> struct DwVec;
template <> struct DwVec<bf16_t> { typedef bf16x8 v8; };
template <> struct DwVec<f16_t> { typedef f16x8 v8; };
__device__ __forceinline__ f32x4 dw_mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 dw_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <typename T>
__global__ __launch_bounds__(512, 4) void dwconv7x7_mfma_kernel(const T* __restrict__ x, long ldx,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 const T* __restrict__ addend, long ldadd,
                                                                 T* __restrict__ y, long ldy, int B, int H, int W, int Cp,
                                                                 int cslices, int tiles_x, int tiles_y, int wg_per_slice) {
  typedef typename DwVec<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T in_planes[16 * MPB_IN];
  __shared__ __attribute__((aligned(16))) float out_planes[16 * MPB_OUT];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  // workgroup -> (channel slice, spatial walker).  The slices of one walker index sit on one XCD (blockIdx % 8 equal) and
  // walk the same tiles at the same time, so the 16-channel pieces of a pixel's row are assembled in that XCD's L2.
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int slice = rest % cslices;
  const int walker = (rest / cslices) * 8 + xcd;
  const int c0 = slice * 16;
  const int nsp = tiles_x * tiles_y * B;  // spatial tiles
  if (walker >= wg_per_slice) return;
  constexpr int NTHR = 512;
  constexpr int NITEM = MIY * MIX * 2, NSTG = (NITEM + NTHR - 1) / NTHR;
  for (int i = tid; i < 16 * MPB_IN / 2; i += NTHR) reinterpret_cast<unsigned*>(in_planes)[i] = 0u;  // margins stay zero

  auto decode = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % tiles_x;
    const int q = t / tiles_x;
    const int ty = q % tiles_y;
    b = q / tiles_y;
    y0 = ty * MTY;
    x0 = tx * MTX;
  };
  // halo tile: item = (pixel, 8-channel chunk of the slice); loads of a thread are issued together.  An item's position
  // inside the tile is the same for every tile of the walk: (iy, ix) is decoded once (the divisions by the 38-pixel row
  // were 1.3k of a tile's 8.4k cycles, phase timestamps of the -DDW_TRACE build)
  uint4 sv[NSTG];
  int item_yx[NSTG];  // iy << 8 | ix, -1 = no item
#pragma unroll
  for (int k = 0; k < NSTG; ++k) {
    const int it = tid + k * NTHR;
    const int q = it >> 1;
    const int iy = q / MIX, ix = q - iy * MIX;
    item_yx[k] = it < NITEM ? (iy << 8 | ix) : -1;
  }
  const int chunk = tid & 1;  // NTHR is even: the same for all items of a thread
  const bool c_ok = c0 + chunk * 8 < Cp;
  auto fetch = [&](int t) {
    int b, y0, x0;
    decode(t, b, y0, x0);
    const T* xb = x + (((long)b * H + (y0 - 3)) * W + (x0 - 3)) * ldx + c0 + chunk * 8;
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
      int yx = item_yx[k];
      asm volatile("" : "+v"(yx));  // opaque: addresses derived from it are recomputed per tile, not kept (and spilled)
      const int iy = yx >> 8, ix = yx & 255;
      const int gy = y0 + iy - 3, gx = x0 + ix - 3;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (yx >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c_ok && !(DW_ABL & 8))
        v = *reinterpret_cast<const uint4*>(xb + ((long)iy * W + ix) * ldx);
      sv[k] = v;
    }
  };
  int t = walker;
  if (t >= nsp) return;
  // the first halo tile is requested in front of the weight table: the two round trips to memory of a workgroup's prologue
  // travel together (at small maps - one or two tiles per workgroup - the prologue is most of the launch)
  fetch(t);
  // band matrices of this wave's 4 channels: A_ky[i][k], lane = (i, k = 8g .. 8g+7).  The slice's 49 x 16 weights go
  // through LDS (the result planes are idle): per-lane gathers straight from global memory would keep 224 loads in flight
  for (int i = tid; i < 49 * 16; i += NTHR) {
    const int tap = i >> 4, cl = i & 15;
    out_planes[i] = (c0 + cl < Cp) ? w[(long)tap * Cp + c0 + cl] : 0.f;
  }
  __syncthreads();
  v8 af[2][7];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int cl = wave * 2 + cc;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      v8 a;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int tap = 8 * g + e - li - 5;
        const bool ok = tap >= 0 && tap < 7;
        const float v = out_planes[((ky * 7 + (ok ? tap : 0)) << 4) + cl];
        a[e] = (T)(ok ? v : 0.f);
      }
      af[cc][ky] = a;
    }
    asm volatile("" ::: "memory");  // one channel's 56 reads at a time
  }
  float bv[2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) bv[cc] = (bias && c0 + wave * 2 + cc < Cp) ? bias[c0 + wave * 2 + cc] : 0.f;

  // channel 8*chunk + k of the slice -> plane 2k + chunk (the two chunk lanes of a pixel land 24 banks apart)
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
      int yx = item_yx[k];
      asm volatile("" : "+v"(yx));
      if (yx >= 0 && !(DW_ABL & 2)) {
        const int iy = yx >> 8, ix = yx & 255;
        unsigned short* dst = reinterpret_cast<unsigned short*>(in_planes) + chunk * MPB_IN + iy * MIP + ix + 5;
        const unsigned wd[4] = {sv[k].x, sv[k].y, sv[k].z, sv[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dst[(2 * e) * 2 * MPB_IN] = (unsigned short)(wd[e] & 0xffffu);
          dst[(2 * e + 1) * 2 * MPB_IN] = (unsigned short)(wd[e] >> 16);
        }
      }
    }
  };

  __syncthreads();  // zero fill done, the weight table in the result planes has been read
  stage();
  __syncthreads();
#ifdef DW_TRACE
  unsigned long long tr_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_readcyclecounter();
#endif
  for (; t < nsp; t += wg_per_slice) {
    int b, y0, x0;
    decode(t, b, y0, x0);
    const bool more = t + wg_per_slice < nsp;
    if (more) fetch(t + wg_per_slice);  // in flight behind this tile's matrix products
    DW_TR(0);

    // ---- compute: 7 MFMAs per (channel, 16-column group); D[i][j]: lane holds out x = 4g .. 4g+3 of row j = li
    // (lane / thread indices opaque per tile: the LDS addresses and item coordinates derived from them are a few integer
    // operations each - kept across tiles they were 19 spilled registers, reloaded through vmcnt behind the prefetch)
    int lane_t = lane, tid_t = tid;
    asm volatile("" : "+v"(lane_t), "+v"(tid_t));
    const int li = lane_t & 15, g = lane_t >> 4;
#pragma unroll
    for (int cc = 0; cc < ((DW_ABL & 1) ? 0 : 2); ++cc) {
      const int cl = wave * 2 + cc;
      const int P = (cl & 7) * 2 + (cl >> 3);
      const T* pl = in_planes + P * MPB_IN + li * MIP + 8 * g;
#pragma unroll
      for (int xg = 0; xg < 2; ++xg) {
        f32x4 acc = {bv[cc], bv[cc], bv[cc], bv[cc]};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
          v8 bf;
          if (DW_ABL & 64) bf = af[cc][(ky + 1) % 7];  // timing only: no fragment reads
          else bf = *reinterpret_cast<const v8*>(pl + ky * MIP + xg * 16);
          if (DW_ABL & 32) acc[ky & 3] += (float)bf[ky];  // timing only: no MFMA
          else acc = dw_mfma(af[cc][ky], bf, acc);
        }
        if (!(DW_ABL & 128)) *reinterpret_cast<f32x4*>(out_planes + P * MPB_OUT + li * MOP + xg * 16 + 4 * g) = acc;
        else asm volatile("" ::"v"(acc));
        asm volatile("" ::: "memory");  // keep the 7 fragment reads of the next group here (hoisting all 56 spills)
      }
    }
    // LDS-only barrier: __syncthreads() would also drain the vector-memory queue, i.e. wait for the prefetch just issued
    // and for the previous tile's stores at every tile (6 us per tile instead of ~2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DW_TR(1);
    __builtin_amdgcn_s_barrier();  // results complete; the input planes are free
    asm volatile("" ::: "memory");
    DW_TR(2);

    // ---- fp32 result planes -> 16-byte channel chunks of y (+ the residual): item = (pixel, chunk)
    constexpr int NOUT = (DW_ABL & 4) ? 0 : MTY * MTX * 2 / NTHR;
#pragma unroll 1
    for (int k = 0; k < NOUT; ++k) {
      const int it = tid_t + k * NTHR;
      const int chunk = it & 1, q = it >> 1;
      const int oy = q / MTX, ox = q - oy * MTX;
      const int gy = y0 + oy, gx = x0 + ox;
      const int c = c0 + chunk * 8;
      if (gy < H && gx < W && c < Cp) {
        const long pix = ((long)b * H + gy) * W + gx;
        float o[8];
        if (addend) load8(addend + pix * ldadd + c, o);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = 0.f;
        }
        const float* src = out_planes + chunk * MPB_OUT + oy * MOP + ox;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += src[e * 2 * MPB_OUT];
        store8(y + pix * ldy + c, o);
      }
    }
    DW_TR(3);
#ifdef DW_TRACE
    tr_acc[6] += 1;
#endif
    if (!more) break;
    stage();          // next halo tile (registers) -> input planes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DW_TR(4);
    __builtin_amdgcn_s_barrier();  // ... visible, and the result planes have been read
    asm volatile("" ::: "memory");
    DW_TR(5);
  }
#ifdef DW_TRACE
  if (tid == 0 && blockIdx.x < 8192)
    for (int i = 0; i < 7; ++i) vkas_dw_trace_buf[blockIdx.x * 8 + i] = tr_acc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-bit weight gradient on the matrix cores: gw[c][ky][kx] = sum_{y,x} dy[y][x][c] * x[y+ky-3][x+kx-3][c].
// For one channel, one input row r and the 32 columns of a tile,
//      D[i][j] += A[i][k] * B[k][j],   A[i][k] = dy[r - 3 + i][x0 + k]        (i < 7: the 7 output rows that see row r),
//                                      B[k][j] = x[r][x0 + k + j - 3]         (j < 7: the 7 lags; a Toeplitz matrix),
// accumulates all 49 taps at once: D[i][j] -> gw[ky = 6 - i][kx = j].  Column j = 7 of B is all ones, so D[3][7] collects
// sum dy (the bias gradient, counted once: i = 3 is the row r itself).  22 MFMAs per channel and tile instead of
// 16 * 32 * 49 FMAs.  A is a 16-byte aligned row read; B needs 8 consecutive elements at an element offset 8g + j that
// is odd for odd j: every lane reads 5 dwords and funnels them by 0 or 16 bits (v_alignbit).  Same persistent
// 16-channel-slice workgroups as the forward kernel; the 4-register accumulator of a channel lives across all tiles of the
// walker and leaves as one partial row per walker (summed by the finalize kernel).
constexpr int WXP = 40;                          // elements per staged x row (38 halo columns + 2)
constexpr int WDP = 48;                          // elements per staged dy row (32 + 16: 96-byte pitch)
constexpr int WPB_X = MIY * WXP;                 // elements per x plane
constexpr int WPB_D = MTY * WDP;                 // elements per dy plane

template <typename T>
__global__ __launch_bounds__(512, 4) void dwconv7x7_wgrad_mfma_kernel(const T* __restrict__ x, long ldx,
                                                                       const T* __restrict__ dy, long lddy,
                                                                       float* __restrict__ partial, int B, int H, int W,
                                                                       int Cp, int cslices, int tiles_x, int tiles_y,
                                                                       int wg_per_slice) {
  typedef typename DwVec<T>::v8 v8;
  constexpr int NTHR = 512;
  __shared__ __attribute__((aligned(16))) T xp[16 * WPB_X];
  __shared__ __attribute__((aligned(16))) T dp[16 * WPB_D + 64];  // + a zero row for the A lanes outside the tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int slice = rest % cslices;
  const int walker = (rest / cslices) * 8 + xcd;
  const int c0 = slice * 16;
  const int nsp = tiles_x * tiles_y * B;
  if (walker >= wg_per_slice) return;
  constexpr int NXI = MIY * MIX * 2, NXS = (NXI + NTHR - 1) / NTHR;   // x halo items (pixel, chunk)
  constexpr int NDI = MTY * MTX * 2, NDS = NDI / NTHR;                 // dy items
  for (int i = tid; i < 32; i += NTHR) reinterpret_cast<unsigned*>(dp + 16 * WPB_D)[i] = 0u;
  for (int i = tid; i < 16 * WPB_X / 2; i += NTHR) reinterpret_cast<unsigned*>(xp)[i] = 0u;  // columns 38, 39 stay zero

  auto decode = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % tiles_x;
    const int q = t / tiles_x;
    const int ty = q % tiles_y;
    b = q / tiles_y;
    y0 = ty * MTY;
    x0 = tx * MTX;
  };
  uint4 sx[NXS], sd[NDS];
  auto fetch = [&](int t) {
    int b, y0, x0;
    decode(t, b, y0, x0);
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int it = tid + k * NTHR;
      const int chunk = it & 1, q = it >> 1;
      const int iy = q / MIX, ix = q - iy * MIX;
      const int gy = y0 + iy - 3, gx = x0 + ix - 3;
      const int c = c0 + chunk * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (it < NXI && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cp && !(DW_ABL & 1024))
        v = *reinterpret_cast<const uint4*>(x + (((long)b * H + gy) * W + gx) * ldx + c);
      sx[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      const int it = tid + k * NTHR;
      const int chunk = it & 1, q = it >> 1;
      const int oy = q / MTX, ox = q - oy * MTX;
      const int gy = y0 + oy, gx = x0 + ox;
      const int c = c0 + chunk * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gy < H && gx < W && c < Cp && !(DW_ABL & 1024)) v = *reinterpret_cast<const uint4*>(dy + (((long)b * H + gy) * W + gx) * lddy + c);
      sd[k] = v;
    }
  };
  // channel 8*chunk + k of the slice -> plane 2k + chunk
  auto scatter8 = [&](unsigned short* dst, int plane_elems, const uint4& v) {
    const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[(2 * e) * 2 * plane_elems] = (unsigned short)(wd[e] & 0xffffu);
      dst[(2 * e + 1) * 2 * plane_elems] = (unsigned short)(wd[e] >> 16);
    }
  };
  auto stage = [&]() {
    if (DW_ABL & 256) {  // timing only (weight gradient: 256 no staging writes, 512 no products, 1024 no global loads)
#pragma unroll
      for (int k = 0; k < NXS; ++k) asm volatile("" ::"v"(sx[k].x), "v"(sx[k].y), "v"(sx[k].z), "v"(sx[k].w));
#pragma unroll
      for (int k = 0; k < NDS; ++k) asm volatile("" ::"v"(sd[k].x), "v"(sd[k].y), "v"(sd[k].z), "v"(sd[k].w));
      return;
    }
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int it = tid + k * NTHR;
      if (it < NXI) {
        const int chunk = it & 1, q = it >> 1;
        const int iy = q / MIX, ix = q - iy * MIX;
        scatter8(reinterpret_cast<unsigned short*>(xp) + chunk * WPB_X + iy * WXP + ix, WPB_X, sx[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      const int it = tid + k * NTHR;
      const int chunk = it & 1, q = it >> 1;
      const int oy = q / MTX, ox = q - oy * MTX;
      scatter8(reinterpret_cast<unsigned short*>(dp) + chunk * WPB_D + oy * WDP + ox, WPB_D, sd[k]);
    }
  };

  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // per-lane constants of the B gather: element offset 8g + j inside an x row (j clamped to a valid lag), funnel shift
  const int jl = li < 7 ? li : 6;
  const int eo = 8 * g + jl;
  const unsigned sh = (eo & 1) ? 16u : 0u;
  v8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;

  int t = walker;
  if (t < nsp) {
    fetch(t);
    __syncthreads();  // zero fills done
    stage();
    __syncthreads();
    for (; t < nsp; t += wg_per_slice) {
      const bool more = t + wg_per_slice < nsp;
      if (more) fetch(t + wg_per_slice);
#pragma unroll
      for (int cc = 0; cc < ((DW_ABL & 512) ? 0 : 2); ++cc) {
        const int cl = wave * 2 + cc;
        const int P = (cl & 7) * 2 + (cl >> 3);
        const unsigned* xrow = reinterpret_cast<const unsigned*>(xp + P * WPB_X) + (eo >> 1);
        const T* dpl = dp + P * WPB_D + 8 * g;
        const T* zrow = dp + 16 * WPB_D + 8 * g;
#pragma unroll 2
        for (int rr = 0; rr < MIY; ++rr) {  // input row r = y0 - 3 + rr
          // A: dy row (rr - 6 + i) of the tile for i < 7, zeros elsewhere
          const int p = rr - 6 + li;
          const T* ap = (li < 7 && (unsigned)p < (unsigned)MTY) ? dpl + p * WDP : zrow;
          const v8 af = *reinterpret_cast<const v8*>(ap);
          // B: 8 elements of x row rr starting at element eo
          const unsigned* bp = xrow + rr * (WXP / 2);
          const unsigned d0 = bp[0], d1 = bp[1], d2 = bp[2], d3 = bp[3], d4 = bp[4];
          union { unsigned u[4]; v8 v; } bf;
          bf.u[0] = __builtin_amdgcn_alignbit(d1, d0, sh);
          bf.u[1] = __builtin_amdgcn_alignbit(d2, d1, sh);
          bf.u[2] = __builtin_amdgcn_alignbit(d3, d2, sh);
          bf.u[3] = __builtin_amdgcn_alignbit(d4, d3, sh);
          acc[cc] = dw_mfma(af, li == 7 ? ones : bf.v, acc[cc]);
        }
      }
      if (!more) break;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done with the planes (LDS-only barrier: the prefetch stays in flight)
      asm volatile("" ::: "memory");
      stage();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  // partial row of this walker: [tap = ky*7 + kx][Cp] + row 49 = bias; D[i][j]: lane holds i = 4g + r, j = li
  float* prow = partial + (long)walker * 50 * Cp;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int ch = c0 + wave * 2 + cc;
    if (ch >= Cp) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * g + r;
      if (i < 7 && li < 7) prow[(long)((6 - i) * 7 + li) * Cp + ch] = acc[cc][r];
      if (i == 3 && li == 7) prow[49L * Cp + ch] = acc[cc][r];
    }
  }
}

// partial[p][tap][c] = sum over the workgroup's pixels of dy[pix][c] * x[pix + tap][c]; row 49 = sum dy
template <typename T>
__global__ __launch_bounds__(256, 2) void dwconv7x7_wgrad_kernel(const T* __restrict__ x, long ldx,
                                                              const T* __restrict__ dy, long lddy,
                                                              float* __restrict__ partial, int H, int W, int Cp,
                                                              int cslices) {
  constexpr int VEC = Vec<T>::N;
  constexpr int CT = 4 * VEC;
  __shared__ __attribute__((aligned(16))) char tile[IY * IX * PIXB];
  __shared__ __attribute__((aligned(16))) char dtile[TY * TX * PIXB];
  const int tid = threadIdx.x;
  const int b = blockIdx.z / cslices;
  const int cs = blockIdx.z - b * cslices;
  const int c0 = cs * CT;
  const int x0 = blockIdx.x * TX;
  const int ky = tid >> 5;  // 0..6 taps rows, 7 = bias row
  const int ch = tid & 3, pl = (tid >> 2) & 7;

  float acc[7][VEC];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[k][c] = 0.f;

  for (int yt = 0; yt < YG; ++yt) {
    const int y0 = (blockIdx.y * YG + yt) * TY;
    if (y0 >= H) break;
    __syncthreads();
    stage_halo<T>(tile, x, ldx, b, H, W, Cp, y0, x0, c0);
    for (int i = tid; i < TY * TX * 4; i += 256) {
      const int cq = i & 3;
      const int p = i >> 2;
      const int py = p / TX, px = p - py * TX;
      const int gy = y0 + py, gx = x0 + px;
      const int c = c0 + cq * VEC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gy < H && gx < W && c < Cp) v = *reinterpret_cast<const uint4*>(dy + (((long)b * H + gy) * W + gx) * lddy + c);
      *reinterpret_cast<uint4*>(dtile + p * PIXB + cq * 16) = v;
    }
    __syncthreads();
    // pixel lane pl walks row pl of the tile left to right with a 7-pixel sliding window of x in registers: one new
    // x chunk (and one dy chunk) is read and converted per pixel instead of seven (the window slot of x[px + kx] is
    // (px + kx) % 7, fixed at compile time by unrolling 7 pixels)
    const char* drow = dtile + (pl * TX) * PIXB + ch * 16;
    if (ky < 7) {
      const char* xrow = tile + ((pl + ky) * IX) * PIXB + ch * 16;
      float win[7][VEC];
#pragma unroll
      for (int j = 0; j < 6; ++j) load_chunk<T>(xrow + j * PIXB, win[j]);
#pragma unroll 1
      for (int px0 = 0; px0 < TX; px0 += 7) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
          const int px = px0 + m;
          if (px < TX) {
            load_chunk<T>(xrow + (px + 6) * PIXB, win[(m + 6) % 7]);
            float d[VEC];
            load_chunk<T>(drow + px * PIXB, d);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
              for (int c = 0; c < VEC; ++c) acc[kx][c] = fmaf(d[c], win[(m + kx) % 7][c], acc[kx][c]);
            asm volatile("" ::: "memory");  // keep the LDS reads of later pixels here: hoisting all 14 costs 112 VGPRs
          }
        }
      }
    } else {
      for (int px = 0; px < TX; ++px) {
        float d[VEC];
        load_chunk<T>(drow + px * PIXB, d);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[0][c] += d[c];
      }
    }
  }
  // reduce over the 8 pixel lanes (thread bits 2..4)
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      float v = acc[k][c];
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      acc[k][c] = v;
    }
  const int cb = c0 + ch * VEC;
  if (pl == 0 && cb < Cp) {
    const long pidx = ((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* dst = partial + pidx * 50 * (long)Cp;
    if (ky < 7) {
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int c = 0; c < VEC; ++c) dst[(long)(ky * 7 + kx) * Cp + cb + c] = acc[kx][c];
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) dst[49L * Cp + cb + c] = acc[0][c];
    }
  }
}

}  // namespace

// persistent walkers per 16-channel slice of the matrix-core kernels: two workgroups per CU in all, whole groups of 8
// (one per XCD), never more than there are spatial tiles (rounded up to 8)
static long dw_mfma_walkers(int B, int H, int W, int Cp) {
  const int cslices = (Cp + 15) / 16;
  const long nsp = (long)vkas_cdiv(W, MTX) * vkas_cdiv(H, MTY) * B;
  long wps = 512 / cslices;
  wps = wps / 8 * 8;
  if (wps < 8) wps = 8;
  if (wps > nsp) wps = (nsp + 7) / 8 * 8;
  return wps;
}

static int dw_check(const char* who, const void* x, long ldx, int B, int H, int W, int Cp) {
  VKAS_CHECK(x && vkas_aligned16(x), "%s: null/misaligned tensor", who);
  VKAS_CHECK(B >= 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0, "%s: bad dims B=%d H=%d W=%d Cp=%d", who, B, H, W, Cp);
  VKAS_CHECK(ldx >= Cp && ldx % 8 == 0, "%s: bad pixel stride %ld", who, ldx);
  return VKAS_OK;
}

extern "C" int vkas_dwconv7x7_fwd(const void* x, long ldx, const float* w, const float* bias, const void* addend,
                                  long ldadd, void* y, long ldy, int B, int H, int W, int Cp, int dtype,
                                  void* stream) {
  int rc = dw_check("vkas_dwconv7x7_fwd", x, ldx, B, H, W, Cp);
  if (rc) return rc;
  rc = dw_check("vkas_dwconv7x7_fwd(y)", y, ldy, B, H, W, Cp);
  if (rc) return rc;
  VKAS_CHECK(w, "vkas_dwconv7x7_fwd: null weights");
  if (addend) {
    rc = dw_check("vkas_dwconv7x7_fwd(addend)", addend, ldadd, B, H, W, Cp);
    if (rc) return rc;
  }
  if (B == 0) return VKAS_OK;
  // The planar-pair kernel is an experiment kept for A/B runs (VKAS_DW_PLANAR=1): with FMAs, conversions, LDS reads and
  // staging all on the same two waves per SIMD it measures 300 us at stage 0 against 252 us of the kernel above
  // (profiles/bench_dw.py; ablations in DESIGN.md): both sit on the fp32 VALU roof, not on HBM.
  static const bool planar = getenv("VKAS_DW_PLANAR") != nullptr;
  static const bool valu = getenv("VKAS_DW_VALU") != nullptr;  // A/B switch: the round-1 vector-ALU kernel for 16-bit types too
  if ((dtype == VKAS_BF16 || dtype == VKAS_F16) && !valu && !planar) {
    const int cslices = (Cp + 15) / 16;
    const int tiles_x = (W + MTX - 1) / MTX, tiles_y = (H + MTY - 1) / MTY;
    const long nsp = (long)tiles_x * tiles_y * B;
    VKAS_CHECK(nsp < (1L << 30), "vkas_dwconv7x7_fwd: too many tiles");
    const long wps = dw_mfma_walkers(B, H, W, Cp);
    const unsigned grid = (unsigned)(wps * cslices);
#define VKAS_DWM(TT)                                                                                                      \
  dwconv7x7_mfma_kernel<TT><<<grid, 512, 0, vkas_stream(stream)>>>((const TT*)x, ldx, w, bias, (const TT*)addend, ldadd, (TT*)y, \
                                                                  ldy, B, H, W, Cp, cslices, tiles_x, tiles_y, (int)wps)
    if (dtype == VKAS_BF16) VKAS_DWM(bf16_t);
    else VKAS_DWM(f16_t);
#undef VKAS_DWM
    VKAS_LAUNCH_CHECK("dwconv7x7_mfma");
    return VKAS_OK;
  }
  if (dtype == VKAS_BF16 && planar) {
    const int cslices = (Cp + 31) / 32;
    const int tiles_x = (W + PTX - 1) / PTX, tiles_y = (H + PTY - 1) / PTY;
    const long ntiles = (long)tiles_x * tiles_y * B * cslices;
    VKAS_CHECK(ntiles < (1L << 30), "vkas_dwconv7x7_fwd: too many tiles");
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);  // 2 persistent workgroups per CU
    dwconv7x7_planar_kernel<<<grid, 256, 0, vkas_stream(stream)>>>((const bf16_t*)x, ldx, w, bias, (const bf16_t*)addend, ldadd,
                                                                   (bf16_t*)y, ldy, B, H, W, Cp, cslices, tiles_x, tiles_y);
    VKAS_LAUNCH_CHECK("dwconv7x7_planar");
    return VKAS_OK;
  }
  VKAS_DISPATCH_DTYPE(dtype, "vkas_dwconv7x7_fwd", {
    constexpr int CT = 4 * Vec<T>::N;
    const int cslices = (Cp + CT - 1) / CT;
    VKAS_CHECK((long)B * cslices <= 65535, "vkas_dwconv7x7_fwd: B*cslices too large");
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, B * cslices);
    dwconv7x7_fwd_kernel<T><<<grid, 256, 0, vkas_stream(stream)>>>((const T*)x, ldx, w, bias, (const T*)addend, ldadd,
                                                                   (T*)y, ldy, H, W, Cp, cslices);
  })
  VKAS_LAUNCH_CHECK("dwconv7x7_fwd");
  return VKAS_OK;
}

static long dw_wgrad_parts(int B, int H, int W) {
  return (long)B * vkas_cdiv(vkas_cdiv(H, TY), YG) * vkas_cdiv(W, TX);
}

#ifdef DW_TRACE
extern "C" int vkas_dw_trace_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vkas_dw_trace_buf), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" size_t vkas_dwconv7x7_wgrad_ws_bytes(int B, int H, int W, int Cp) {
  long parts = dw_wgrad_parts(B, H, W);
  const long walkers = dw_mfma_walkers(B, H, W, Cp);
  if (walkers > parts) parts = walkers;
  return (size_t)parts * 50 * (size_t)Cp * sizeof(float);
}

extern "C" long vkas_dwconv7x7_wgrad_parts(int B, int H, int W, int Cp, int dtype) {  // partial rows the launch leaves in ws
  if (B == 0) return 0;
  static const bool valu = getenv("VKAS_DW_VALU") != nullptr;
  return ((dtype == VKAS_BF16 || dtype == VKAS_F16) && !valu) ? dw_mfma_walkers(B, H, W, Cp) : dw_wgrad_parts(B, H, W);
}

extern "C" int vkas_dwconv7x7_wgrad(const void* x, long ldx, const void* dy, long lddy, float* gw, float* gb,
                                    float* ws, size_t ws_bytes, int B, int H, int W, int Cp, int dtype,
                                    void* stream) {
  int rc = dw_check("vkas_dwconv7x7_wgrad", x, ldx, B, H, W, Cp);
  if (rc) return rc;
  rc = dw_check("vkas_dwconv7x7_wgrad(dy)", dy, lddy, B, H, W, Cp);
  if (rc) return rc;
  VKAS_CHECK(ws && (gw != nullptr) == (gb != nullptr), "vkas_dwconv7x7_wgrad: null output/workspace");
  VKAS_CHECK(ws_bytes >= vkas_dwconv7x7_wgrad_ws_bytes(B, H, W, Cp), "vkas_dwconv7x7_wgrad: workspace too small");
  hipStream_t st = vkas_stream(stream);
  if (B == 0) {
    if (gw) {
      (void)hipMemsetAsync(gw, 0, 49L * Cp * sizeof(float), st);
      (void)hipMemsetAsync(gb, 0, (long)Cp * sizeof(float), st);
    }
    return VKAS_OK;
  }
  static const bool valu = getenv("VKAS_DW_VALU") != nullptr;  // A/B switch: the round-1 vector-ALU kernel
  if ((dtype == VKAS_BF16 || dtype == VKAS_F16) && !valu) {
    const int cslices = (Cp + 15) / 16;
    const int tiles_x = (W + MTX - 1) / MTX, tiles_y = (H + MTY - 1) / MTY;
    VKAS_CHECK((long)tiles_x * tiles_y * B < (1L << 30), "vkas_dwconv7x7_wgrad: too many tiles");
    const long wps = dw_mfma_walkers(B, H, W, Cp);
    const unsigned grid = (unsigned)(wps * cslices);
    if (dtype == VKAS_BF16)
      dwconv7x7_wgrad_mfma_kernel<bf16_t><<<grid, 512, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, ws, B, H, W, Cp,
                                                               cslices, tiles_x, tiles_y, (int)wps);
    else
      dwconv7x7_wgrad_mfma_kernel<f16_t><<<grid, 512, 0, st>>>((const f16_t*)x, ldx, (const f16_t*)dy, lddy, ws, B, H, W, Cp,
                                                              cslices, tiles_x, tiles_y, (int)wps);
    VKAS_LAUNCH_CHECK("dwconv7x7_wgrad_mfma");
    if (!gw) return VKAS_OK;  // partial rows (50 Cp floats: 49 Cp weights | Cp bias) stay in ws: vkas_dwconv7x7_wgrad_parts
    if (gb == gw + 49L * Cp) return vkas_colreduce_finalize(ws, wps, 50 * Cp, 50 * Cp, gw, 0, st);
    rc = vkas_colreduce_finalize(ws, wps, 49 * Cp, 50 * Cp, gw, 0, st);
    if (rc) return rc;
    return vkas_colreduce_finalize(ws + 49L * Cp, wps, Cp, 50 * Cp, gb, 0, st);
  }
  const long P = dw_wgrad_parts(B, H, W);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_dwconv7x7_wgrad", {
    constexpr int CT = 4 * Vec<T>::N;
    const int cslices = (Cp + CT - 1) / CT;
    VKAS_CHECK((long)B * cslices <= 65535, "vkas_dwconv7x7_wgrad: B*cslices too large");
    dim3 grid((W + TX - 1) / TX, (unsigned)vkas_cdiv(vkas_cdiv(H, TY), YG), B * cslices);
    dwconv7x7_wgrad_kernel<T><<<grid, 256, 0, st>>>((const T*)x, ldx, (const T*)dy, lddy, ws, H, W, Cp, cslices);
  })
  VKAS_LAUNCH_CHECK("dwconv7x7_wgrad");
  if (!gw) return VKAS_OK;
  if (gb == gw + 49L * Cp) return vkas_colreduce_finalize(ws, P, 50 * Cp, 50 * Cp, gw, 0, st);  // one launch
  rc = vkas_colreduce_finalize(ws, P, 49 * Cp, 50 * Cp, gw, 0, st);
  if (rc) return rc;
  return vkas_colreduce_finalize(ws + 49L * Cp, P, Cp, 50 * Cp, gb, 0, st);
}
