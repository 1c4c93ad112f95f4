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
template <> __device__ __forceinline__ void load_chunk<float>(const char* p, float* v) {
  load4(reinterpret_cast<const float*>(p), v);
}
template <typename T> __device__ __forceinline__ void store_chunk(T* p, const float* v);
template <> __device__ __forceinline__ void store_chunk<bf16_t>(bf16_t* p, const float* v) { store8(p, v); }
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
constexpr int PPW = 898;                        // words per input plane: >= PIY * PPITCH, = 2 (mod 32)
constexpr int POW = 2 * PTY * PTX + 4;          // words per fp32 output plane (2 per pixel): = 4 (mod 64)
static_assert(PPW >= PIY * PPITCH && PPW % 32 == 2 && POW % 64 == 4, "plane strides");
static_assert(8 * POW <= 16 * PPW, "an output half must fit the input planes it overwrites");

typedef __attribute__((ext_vector_type(2))) float f32x2;

__global__ __launch_bounds__(256, 2) void dwconv7x7_planar_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const bf16_t* __restrict__ addend, long ldadd,
                                                                   bf16_t* __restrict__ y, long ldy, int H, int W, int Cp,
                                                                   int cslices) {
  __shared__ __attribute__((aligned(16))) unsigned planes[16 * PPW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z / cslices;
  const int c0 = (blockIdx.z - b * cslices) * 32;
  const int x0 = blockIdx.x * PTX, y0 = blockIdx.y * PTY;

  // ---- stage the halo tile: item = (pixel, 16-byte chunk); 4 words -> planes 4*chunk .. 4*chunk+3
  for (int it = tid; it < PIY * PIX * 4; it += 256) {
    const int chunk = it & 3, q = it >> 2;
    const int iy = q / PIX, ix = q - iy * PIX;
    const int gy = y0 + iy - 3, gx = x0 + ix - 3;
    const int c = c0 + chunk * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cp)
      v = *reinterpret_cast<const uint4*>(x + (((long)b * H + gy) * W + gx) * ldx + c);
    unsigned* dst = planes + (chunk * 4) * PPW + iy * PPITCH + ix;
    dst[0] = v.x;
    dst[PPW] = v.y;
    dst[2 * PPW] = v.z;
    dst[3 * PPW] = v.w;
  }
  __syncthreads();

  // ---- compute: wave = chunk (4 pairs), lane = row r, strip s (8 pixels)
  const int r = lane >> 2, s = lane & 3;
  f32x2 res[4][8];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    const int P = wave * 4 + pp;          // plane = channel pair, wave-uniform
    const int ch = c0 + 2 * P;            // first channel of the pair
    const bool pok = ch < Cp;             // wave-uniform
    f32x2 acc[8];
    f32x2 bv = {0.f, 0.f};
    if (bias && pok) bv = *reinterpret_cast<const f32x2*>(bias + ch);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bv;
    if (pok) {
      const unsigned* pl = planes + P * PPW + r * PPITCH + s * 8;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const uint4 a0 = *reinterpret_cast<const uint4*>(pl + ky * PPITCH);
        const uint4 a1 = *reinterpret_cast<const uint4*>(pl + ky * PPITCH + 4);
        const uint4 a2 = *reinterpret_cast<const uint4*>(pl + ky * PPITCH + 8);
        const uint2 a3 = *reinterpret_cast<const uint2*>(pl + ky * PPITCH + 12);
        const unsigned wd[14] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y};
        f32x2 in[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) {
          in[j][0] = __builtin_bit_cast(float, wd[j] << 16);
          in[j][1] = __builtin_bit_cast(float, wd[j] & 0xffff0000u);
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x2 wk = *reinterpret_cast<const f32x2*>(w + (long)(ky * 7 + kx) * Cp + ch);  // wave-uniform: SGPR pair
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_elementwise_fma(in[j + kx], wk, acc[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) res[pp][j] = acc[j];
  }
  __syncthreads();  // every wave is done with the input planes

  // ---- results -> fp32 output planes (two halves: chunks {0,1} then {2,3}) -> 16-byte chunks of y
  float* oplanes = reinterpret_cast<float*>(planes);
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if ((wave >> 1) == half) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        float* dst = oplanes + ((wave & 1) * 4 + pp) * POW + (r * PTX + s * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; j += 2)
          *reinterpret_cast<float4*>(dst + j * 2) = make_float4(res[pp][j][0], res[pp][j][1], res[pp][j + 1][0], res[pp][j + 1][1]);
      }
    }
    __syncthreads();
    for (int it = tid; it < PTY * PTX * 2; it += 256) {
      const int chunk = it & 1, q = it >> 1;
      const int oy = q / PTX, ox = q - oy * PTX;
      const int gy = y0 + oy, gx = x0 + ox;
      const int c = c0 + (half * 2 + chunk) * 8;
      if (gy < H && gx < W && c < Cp) {
        const float* src = oplanes + (chunk * 4) * POW + q * 2;
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float2 t = *reinterpret_cast<const float2*>(src + k * POW);
          o[2 * k] = t.x;
          o[2 * k + 1] = t.y;
        }
        const long pix = ((long)b * H + gy) * W + gx;
        if (addend) {
          float a[8];
          load8(addend + pix * ldadd + c, a);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] += a[k];
        }
        store8(y + pix * ldy + c, o);
      }
    }
    __syncthreads();
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
  static const bool old_kernel = getenv("VKAS_DW_OLD") != nullptr;  // A/B switch
  if (dtype == VKAS_BF16 && !old_kernel) {
    const int cslices = (Cp + 31) / 32;
    VKAS_CHECK((long)B * cslices <= 65535, "vkas_dwconv7x7_fwd: B*cslices too large");
    dim3 grid((W + PTX - 1) / PTX, (H + PTY - 1) / PTY, B * cslices);
    dwconv7x7_planar_kernel<<<grid, 256, 0, vkas_stream(stream)>>>((const bf16_t*)x, ldx, w, bias, (const bf16_t*)addend, ldadd,
                                                                   (bf16_t*)y, ldy, H, W, Cp, cslices);
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

extern "C" size_t vkas_dwconv7x7_wgrad_ws_bytes(int B, int H, int W, int Cp) {
  return (size_t)dw_wgrad_parts(B, H, W) * 50 * (size_t)Cp * sizeof(float);
}

extern "C" int vkas_dwconv7x7_wgrad(const void* x, long ldx, const void* dy, long lddy, float* gw, float* gb,
                                    float* ws, size_t ws_bytes, int B, int H, int W, int Cp, int dtype,
                                    void* stream) {
  int rc = dw_check("vkas_dwconv7x7_wgrad", x, ldx, B, H, W, Cp);
  if (rc) return rc;
  rc = dw_check("vkas_dwconv7x7_wgrad(dy)", dy, lddy, B, H, W, Cp);
  if (rc) return rc;
  VKAS_CHECK(gw && gb && ws, "vkas_dwconv7x7_wgrad: null output/workspace");
  VKAS_CHECK(ws_bytes >= vkas_dwconv7x7_wgrad_ws_bytes(B, H, W, Cp), "vkas_dwconv7x7_wgrad: workspace too small");
  hipStream_t st = vkas_stream(stream);
  if (B == 0) {
    (void)hipMemsetAsync(gw, 0, 49L * Cp * sizeof(float), st);
    (void)hipMemsetAsync(gb, 0, (long)Cp * sizeof(float), st);
    return VKAS_OK;
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
  if (gb == gw + 49L * Cp) return vkas_colreduce_finalize(ws, P, 50 * Cp, 50 * Cp, gw, 0, st);  // one launch
  rc = vkas_colreduce_finalize(ws, P, 49 * Cp, 50 * Cp, gw, 0, st);
  if (rc) return rc;
  return vkas_colreduce_finalize(ws + 49L * Cp, P, Cp, 50 * Cp, gb, 0, st);
}
