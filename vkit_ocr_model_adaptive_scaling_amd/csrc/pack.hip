// Layout conversions between the reference's parameter / tensor layouts (fp32, (N,C,KH,KW), NCHW) and the
// kernels' layouts (packed K-major weights, NHWC activations), plus small elementwise kernels.
#include "vkas_common.h"

namespace {

// n_off / Nt: the weight fills output channels [n_off, n_off + Np) of an operand with Nt output channels in total (several
// convolutions that share their input packed side by side: the heads of a pass); n_off = 0, Nt = Np is the plain case.
// element i of a packed conv weight: value and destination index
__device__ __forceinline__ float pack_conv_elem(const float* __restrict__ w, long i, int N, int C, int KH, int KW, int Np, int Cp,
                                                int mode, int n_off, int Nt, long& o) {
  int n, c, ky, kx;
  long r = i;
  o = i;
  if (mode == 0) {  // [n][ky][kx][c]
    c = (int)(r % Cp); r /= Cp;
    kx = (int)(r % KW); r /= KW;
    ky = (int)(r % KH); r /= KH;
    n = (int)r;
    o = i + (long)n_off * KH * KW * Cp;
  } else if (mode == 1) {  // [c][ky'][kx'][n], taps rotated by 180 degrees
    n = (int)(r % Np); r /= Np;
    o = r * Nt + n_off + n;
    kx = KW - 1 - (int)(r % KW); r /= KW;
    ky = KH - 1 - (int)(r % KH); r /= KH;
    c = (int)r;
  } else {  // [(ky,kx,c)][n]
    n = (int)(r % Np); r /= Np;
    c = (int)(r % Cp); r /= Cp;
    kx = (int)(r % KW); r /= KW;
    ky = (int)r;
  }
  return (n < N && c < C) ? w[(((long)n * C + c) * KH + ky) * KW + kx] : 0.f;
}

// element i of a packed depthwise weight (see pack_dw_weight_kernel); false beyond the image
__device__ __forceinline__ bool pack_dw_elem(const float* __restrict__ w, int i, int C, int Cp, int flip, float& v) {
  if (i < 49 * Cp) {
    const int tap = i / Cp, c = i - tap * Cp;
    const int src_tap = flip ? 48 - tap : tap;
    v = c < C ? w[(long)c * 49 + src_tap] : 0.f;
    return true;
  }
  const int j = i - 49 * Cp;
  if (j >= (Cp / 2) * 112) return false;
  const int p = j / 112, r = j - p * 112;
  const int ky = r >> 4, e = r & 15;
  const int kx = e >> 1, c = 2 * p + (e & 1);
  v = 0.f;
  if (kx < 7 && c < C) {
    const int tap = ky * 7 + kx;
    v = w[(long)c * 49 + (flip ? 48 - tap : tap)];
  }
  return true;
}

// Every packed operand of a training step in ONE launch (vkas_pack_many): the optimizer changes all parameters at once, so
// their GEMM / depthwise images are rebuilt together instead of by ~150 launches of a few microseconds each.  Workgroup b
// serves 2048 elements of the entry e with block_start[e] <= b < block_start[e+1] (binary search in the table).
constexpr int PM_ELEMS = 2048;
__global__ __launch_bounds__(256) void pack_many_kernel(const vkas_pack_desc* __restrict__ descs,
                                                        const int* __restrict__ block_start, int count) {
  int lo = 0, hi = count;  // block_start[lo] <= blockIdx.x < block_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= block_start[mid]) lo = mid;
    else hi = mid;
  }
  const vkas_pack_desc d = descs[lo];
  const int lb = (int)blockIdx.x - block_start[lo];
  if (d.kind == 0 && d.mode == 1) {
    // dgrad layout [c][ky'][kx'][n] (taps rotated): a transpose of the (N, C*KH*KW) source.  Tile = 32 output rows x 64 output
    // channels through LDS, so that the fp32 reads run along the source row and the 16-bit writes along n (element by
    // element the reads of consecutive n are C*KH*KW floats apart: 0.5 ms per step for the 35 M parameters of config #3)
    __shared__ float tile[64][33];
    const int KHW = d.KH * d.KW, J = d.C * KHW, Jp = d.Cp * KHW;
    const int tiles_n = (d.Np + 63) >> 6;
    const int jo0 = (lb / tiles_n) * 32, n0 = (lb % tiles_n) * 64;
    {
      const int l = threadIdx.x & 31, r = threadIdx.x >> 5;
      const int jo = jo0 + l;
      const int c = jo / KHW, tap = KHW - 1 - (jo - c * KHW);
      const bool j_ok = jo < Jp && c < d.C;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int nl = rr * 8 + r, n = n0 + nl;
        tile[nl][l] = (j_ok && n < d.N) ? d.w[(long)n * J + c * KHW + tap] : 0.f;
      }
    }
    __syncthreads();
    const int l2 = threadIdx.x & 63, r2 = threadIdx.x >> 6;
    const int n = n0 + l2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row = k * 4 + r2, jo = jo0 + row;
      if (jo < Jp && n < d.Np) {
        const long o = (long)jo * d.Nt + d.n_off + n;
        const float v = tile[l2][row];
        if (d.dtype == VKAS_BF16) reinterpret_cast<bf16_t*>(d.out)[o] = from_f32<bf16_t>(v);
        else if (d.dtype == VKAS_F16) reinterpret_cast<f16_t*>(d.out)[o] = from_f32<f16_t>(v);
        else reinterpret_cast<float*>(d.out)[o] = v;
      }
    }
    return;
  }
  const long base = (long)lb * PM_ELEMS;
#pragma unroll
  for (int u = 0; u < PM_ELEMS / 256; ++u) {
    const long i = base + u * 256 + threadIdx.x;
    if (i >= d.total) return;
    if (d.kind == 1) {
      float v;
      if (pack_dw_elem(d.w, (int)i, d.C, d.Cp, d.mode, v)) reinterpret_cast<float*>(d.out)[i] = v;
    } else {
      long o;
      const float v = pack_conv_elem(d.w, i, d.N, d.C, d.KH, d.KW, d.Np, d.Cp, d.mode, d.n_off, d.Nt, o);
      if (d.dtype == VKAS_BF16) reinterpret_cast<bf16_t*>(d.out)[o] = from_f32<bf16_t>(v);
      else if (d.dtype == VKAS_F16) reinterpret_cast<f16_t*>(d.out)[o] = from_f32<f16_t>(v);
      else reinterpret_cast<float*>(d.out)[o] = v;
    }
  }
}

template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int N, int C, int KH, int KW,
                                        int Np, int Cp, int mode, long total, int n_off, int Nt) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long o2;
    const float v2 = pack_conv_elem(w, i, N, C, KH, KW, Np, Cp, mode, n_off, Nt, o2);
    out[o2] = from_f32<T>(v2);
  }
}

__global__ void unpack_conv_wgrad_kernel(const float* __restrict__ gw, float* __restrict__ grad, int N, int C, int KH,
                                         int KW, int Np, int Cp, int accumulate, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;  // index into grad (N,C,KH,KW)
    const int kx = (int)(r % KW); r /= KW;
    const int ky = (int)(r % KH); r /= KH;
    const int c = (int)(r % C); r /= C;
    const int n = (int)r;
    const float v = gw[(((long)n * KH + ky) * KW + kx) * Cp + c];
    grad[i] = accumulate ? grad[i] + v : v;
  }
}

__global__ void pad_vector_kernel(const float* __restrict__ v, float* __restrict__ out, int n, int np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) out[i] = i < n ? v[i] : 0.f;
}

// out = [49][Cp] taps x channels, followed by the channel-pair rows the planar bf16 kernel reads with ONE scalar load per
// kernel row: [Cp / 2 pairs][7 ky][16] = (w[ky][kx][2p], w[ky][kx][2p + 1]) for kx = 0..6, two pad floats
__global__ void pack_dw_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int Cp, int flip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 49*Cp + (Cp/2)*7*16
  float pv;
  if (pack_dw_elem(w, i, C, Cp, flip, pv)) out[i] = pv;
}

__global__ void unpack_dw_wgrad_kernel(const float* __restrict__ gw, float* __restrict__ grad, int C, int Cp,
                                       int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over C*49
  if (i >= C * 49) return;
  const int c = i / 49, tap = i - c * 49;
  const float v = gw[(long)tap * Cp + c];
  grad[i] = accumulate ? grad[i] + v : v;
}

template <typename T>
__global__ void image_to_nhwc8_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C, int H, int W) {
  const long total = (long)B * H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const long hw = (long)H * W;
    const int b = (int)(p / hw);
    const long r = p - (long)b * hw;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; ++c) v[c] = img[((long)b * C + c) * hw + r];
    store8(out + p * 8, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, long ld, float* __restrict__ out, int B, long HW, int C) {
  const long total = (long)B * C * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i % HW;
    const long bc = i / HW;
    const int c = (int)(bc % C);
    const long b = bc / C;
    out[i] = to_f32(x[(b * HW + r) * ld + c]);
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ g, T* __restrict__ out, long ld, int B, long HW, int C,
                                    int Cp) {
  const long total = (long)B * HW * Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const long p = i / Cp;
    const long b = p / HW;
    const long r = p - b * HW;
    const float v = c < C ? g[(b * C + c) * HW + r] : 0.f;
    out[p * ld + c] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long M, int nvec,
                                     int accumulate) {
  const long total = M * nvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const long m = i / nvec;
    float t[8];
    load8(x + m * ldx + v * 8, t);
    if (accumulate) {
      float u[8];
      load8(y + m * ldy + v * 8, u);
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] += u[k];
    }
    store8(y + m * ldy + v * 8, t);
  }
}

// Element-granular channel-range copy: y[m][c_dst + c] = c < C ? x[m][c_src + c] : 0 for c in [0, C + zero_tail).
// torch.cat of parts whose widths are not multiples of 8 (fpn.py:144 with out_channels = 400 -> 4 x 100 channels;
// upernext.py:82,197) lays the parts side by side WITHOUT pad channels, so a part starts at a 2-byte granular offset: the
// 16-byte vector kernels cannot address it.  Lanes run along the channels of a pixel (coalesced 2- / 4-byte accesses).
template <typename T>
__global__ void copy_channel_range_kernel(const T* __restrict__ x, long ldx, int c_src, T* __restrict__ y, long ldy, int c_dst,
                                          long M, int C, int Cz) {
  const long total = M * Cz;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cz);
    const long m = i / Cz;
    y[m * ldy + c_dst + c] = c < C ? x[m * ldx + c_src + c] : from_f32<T>(0.f);
  }
}

// nn.Softplus(beta=1, threshold=20): adaptive_scaling.py:101,140
__global__ void softplus_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v > 20.f ? v : log1pf(expf(v));
  }
}
__global__ void softplus_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                    long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    dx[i] = v > 20.f ? dy[i] : dy[i] / (1.f + expf(-v));
  }
}

static inline unsigned grid1d(long total, int bs = 256) {
  long g = vkas_cdiv(total, bs);
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int vkas_pack_conv_weight(const float* w, void* out, int N, int C, int KH, int KW, int Np, int Cp, int mode,
                                     int dtype, void* stream) {
  VKAS_CHECK(w && out, "vkas_pack_conv_weight: null pointer");
  VKAS_CHECK(N > 0 && C > 0 && KH > 0 && KW > 0 && Np >= N && Cp >= C && Np % 8 == 0 && Cp % 8 == 0,
             "vkas_pack_conv_weight: bad dims N=%d C=%d Np=%d Cp=%d", N, C, Np, Cp);
  VKAS_CHECK(mode >= 0 && mode <= 2, "vkas_pack_conv_weight: bad mode %d", mode);
  const long total = (long)Np * KH * KW * Cp;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_pack_conv_weight", {
    pack_conv_weight_kernel<T><<<grid1d(total), 256, 0, vkas_stream(stream)>>>(w, (T*)out, N, C, KH, KW, Np, Cp, mode,
                                                                              total, 0, Np);
  })
  VKAS_LAUNCH_CHECK("pack_conv_weight");
  return VKAS_OK;
}

extern "C" int vkas_pack_conv_weight_slice(const float* w, void* out, int N, int C, int KH, int KW, int Np, int Cp, int mode,
                                           int n_off, int Nt, int dtype, void* stream) {
  VKAS_CHECK(w && out, "vkas_pack_conv_weight_slice: null pointer");
  VKAS_CHECK(N > 0 && C > 0 && KH > 0 && KW > 0 && Np >= N && Cp >= C && Np % 8 == 0 && Cp % 8 == 0,
             "vkas_pack_conv_weight_slice: bad dims N=%d C=%d Np=%d Cp=%d", N, C, Np, Cp);
  VKAS_CHECK(mode == 0 || mode == 1, "vkas_pack_conv_weight_slice: mode must be 0 (forward) or 1 (dgrad)");
  VKAS_CHECK(n_off >= 0 && n_off % 8 == 0 && Nt % 8 == 0 && n_off + Np <= Nt, "vkas_pack_conv_weight_slice: bad slice %d + %d of %d",
             n_off, Np, Nt);
  const long total = (long)Np * KH * KW * Cp;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_pack_conv_weight_slice", {
    pack_conv_weight_kernel<T><<<grid1d(total), 256, 0, vkas_stream(stream)>>>(w, (T*)out, N, C, KH, KW, Np, Cp, mode,
                                                                              total, n_off, Nt);
  })
  VKAS_LAUNCH_CHECK("pack_conv_weight_slice");
  return VKAS_OK;
}

extern "C" int vkas_unpack_conv_wgrad(const float* gw, float* grad, int N, int C, int KH, int KW, int Np, int Cp,
                                      int accumulate, void* stream) {
  VKAS_CHECK(gw && grad, "vkas_unpack_conv_wgrad: null pointer");
  VKAS_CHECK(N > 0 && C > 0 && Np >= N && Cp >= C, "vkas_unpack_conv_wgrad: bad dims");
  const long total = (long)N * C * KH * KW;
  unpack_conv_wgrad_kernel<<<grid1d(total), 256, 0, vkas_stream(stream)>>>(gw, grad, N, C, KH, KW, Np, Cp, accumulate,
                                                                          total);
  VKAS_LAUNCH_CHECK("unpack_conv_wgrad");
  return VKAS_OK;
}

extern "C" int vkas_pad_vector(const float* v, float* out, int n, int np, void* stream) {
  VKAS_CHECK(v && out && n > 0 && np >= n, "vkas_pad_vector: bad arguments");
  pad_vector_kernel<<<(unsigned)vkas_cdiv(np, 256), 256, 0, vkas_stream(stream)>>>(v, out, n, np);
  VKAS_LAUNCH_CHECK("pad_vector");
  return VKAS_OK;
}

extern "C" int vkas_pack_many_blocks(const vkas_pack_desc* d) {
  if (!d) return 0;
  if (d->kind == 0 && d->mode == 1) return (int)(vkas_cdiv((long)d->Cp * d->KH * d->KW, 32) * vkas_cdiv(d->Np, 64));
  return (int)vkas_cdiv(d->total, 2048);
}

extern "C" int vkas_pack_many(const vkas_pack_desc* descs, const int* block_start, int count, int total_blocks, void* stream) {
  VKAS_CHECK(descs && block_start && count > 0 && total_blocks > 0, "vkas_pack_many: bad arguments");
  pack_many_kernel<<<(unsigned)total_blocks, 256, 0, vkas_stream(stream)>>>(descs, block_start, count);
  VKAS_LAUNCH_CHECK("pack_many");
  return VKAS_OK;
}

extern "C" size_t vkas_dw_weight_elems(int Cp) { return (size_t)105 * (size_t)Cp; }  // 49 Cp + (Cp / 2) * 7 * 16

extern "C" int vkas_pack_dw_weight(const float* w, float* out, int C, int Cp, int flip, void* stream) {
  VKAS_CHECK(w && out && C > 0 && Cp >= C && Cp % 8 == 0, "vkas_pack_dw_weight: bad arguments");
  pack_dw_weight_kernel<<<(unsigned)vkas_cdiv(105L * Cp, 256), 256, 0, vkas_stream(stream)>>>(w, out, C, Cp, flip);
  VKAS_LAUNCH_CHECK("pack_dw_weight");
  return VKAS_OK;
}

// dst[k][i] += src[k][i] for up to 16 short fp32 vectors in one launch (blockIdx.y = vector): the per-parameter
// gradient vectors of a layer (biases, LayerNorm affine, layer scale) added onto their flat-buffer views.
struct AccumMany {
  const float* src[16];
  float* dst[16];
  int n[16];
};
__global__ void accumulate_many_kernel(AccumMany a) {
  const int k = blockIdx.y;
  const float* s = a.src[k];
  float* d = a.dst[k];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += gridDim.x * blockDim.x) d[i] += s[i];
}

extern "C" int vkas_accumulate_many(int count, const float* const* src, float* const* dst, const int* n, void* stream) {
  VKAS_CHECK(count >= 0 && count <= 16 && (count == 0 || (src && dst && n)), "vkas_accumulate_many: bad arguments (count=%d)", count);
  if (count == 0) return VKAS_OK;
  AccumMany a;
  int nmax = 0;
  for (int k = 0; k < count; ++k) {
    VKAS_CHECK(src[k] && dst[k] && n[k] >= 0, "vkas_accumulate_many: bad vector %d", k);
    a.src[k] = src[k];
    a.dst[k] = dst[k];
    a.n[k] = n[k];
    nmax = n[k] > nmax ? n[k] : nmax;
  }
  if (nmax == 0) return VKAS_OK;
  long bx = vkas_cdiv(nmax, 256);
  if (bx > 64) bx = 64;
  accumulate_many_kernel<<<dim3((unsigned)bx, (unsigned)count), 256, 0, vkas_stream(stream)>>>(a);
  VKAS_LAUNCH_CHECK("accumulate_many");
  return VKAS_OK;
}

extern "C" int vkas_unpack_dw_wgrad(const float* gw, float* grad, int C, int Cp, int accumulate, void* stream) {
  VKAS_CHECK(gw && grad && C > 0 && Cp >= C, "vkas_unpack_dw_wgrad: bad arguments");
  unpack_dw_wgrad_kernel<<<(unsigned)vkas_cdiv(49L * C, 256), 256, 0, vkas_stream(stream)>>>(gw, grad, C, Cp, accumulate);
  VKAS_LAUNCH_CHECK("unpack_dw_wgrad");
  return VKAS_OK;
}

extern "C" int vkas_image_nchw_to_nhwc8(const float* img, void* out, int B, int C, int H, int W, int dtype,
                                        void* stream) {
  VKAS_CHECK(img && out && vkas_aligned16(out), "vkas_image_nchw_to_nhwc8: null/misaligned pointer");
  VKAS_CHECK(B >= 0 && C > 0 && C <= 8 && H > 0 && W > 0, "vkas_image_nchw_to_nhwc8: bad dims");
  if (B == 0) return VKAS_OK;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_image_nchw_to_nhwc8", {
    image_to_nhwc8_kernel<T><<<grid1d((long)B * H * W), 256, 0, vkas_stream(stream)>>>(img, (T*)out, B, C, H, W);
  })
  VKAS_LAUNCH_CHECK("image_nchw_to_nhwc8");
  return VKAS_OK;
}

extern "C" int vkas_nhwc_to_nchw_f32(const void* x, long ld, float* out, int B, int H, int W, int C, int dtype,
                                     void* stream) {
  VKAS_CHECK(x && out && B >= 0 && H > 0 && W > 0 && C > 0 && ld >= C, "vkas_nhwc_to_nchw_f32: bad arguments");
  if (B == 0) return VKAS_OK;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_nhwc_to_nchw_f32", {
    nhwc_to_nchw_kernel<T><<<grid1d((long)B * C * H * W), 256, 0, vkas_stream(stream)>>>((const T*)x, ld, out, B,
                                                                                        (long)H * W, C);
  })
  VKAS_LAUNCH_CHECK("nhwc_to_nchw_f32");
  return VKAS_OK;
}

extern "C" int vkas_nchw_f32_to_nhwc(const float* g, void* out, long ld, int B, int H, int W, int C, int Cp, int dtype,
                                     void* stream) {
  VKAS_CHECK(g && out && B >= 0 && H > 0 && W > 0 && C > 0 && Cp >= C && ld >= Cp, "vkas_nchw_f32_to_nhwc: bad arguments");
  if (B == 0) return VKAS_OK;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_nchw_f32_to_nhwc", {
    nchw_to_nhwc_kernel<T><<<grid1d((long)B * H * W * Cp), 256, 0, vkas_stream(stream)>>>(g, (T*)out, ld, B, (long)H * W,
                                                                                         C, Cp);
  })
  VKAS_LAUNCH_CHECK("nchw_f32_to_nhwc");
  return VKAS_OK;
}

extern "C" int vkas_copy_channels(const void* x, long ldx, void* y, long ldy, long M, int Cp, int accumulate, int dtype,
                                  void* stream) {
  VKAS_CHECK(x && y && vkas_aligned16(x) && vkas_aligned16(y), "vkas_copy_channels: null/misaligned pointer");
  VKAS_CHECK(Cp > 0 && Cp % 8 == 0 && ldx >= Cp && ldy >= Cp && ldx % 8 == 0 && ldy % 8 == 0, "vkas_copy_channels: bad strides");
  if (M <= 0) return VKAS_OK;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_copy_channels", {
    copy_channels_kernel<T><<<grid1d(M * (Cp / 8)), 256, 0, vkas_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, M,
                                                                                  Cp / 8, accumulate);
  })
  VKAS_LAUNCH_CHECK("copy_channels");
  return VKAS_OK;
}

extern "C" int vkas_copy_channel_range(const void* x, long ldx, int c_src, void* y, long ldy, int c_dst, long M, int C,
                                       int zero_tail, int dtype, void* stream) {
  VKAS_CHECK(x && y, "vkas_copy_channel_range: null pointer");
  VKAS_CHECK(C > 0 && zero_tail >= 0 && c_src >= 0 && c_dst >= 0 && ldx >= c_src + C && ldy >= c_dst + C + zero_tail,
             "vkas_copy_channel_range: channel range [%d, %d) / [%d, %d) outside the pixel strides %ld / %ld", c_src, c_src + C,
             c_dst, c_dst + C + zero_tail, ldx, ldy);
  if (M <= 0) return VKAS_OK;
  VKAS_DISPATCH_DTYPE(dtype, "vkas_copy_channel_range", {
    copy_channel_range_kernel<T><<<grid1d(M * (C + zero_tail)), 256, 0, vkas_stream(stream)>>>(
        (const T*)x, ldx, c_src, (T*)y, ldy, c_dst, M, C, C + zero_tail);
  })
  VKAS_LAUNCH_CHECK("copy_channel_range");
  return VKAS_OK;
}

extern "C" int vkas_softplus_fwd(const float* x, float* y, long n, void* stream) {
  VKAS_CHECK(x && y && n >= 0, "vkas_softplus_fwd: bad arguments");
  if (n == 0) return VKAS_OK;
  softplus_fwd_kernel<<<grid1d(n), 256, 0, vkas_stream(stream)>>>(x, y, n);
  VKAS_LAUNCH_CHECK("softplus_fwd");
  return VKAS_OK;
}

extern "C" int vkas_softplus_bwd(const float* x, const float* dy, float* dx, long n, void* stream) {
  VKAS_CHECK(x && dy && dx && n >= 0, "vkas_softplus_bwd: bad arguments");
  if (n == 0) return VKAS_OK;
  softplus_bwd_kernel<<<grid1d(n), 256, 0, vkas_stream(stream)>>>(x, dy, dx, n);
  VKAS_LAUNCH_CHECK("softplus_bwd");
  return VKAS_OK;
}
