// C-ABI entry points of the implicit-GEMM family + error plumbing + column sums.
#include <stdlib.h>
#include <string.h>

#include "gemm_parts.h"

int vkas_gemm_nt_simple(const void*, const vkas_conv_geom*, const void*, int, const vkas_epilogue*, int, hipStream_t);
int vkas_gemm_tn_simple(const void*, const vkas_conv_geom*, const void*, long, int, float*, float*, int, int, hipStream_t);
int vkas_gemm_nt_mfma_bf16(const void*, const vkas_conv_geom*, const void*, int, const vkas_epilogue*, hipStream_t);
int vkas_gemm_tn_mfma_bf16(const void*, const vkas_conv_geom*, const void*, long, int, float*, float*, int, hipStream_t);
int vkas_gemm_nt_mfma_f16(const void*, const vkas_conv_geom*, const void*, int, const vkas_epilogue*, hipStream_t);
int vkas_gemm_tn_mfma_f16(const void*, const vkas_conv_geom*, const void*, long, int, float*, float*, int, hipStream_t);

int vkas_gemm_nt_tile_choice(long M, int Np);
int vkas_gemm_tn_tile_choice(long M, int Np, int K);
bool vkas_nt_slab_eligible(const vkas_conv_geom* g, int Np);
int vkas_gemm_nt_ring_stages(const vkas_conv_geom* g, int Np);
bool vkas_tn_slab_eligible(const vkas_conv_geom* g, int Np, long lddy);
bool vkas_tn_slab_n112(int Np);
bool vkas_tn_slab_n96(int Np);

static thread_local char g_err[512] = "";

void vkas_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vkas_last_error(void) { return g_err; }
extern "C" int vkas_abi_version(void) { return 1; }

static bool force_simple() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("VKAS_GEMM");
    v = (s && strcmp(s, "simple") == 0) ? 1 : 0;
  }
  return v == 1;
}

int vkas_check_geom(const char* who, const void* x, const vkas_conv_geom* g, int Np) {
  VKAS_CHECK(x && g, "%s: null input", who);
  VKAS_CHECK(g->B >= 0 && g->Hin > 0 && g->Win > 0 && g->Hout > 0 && g->Wout > 0, "%s: bad spatial dims", who);
  VKAS_CHECK(g->Cp > 0 && g->Cp % 8 == 0, "%s: Cp=%d must be a positive multiple of 8", who, g->Cp);
  VKAS_CHECK(g->ldx >= g->Cp && g->ldx % 8 == 0, "%s: ldx=%d must be >= Cp and a multiple of 8", who, g->ldx);
  VKAS_CHECK(g->KH > 0 && g->KW > 0 && g->stride > 0 && g->pad >= 0, "%s: bad kernel geometry", who);
  VKAS_CHECK((g->Hin + 2 * g->pad - g->KH) / g->stride + 1 == g->Hout &&
                 (g->Win + 2 * g->pad - g->KW) / g->stride + 1 == g->Wout,
             "%s: output size %dx%d inconsistent with input %dx%d k=%dx%d s=%d p=%d", who, g->Hout, g->Wout, g->Hin,
             g->Win, g->KH, g->KW, g->stride, g->pad);
  VKAS_CHECK(Np > 0 && Np % 8 == 0, "%s: Np=%d must be a positive multiple of 8", who, Np);
  VKAS_CHECK(vkas_aligned16(x), "%s: x not 16-byte aligned", who);
  VKAS_CHECK((long)g->B * g->Hin * g->Win * (long)g->ldx < (1L << 40), "%s: tensor too large", who);
  return VKAS_OK;
}

int vkas_check_epilogue(const char* who, const vkas_epilogue* e, int Np) {
  // the fused head tail may run without z / statistics outputs (inference: nothing is kept for a backward pass)
  VKAS_CHECK(e && (e->out || e->mode == VKAS_EPI_HEAD || (e->mode == VKAS_EPI_GELU && e->out2)), "%s: null output", who);
  VKAS_CHECK(vkas_aligned16(e->out) && e->ldo % 8 == 0, "%s: out misaligned (ldo=%ld)", who, e->ldo);
  VKAS_CHECK(e->mode >= VKAS_EPI_NONE && e->mode <= VKAS_EPI_HEAD, "%s: bad epilogue mode %d", who, e->mode);
  if (e->mode == VKAS_EPI_HEAD) {
    const vkas_head_desc* h = &e->head;
    VKAS_CHECK(h->n_heads >= 1 && h->n_heads <= 4 && h->pw % 8 == 0 && h->pw >= 8 && h->pw <= 224 && h->params &&
                   (h->stats != nullptr) == (e->out != nullptr) && h->proj && e->bias && vkas_aligned16(h->params) && vkas_aligned16(h->proj),
               "%s: bad head descriptor", who);
    for (int i = 0; i < h->n_heads; ++i)
      VKAS_CHECK(h->np[i] % 8 == 0 && h->np[i] > 0 && h->np[i] <= h->pw && h->c[i] > 0 && h->c[i] <= h->np[i] &&
                     h->oc[i] >= 1 && h->oc[i] <= 4 && h->n0[i] % 8 == 0 && h->n0[i] >= 0 && h->n0[i] + h->np[i] <= Np,
                 "%s: bad head %d (n0=%d np=%d c=%d oc=%d)", who, i, h->n0[i], h->np[i], h->c[i], h->oc[i]);
  }
  VKAS_CHECK((!e->bias || vkas_aligned16(e->bias)) && (!e->colscale || vkas_aligned16(e->colscale)),
             "%s: bias / colscale must be 16-byte aligned", who);
  if (e->mode != VKAS_EPI_PATCH && e->out) VKAS_CHECK(e->ldo >= Np, "%s: ldo=%ld < Np=%d", who, e->ldo, Np);
  if (e->mode == VKAS_EPI_GELU)
    VKAS_CHECK(e->out2 && vkas_aligned16(e->out2) && e->ldo2 >= Np && e->ldo2 % 8 == 0, "%s: GELU needs out2", who);
  if (e->mode == VKAS_EPI_SCALE_RES) {
    VKAS_CHECK(e->aux && e->colscale && e->rows_per_image > 0, "%s: SCALE_RES needs aux, colscale, rows_per_image", who);
    VKAS_CHECK(!e->out2 || (vkas_aligned16(e->out2) && e->ldo2 >= Np && e->ldo2 % 8 == 0), "%s: bad out2", who);
  }
  if (e->mode == VKAS_EPI_SCALE_RES || e->mode == VKAS_EPI_DGELU || e->mode == VKAS_EPI_ADD)
    VKAS_CHECK(e->aux && vkas_aligned16(e->aux) && e->ldaux >= Np && e->ldaux % 8 == 0, "%s: bad aux", who);
  if (e->mode == VKAS_EPI_PATCH)
    VKAS_CHECK(e->patch > 0 && e->patch_Cp % 8 == 0 && e->patch_Cp > 0 && e->patch * e->patch * e->patch_Cp == Np &&
                   e->ldo >= e->patch_Cp && e->patch_Hs > 0 && e->patch_Ws > 0,
               "%s: bad patch scatter geometry", who);
  return VKAS_OK;
}

extern "C" int vkas_conv_gemm_fwd(const void* x, const vkas_conv_geom* g, const void* Bw, int Np,
                                  const vkas_epilogue* epi, int dtype, void* stream) {
  int rc = vkas_check_geom("vkas_conv_gemm_fwd", x, g, Np);
  if (rc) return rc;
  rc = vkas_check_epilogue("vkas_conv_gemm_fwd", epi, Np);
  if (rc) return rc;
  VKAS_CHECK(Bw && vkas_aligned16(Bw), "vkas_conv_gemm_fwd: weights null/misaligned");
  if (epi->mode == VKAS_EPI_PATCH)
    VKAS_CHECK((long)g->B * g->Hout * g->Wout == (long)g->B * epi->patch_Hs * epi->patch_Ws || g->B == 0,
               "vkas_conv_gemm_fwd: patch grid does not match the GEMM rows");
  if (epi->mode == VKAS_EPI_HEAD)
    VKAS_CHECK((dtype == VKAS_BF16 || dtype == VKAS_F16) && !force_simple() && (long)g->B * g->Hout * g->Wout >= 16384,
               "vkas_conv_gemm_fwd: the fused head epilogue exists for the 16-bit MFMA kernels (M >= 16384) only");
  if (dtype == VKAS_BF16 && !force_simple()) return vkas_gemm_nt_mfma_bf16(x, g, Bw, Np, epi, vkas_stream(stream));
  if (dtype == VKAS_F16 && !force_simple()) return vkas_gemm_nt_mfma_f16(x, g, Bw, Np, epi, vkas_stream(stream));
  return vkas_gemm_nt_simple(x, g, Bw, Np, epi, dtype, vkas_stream(stream));
}

// Which kernel instantiation a bf16 call with these sizes runs (profiling aid; the names match rocprofv3's).
extern "C" int vkas_conv_gemm_tile(int wgrad, long M, int Np, int K) {
  if (force_simple()) return 0;
  return wgrad ? vkas_gemm_tn_tile_choice(M, Np, K) : vkas_gemm_nt_tile_choice(M, Np);
}

// Which kernel a bf16 call with this geometry runs: 0 plain fp32-FMA kernels forced; fwd: 1 = 128x128, 12 / 13 / 14 =
// gemm_nt_ring_kernel<2 / 3 / 4>, 128 / 192 / 224 =
// N extent of the generic 256-row tile, 1000 + TN = conv3x3_slab_mfma_kernel<TN, .> (TN = 4, 6, 7); wgrad: 128 / 192 /
// 224 generic, 2000 + TNn = conv3x3_wgrad_slab_kernel<TNn>.  head_width > 0: a fused-head launch whose widest head has
// that many columns.
extern "C" int vkas_conv_gemm_kernel_id(int wgrad, const vkas_conv_geom* g, int Np, long lddy, int head_width) {
  if (force_simple() || !g) return 0;
  const long M = (long)g->B * g->Hout * g->Wout;
  const int K = g->KH * g->KW * g->Cp;
  if (wgrad) {
    if (vkas_tn_slab_eligible(g, Np, lddy)) return 2000 + (vkas_tn_slab_n96(Np) ? 6 : (vkas_tn_slab_n112(Np) ? 7 : 8));
    return vkas_gemm_tn_tile_choice(M, Np, K);
  }
  int choice = vkas_gemm_nt_tile_choice(M, Np);
  if (head_width > 0) choice = head_width <= 128 ? 128 : (head_width <= 192 ? 192 : 224);
  if (choice != 1 && vkas_nt_slab_eligible(g, Np)) return 1000 + choice / 32;
  if (choice == 1 && head_width <= 0 && vkas_gemm_nt_ring_stages(g, Np) > 0) return 10 + vkas_gemm_nt_ring_stages(g, Np);
  return choice;
}

static int conv_gemm_wgrad(const char* who, const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np,
                           float* gw, float* gb, int x_gelu, int dtype, void* stream) {
  int rc = vkas_check_geom(who, x, g, Np);
  if (rc) return rc;
  VKAS_CHECK(dy && vkas_aligned16(dy) && lddy >= Np && lddy % 8 == 0, "%s: bad dy (lddy=%ld)", who, lddy);
  VKAS_CHECK(gw, "%s: null gw", who);
  if (dtype == VKAS_BF16 && !force_simple())
    return vkas_gemm_tn_mfma_bf16(x, g, dy, lddy, Np, gw, gb, x_gelu, vkas_stream(stream));
  if (dtype == VKAS_F16 && !force_simple())
    return vkas_gemm_tn_mfma_f16(x, g, dy, lddy, Np, gw, gb, x_gelu, vkas_stream(stream));
  return vkas_gemm_tn_simple(x, g, dy, lddy, Np, gw, gb, x_gelu, dtype, vkas_stream(stream));
}

extern "C" int vkas_conv_gemm_wgrad(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np,
                                    float* gw, float* gb, int dtype, void* stream) {
  return conv_gemm_wgrad("vkas_conv_gemm_wgrad", x, g, dy, lddy, Np, gw, gb, 0, dtype, stream);
}

extern "C" int vkas_conv_gemm_wgrad_ordered(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np,
                                            float* gw, int dtype, void* stream) {
  return conv_gemm_wgrad("vkas_conv_gemm_wgrad_ordered", x, g, dy, lddy, Np, gw, nullptr, 2, dtype, stream);
}

extern "C" int vkas_conv_gemm_wgrad_gelu(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np,
                                         float* gw, float* gb, int dtype, void* stream) {
  return conv_gemm_wgrad("vkas_conv_gemm_wgrad_gelu", x, g, dy, lddy, Np, gw, gb, 1, dtype, stream);
}

// ---- column sums ------------------------------------------------------------------------------------
namespace {
static inline long cs_rows_per_block(long M) { long r = vkas_cdiv(M > 0 ? M : 1, 1024); return r < 32 ? 32 : r; }

// partial[blk][c] = sum over this block's rows of y[m][c]; thread = one 8-channel vector, rows strided.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ y, long ld, long M, int Np,
                                                             long rows_per_block, float* __restrict__ partial,
                                                             int ldp) {
  const int nvec = Np >> 3;
  const int lanes_r = 256 / nvec > 0 ? 256 / nvec : 1;  // row lanes per block (nvec <= 256 checked by host)
  const int v = threadIdx.x % nvec;
  const int rl = threadIdx.x / nvec;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long mbeg = (long)blockIdx.x * rows_per_block;
  const long mend = mbeg + rows_per_block < M ? mbeg + rows_per_block : M;
  if (rl < lanes_r) {
    for (long m = mbeg + rl; m < mend; m += lanes_r) {
      float t[8];
      load8(y + m * ld + v * 8, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += t[i];
    }
  }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
  __syncthreads();
  if (rl == 0) {
    for (int r = 1; r < lanes_r; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += red[(r * nvec + v) * 8 + i];
    float* dst = partial + (long)blockIdx.x * ldp + v * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = acc[i];
  }
}
}  // namespace

// out[c] (+)= sum_p partial[p][c]; 16 columns x 16 row lanes per block, fixed summation order => deterministic.
__global__ __launch_bounds__(256) void vkas_colreduce_finalize_kernel(const float* __restrict__ partial, long P, int n,
                                                                      int ldp, float* __restrict__ out,
                                                                      int accumulate) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  // 8 independent partial sums per lane: with two the loop was a chain of dependent L2 round trips (9 us per launch,
  // 118 launches per step)
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (c < n) {
    long p = rl;
    for (; p + 7 * 16 < P; p += 8 * 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += partial[(p + 16 * i) * ldp + c];
    }
    for (; p < P; p += 16) acc[0] += partial[p * ldp + c];
  }
  red[rl][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (rl == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][cl];
    out[c] = accumulate ? out[c] + s : s;
  }
}

// Up to 8 column reductions in one launch (the three second-stage sums of a ConvNeXt layer's backward + their delivery into
// the flat gradient views used to be 3 finalize launches + 1 accumulate launch per layer): block b belongs to reduction k with
// first[k] <= b < first[k + 1] and sums 16 columns of it exactly as vkas_colreduce_finalize_kernel does.
struct FinalizeMany {
  const float* partial[8];
  float* out[8];
  long P[8];
  int n[8], ldp[8], accumulate[8], first[9];
};
__global__ __launch_bounds__(256) void vkas_finalize_many_kernel(FinalizeMany a) {
  __shared__ float red[16][17];
  int k = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) k += ((int)blockIdx.x >= a.first[i]) ? 1 : 0;
  const float* __restrict__ partial = a.partial[k];
  const long P = a.P[k];
  const int n = a.n[k], ldp = a.ldp[k];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = ((int)blockIdx.x - a.first[k]) * 16 + cl;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (c < n) {
    long p = rl;
    for (; p + 7 * 16 < P; p += 8 * 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += partial[(p + 16 * i) * ldp + c];
    }
    for (; p < P; p += 16) acc[0] += partial[p * ldp + c];
  }
  red[rl][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (rl == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][cl];
    float* out = a.out[k];
    out[c] = a.accumulate[k] ? out[c] + s : s;
  }
}

extern "C" int vkas_finalize_many(int count, const float* const* partial, const long* P, const int* n, const int* ldp,
                                  float* const* out, const int* accumulate, void* stream) {
  VKAS_CHECK(count >= 0 && count <= 8 && (count == 0 || (partial && P && n && ldp && out && accumulate)),
             "vkas_finalize_many: bad arguments (count=%d)", count);
  if (count == 0) return VKAS_OK;
  FinalizeMany a;
  int blocks = 0;
  for (int k = 0; k < 8; ++k) {
    a.first[k] = blocks;
    if (k < count) {
      VKAS_CHECK(partial[k] && out[k] && P[k] >= 0 && n[k] >= 0 && ldp[k] >= n[k], "vkas_finalize_many: bad reduction %d", k);
      a.partial[k] = partial[k];
      a.out[k] = out[k];
      a.P[k] = P[k];
      a.n[k] = n[k];
      a.ldp[k] = ldp[k];
      a.accumulate[k] = accumulate[k];
      blocks += (int)vkas_cdiv(n[k], 16);
    } else {
      a.partial[k] = nullptr; a.out[k] = nullptr; a.P[k] = 0; a.n[k] = 0; a.ldp[k] = 0; a.accumulate[k] = 0;
    }
  }
  a.first[8] = blocks;
  for (int k = count; k < 8; ++k) a.first[k] = blocks;  // no block maps to an unused slot
  if (blocks == 0) return VKAS_OK;
  vkas_finalize_many_kernel<<<(unsigned)blocks, 256, 0, vkas_stream(stream)>>>(a);
  VKAS_LAUNCH_CHECK("finalize_many");
  return VKAS_OK;
}

int vkas_colreduce_finalize(const float* partial, long P, int n, int ldp, float* out, int accumulate,
                            hipStream_t st) {
  vkas_colreduce_finalize_kernel<<<(unsigned)vkas_cdiv(n, 16), 256, 0, st>>>(partial, P, n, ldp, out, accumulate);
  VKAS_LAUNCH_CHECK("colreduce_finalize");
  return VKAS_OK;
}

extern "C" size_t vkas_colsum_ws_bytes(long M, int Np) {
  return (size_t)vkas_cdiv(M > 0 ? M : 1, cs_rows_per_block(M)) * (size_t)Np * sizeof(float);
}

extern "C" int vkas_colsum(const void* y, long ld, long M, int Np, float* out, int accumulate, float* ws,
                           size_t ws_bytes, int dtype, void* stream) {
  VKAS_CHECK(y && out && ws, "vkas_colsum: null pointer");
  VKAS_CHECK(Np > 0 && Np % 8 == 0, "vkas_colsum: Np=%d must be a positive multiple of 8", Np);
  VKAS_CHECK(ld >= Np && ld % 8 == 0 && vkas_aligned16(y), "vkas_colsum: bad ld/alignment");
  VKAS_CHECK(ws_bytes >= vkas_colsum_ws_bytes(M, Np), "vkas_colsum: workspace too small");
  if (M <= 0) {
    if (!accumulate) (void)hipMemsetAsync(out, 0, Np * sizeof(float), vkas_stream(stream));
    return VKAS_OK;
  }
  const long rpb = cs_rows_per_block(M);
  const long P = vkas_cdiv(M, rpb);
  VKAS_DISPATCH_DTYPE(dtype, "vkas_colsum", {
    for (int c0 = 0; c0 < Np; c0 += 2048) {  // 256 threads cover at most 256 8-channel vectors per pass
      const int n = Np - c0 < 2048 ? Np - c0 : 2048;
      colsum_partial_kernel<T><<<(unsigned)P, 256, 0, vkas_stream(stream)>>>((const T*)y + c0, ld, M, n, rpb, ws + c0, Np);
    }
  })
  VKAS_LAUNCH_CHECK("colsum_partial");
  return vkas_colreduce_finalize(ws, P, Np, Np, out, accumulate, vkas_stream(stream));
}
