/*
 * vkas.h — C ABI of libvkas.so: the MI355X (gfx950) kernels behind the adaptive-scaling
 * text-detection forward/backward path.
 *
 * The reference (vkit_open_model, pure PyTorch) has no FFI boundary of its own; its boundary for
 * this path is the nn.Module / loss-callable API.  Each entry point below replaces
 * the ATen op(s) that a reference call site dispatches to; the file:line given is that call site
 * (paths relative to /root/reference/vkit_open_model/).  The Python host layer
 * (vkit_ocr_model_adaptive_scaling_amd/) binds these through ctypes and mirrors the reference's
 * module API on top.  Nothing here depends on torch: plain pointers, sizes, a HIP stream.
 *
 * Conventions
 *  - every function returns 0 on success, a negative VKAS_E_* code otherwise (never aborts);
 *    vkas_last_error() returns a thread-local message for the last failure.
 *  - `dtype` selects the storage type of activations: VKAS_F32, VKAS_BF16 or VKAS_F16.  Accumulation is
 *    always fp32.  Parameters handed over in the reference's layout are always fp32.
 *  - activations are NHWC: element (b, y, x, c) of a tensor with pixel stride `ld` (elements) lives
 *    at base[((b*H + y)*W + x)*ld + c].  `ld >= C`, `ld % 8 == 0`, bases 16-byte aligned; a channel
 *    slice of a wider tensor is expressed by offsetting the base and keeping the wide `ld`
 *    (this is how torch.cat along C, upernext.py:82,197 / fpn.py:144, disappears).
 *  - logical channel counts are padded to a multiple of 8 (`Cp`); pad channels hold zeros.
 *  - all kernels are enqueued on `stream` (a hipStream_t passed as void*); no call synchronises,
 *    allocates or frees device memory.  Workspaces are caller-provided.
 */
#ifndef VKAS_H
#define VKAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKAS_F32 0
#define VKAS_BF16 1
#define VKAS_F16 2 /* IEEE half storage, fp32 accumulate: BASELINE.json configs[4] (fp16 inference of the Base model) */

#define VKAS_OK 0
#define VKAS_E_ARG (-1)     /* bad shape / alignment / dtype */
#define VKAS_E_LAUNCH (-2)  /* hipGetLastError() != hipSuccess after a launch */

/* GEMM epilogues (vkas_conv_gemm_fwd) */
#define VKAS_EPI_NONE 0       /* out = acc + bias */
#define VKAS_EPI_GELU 1       /* out = acc + bias (pre-activation; NULL = not kept), out2 = gelu(out)  helper.py:100 */
#define VKAS_EPI_SCALE_RES 2  /* out2 = acc + bias; out = aux + rowscale[b]*colscale[n]*out2  convnext.py:56-58 */
#define VKAS_EPI_DGELU 3      /* out = (acc) * gelu'(aux)                                     backward of helper.py:100 */
#define VKAS_EPI_ADD 4        /* out = acc + bias + aux                                       gradient accumulation */
#define VKAS_EPI_PATCH 5      /* out scattered to non-overlapping kxk patches: dgrad of helper.py:43-58 */
#define VKAS_EPI_HEAD 6       /* fused head tail: out = z = acc + bias (pre-LN, kept for backward); per pixel
                               * LayerNorm -> GELU -> Linear(C -> 1..4) in the epilogue; the (M, C) activation never
                               * reaches HBM.  upernext.py:215-223 / fpn.py:165-183.  16-bit MFMA kernels only.  out = NULL
                               * together with head.stats = NULL: inference, z and the row statistics are not stored. */

const char* vkas_last_error(void);
int vkas_abi_version(void);

/* Geometry of a convolution seen as an implicit GEMM: M = B*Hout*Wout rows, K = KH*KW*Cp columns.
 * 1x1 / nn.Linear is KH=KW=1, stride 1, pad 0 (helper.py:18-22); 3x3 is pad 1 (helper.py:25-31);
 * the patchify convs are KH=KW=stride, pad 0 (helper.py:43-58). */
typedef struct vkas_conv_geom {
  int B, Hin, Win, Hout, Wout;
  int Cp;      /* input channels per tap, padded to a multiple of 8 */
  int ldx;     /* input pixel stride in elements */
  int KH, KW, stride, pad;
} vkas_conv_geom;

/* VKAS_EPI_HEAD: the N tiles of the launch are the heads (all reading the same input).  Head h owns output columns
 * [n0[h], n0[h] + np[h]) (np = rup8(c)), of which c are real; params + h*(6*pw+8) holds, as floats,
 * gamma[pw] | beta[pw] | Wproj[4][pw] | bproj[4] | pad[4] (zero padded; vkas_pack_head_params).
 * stats: (n_heads, M, 2) fp32 (mean, rstd); proj: (n_heads, M, 8) fp32, columns >= oc are zero. */
typedef struct vkas_head_desc {
  int n_heads, pw;
  int n0[4], np[4], c[4], oc[4];
  const float* params;
  float* stats;
  float* proj;
} vkas_head_desc;

typedef struct vkas_epilogue {
  int mode;               /* VKAS_EPI_* */
  const float* bias;      /* [Np] or NULL */
  void* out;  long ldo;   /* primary output, row stride (elements) */
  void* out2; long ldo2;  /* secondary output (GELU, SCALE_RES) or NULL */
  const void* aux; long ldaux; /* residual / saved pre-activation, same dtype as out */
  const float* colscale;  /* [Np]   SCALE_RES: block_scale */
  const float* rowscale;  /* [B] or NULL  SCALE_RES: stochastic-depth keep mask / keep prob */
  int rows_per_image;     /* Hout*Wout, to index rowscale */
  int patch;              /* PATCH: patch edge k; rows are the (B,Hs,Ws) grid, columns (ky,kx,c) */
  int patch_Hs, patch_Ws, patch_Cp;
  vkas_head_desc head;    /* HEAD */
} vkas_epilogue;

/* ---- parameter layout conversion (reference layouts -> kernel layouts and back) ------------------ */
/* w (N,C,KH,KW) fp32 [nn.Conv2d / nn.Linear weight] -> out (Np, KH, KW, Cp) in `dtype`, zero padded.
 * mode 0: forward operand.  mode 1: dgrad operand of a stride-1 conv: out (Cp_as_rows..) = W flipped and
 * in/out swapped: out[c][KH-1-ky][KW-1-kx][n] = w[n][c][ky][kx]  (rows = Cp(C), columns (ky,kx,Np)).
 * mode 2: dgrad operand of a patchify conv: out[(ky,kx,c)][n] = w[n][c][ky][kx] (rows KH*KW*Cp, cols Np). */
int vkas_pack_conv_weight(const float* w, void* out, int N, int C, int KH, int KW, int Np, int Cp, int mode,
                          int dtype, void* stream);
/* the same for one of several convolutions that share their input and are packed side by side (the heads of a pass,
 * adaptive_scaling.py:150-152,163-170): fills output channels [n_off, n_off + Np) of an operand with Nt output channels
 * in total; mode 0 (forward) or 1 (dgrad). */
int vkas_pack_conv_weight_slice(const float* w, void* out, int N, int C, int KH, int KW, int Np, int Cp, int mode,
                                int n_off, int Nt, int dtype, void* stream);
/* gw (Np, KH, KW, Cp) fp32 [wgrad output] -> grad (N,C,KH,KW) fp32, grad += (accumulate != 0) or = */
int vkas_unpack_conv_wgrad(const float* gw, float* grad, int N, int C, int KH, int KW, int Np, int Cp,
                           int accumulate, void* stream);
/* dst[k][0..n[k]) += src[k][0..n[k]) for count <= 16 fp32 vectors in one launch: the small per-parameter gradients of a
 * layer (bias, LayerNorm affine, block_scale; what autograd's AccumulateGrad does one launch per tensor) */
int vkas_accumulate_many(int count, const float* const* src, float* const* dst, const int* n, void* stream);
/* Up to 8 second-stage column sums in one launch: out[k][c] (+)= sum over the P[k] partial rows (row pitch ldp[k]) of column c <
   n[k].  vkas_layernorm_bwd / vkas_scale_res_bwd / vkas_dwconv7x7_wgrad called with NULL gradient outputs leave their
   per-workgroup partial rows in the workspace (vkas_*_parts() rows); the backward of a ConvNeXt layer (convnext.py:29-59) then
   sums all of them - straight into the parameters' gradient views - with this one launch.  Fixed order: deterministic. */
int vkas_finalize_many(int count, const float* const* partial, const long* P, const int* n, const int* ldp,
                       float* const* out, const int* accumulate, void* stream);
long vkas_layernorm_bwd_parts(long M, int Cp);
long vkas_scale_res_bwd_parts(long M, int Cp);
long vkas_dwconv7x7_wgrad_parts(int B, int H, int W, int Cp, int dtype);
/* v (n) fp32 -> out (np) fp32 zero padded (bias, LayerNorm affine, block_scale) */
int vkas_pad_vector(const float* v, float* out, int n, int np, void* stream);
/* depthwise weight (C,1,7,7) fp32 -> vkas_dw_weight_elems(Cp) fp32: (49, Cp) taps x channels followed by the same values
 * as (Cp/2, 7, 16) channel-pair kernel rows (scalar-load operand of the bf16 kernel); flip != 0 rotates the taps by 180
 * degrees (dgrad operand) */
size_t vkas_dw_weight_elems(int Cp);
int vkas_pack_dw_weight(const float* w, float* out, int C, int Cp, int flip, void* stream);
/* gw (49, Cp) fp32 -> grad (C,1,7,7) fp32 (+=) */
int vkas_unpack_dw_wgrad(const float* gw, float* grad, int C, int Cp, int accumulate, void* stream);
/* image (B,3,H,W) fp32 NCHW [dataset/adaptive_scaling.py:296] -> (B,H,W,8) `dtype`, channels 3..7 zero */
int vkas_image_nchw_to_nhwc8(const float* img, void* out, int B, int C, int H, int W, int dtype, void* stream);
/* small head outputs: (B,H,W,ld) `dtype` -> (B,C,H,W) fp32 and the reverse (gradient) direction */
int vkas_nhwc_to_nchw_f32(const void* x, long ld, float* out, int B, int H, int W, int C, int dtype, void* stream);
int vkas_nchw_f32_to_nhwc(const float* g, void* out, long ld, int B, int H, int W, int C, int Cp, int dtype, void* stream);

/* ---- implicit-GEMM convolution: helper.py:18-58 (conv1x1 / conv3x3 / pconv2x2 / pconv4x4) -------- */
/* D[m][n] = epilogue( sum_k A(m,k) * Bw[n][k] ), A gathered from x by `g`; Bw (Np, K) packed `dtype`. */
int vkas_conv_gemm_fwd(const void* x, const vkas_conv_geom* g, const void* Bw, int Np, const vkas_epilogue* epi,
                       int dtype, void* stream);
/* wgrad: gw[n][k] += sum_m dy[m][n] * A(m,k), gw (Np, K) fp32; optional fused bias gradient gb[n] += sum_m dy[m][n]
 * (gb (Np) fp32 or NULL).  gw and gb must be zero-filled by the caller: the kernel adds split-M partials with
 * fp32 atomics. */
int vkas_conv_gemm_wgrad(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, float* gw,
                         float* gb, int dtype, void* stream);
/* the same without the split over M (no bias gradient): every gw tile is summed over all rows by one workgroup, in row
 * order, and added once to the zero-filled gw, so the fp32 result does not depend on the order workgroups run in.  For
 * the one use whose result is rounded into a 16-bit activation gradient afterwards (the input gradient of the heads the
 * loss reads at label points, loss_function/adaptive_scaling.py:235-262 -> upernext.py:215-223): there the last-bit
 * spread of atomically added partials flips 16-bit roundings from run to run and the backbone's backward amplifies it. */
int vkas_conv_gemm_wgrad_ordered(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, float* gw,
                                 int dtype, void* stream);
/* the same with A(m,k) = gelu(x(m,k)): weight gradient of the second MLP Linear when the forward kept only the
 * pre-activation h (vkas_mlp_chain_fwd): backward of convnext.py:34-35 */
int vkas_conv_gemm_wgrad_gelu(const void* x, const vkas_conv_geom* g, const void* dy, long lddy, int Np, float* gw,
                              float* gb, int dtype, void* stream);

/* ---- fused ConvNeXt MLP (convnext.py:33-35,54-58), C % 8 == 0, C <= 512, 16-bit storage --------------------------------------
 * Linear(C,4C) -> GELU -> Linear(4C,C) -> layer scale -> stochastic depth -> residual as one kernel per direction; the
 * (M, 4C) activation is written once (h, for backward) and never read back between the two matrix products.
 * Weights are consumed as a packed, pre-swizzled LDS image: vkas_mlp_chain_image_elems(C) elements of `dtype`, built by
 * vkas_mlp_chain_pack from w1 (4C, C) / w2 (C, 4C) fp32 in the reference layout; mode 0 = forward image, 1 = backward
 * image (transposed roles; b1 may be NULL).  image_elems returns 0 when C is not covered. */
size_t vkas_mlp_chain_image_elems(int C);
int vkas_mlp_chain_pack(const float* w1, const float* w2, const float* b1, int C, int mode, void* img, int dtype,
                        void* stream);
/* h = yn W1^T + b1 (stored; b1 (4C) travels inside the forward image); z = gelu(h) W2^T + b2 (stored);
 * out = x + rowscale[m / rows_per_image] * colscale * z.  b2 (C), colscale (C) fp32; rowscale (images) fp32 or NULL.
 * h = z = NULL: inference (no-grad) call, nothing is stored for a backward pass. */
int vkas_mlp_chain_fwd(const void* yn, long ldyn, const void* img, const float* b2, const void* x, long ldx,
                       const float* colscale, const float* rowscale, int rows_per_image, void* h, long ldh, void* z,
                       long ldz, void* out, long ldo, long M, int C, int dtype, void* stream);
/* the same with the LayerNorm in front of the MLP (convnext.py:32, helper.py:96-101; eps 1e-6) applied to the rows on their
 * way into the first matrix product: y = the depthwise output (M, C), ln_gamma / ln_beta (C) fp32.  Training: the
 * normalised rows yn (operand of the W1 weight gradient) and stats (M, 2) fp32 = mean | rstd per row (what
 * vkas_layernorm_fwd writes, consumed by vkas_layernorm_bwd) are stored; inference: yn = stats = h = z = NULL and the
 * normalised rows never reach memory.  Replaces vkas_layernorm_fwd + vkas_mlp_chain_fwd of a ConvNeXt layer. */
int vkas_mlp_chain_ln_fwd(const void* y, long ldy, const float* ln_gamma, const float* ln_beta, void* yn, long ldyn,
                          float* stats, const void* img, const float* b2, const void* x, long ldx, const float* colscale,
                          const float* rowscale, int rows_per_image, void* h, long ldh, void* z, long ldz, void* out,
                          long ldo, long M, int C, int dtype, void* stream);
/* dh = (dz W2) * gelu'(h) (stored, operand of the W1 weight gradient); dyn = dh W1.  img_t = the mode-1 image. */
int vkas_mlp_chain_bwd(const void* dz, long lddz, const void* img_t, const void* h, long ldh, void* dh, long lddh,
                       void* dyn, long lddyn, long M, int C, int dtype, void* stream);
/* profiling aid: tile configuration a bf16 call of these sizes runs.  fwd (wgrad == 0): 1 = 128x128 (4 waves), else the
 * N extent 128 / 192 / 224 of the 256-row 8-wave tile; wgrad: N extent 128 (4 waves), 192 / 224 (8 waves, 256 K columns) or 384 (8 waves, 128 K columns); 0 when the
 * plain fp32-FMA kernels are forced (VKAS_GEMM=simple). */
int vkas_conv_gemm_tile(int wgrad, long M, int Np, int K);
/* profiling aid: the kernel a bf16 call with this geometry runs.  0 = plain fp32-FMA kernels forced.  fwd: 1 / 128 / 192 /
 * 224 as above, 1000 + TN = the 3x3 row-slab kernel conv3x3_slab_mfma_kernel<TN, .> (TN = 4, 6, 7); wgrad: 128 / 192 /
 * 224, 2000 + TNn = conv3x3_wgrad_slab_kernel<TNn> (7, 8).  head_width > 0 for a fused-head launch (widest head). */
int vkas_conv_gemm_kernel_id(int wgrad, const vkas_conv_geom* g, int Np, long lddy, int head_width);
/* column sums: out[n] (+)= sum_m y[m][n]   (bias gradients) */
int vkas_colsum(const void* y, long ld, long M, int Np, float* out, int accumulate, float* ws, size_t ws_bytes,
                int dtype, void* stream);
size_t vkas_colsum_ws_bytes(long M, int Np);

/* ---- depthwise 7x7: helper.py:61-73 @ convnext.py:30 ------------------------------------------- */
/* y = dw7x7(x; w) + bias (+ addend).  w: the vkas_dw_weight_elems(Cp) floats written by vkas_pack_dw_weight. */
int vkas_dwconv7x7_fwd(const void* x, long ldx, const float* w, const float* bias, const void* addend, long ldadd,
                       void* y, long ldy, int B, int H, int W, int Cp, int dtype, void* stream);
/* gw (49, Cp) and gb (Cp) fp32, overwritten */
int vkas_dwconv7x7_wgrad(const void* x, long ldx, const void* dy, long lddy, float* gw, float* gb, float* ws,
                         size_t ws_bytes, int B, int H, int W, int Cp, int dtype, void* stream);
size_t vkas_dwconv7x7_wgrad_ws_bytes(int B, int H, int W, int Cp);

/* ---- LayerNorm over C (+ optional GELU): helper.py:96-101 --------------------------------------- */
/* y = act(LN(x) * gamma + beta); stats (M, 2) fp32 = (mean, rstd).  C = logical channels, pad channels -> 0.
 * gamma / beta: Cp floats, zero padded beyond C (vkas_pad_vector). */
int vkas_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                       float* stats, long M, int C, int Cp, int act_gelu, int dtype, void* stream);
/* dx from dy; dgamma/dbeta (Cp) fp32 overwritten.  `x` is the forward input, `stats` from forward. */
int vkas_layernorm_bwd(const void* x, long ldx, const float* gamma, const float* beta, const float* stats,
                       const void* dy, long lddy, void* dx, long lddx, float* dgamma, float* dbeta, float* ws,
                       size_t ws_bytes, long M, int C, int Cp, int act_gelu, int dtype, void* stream);
size_t vkas_layernorm_bwd_ws_bytes(long M, int Cp);

/* ---- fused head tail (LayerNorm -> GELU -> Linear(C -> oc), oc <= 4): upernext.py:215-223, fpn.py:165-183 ------ */
/* gamma/beta (C), wproj (oc, C), bproj (oc) fp32 -> out[6*pw + 8] as laid out in vkas_head_desc */
int vkas_pack_head_params(const float* gamma, const float* beta, const float* wproj, const float* bproj, int C, int oc,
                          int pw, float* out, void* stream);
/* The same head tail as a pass of its own, for heads wider than the 224 columns the GEMM epilogue holds (ConvNeXt-Base /
 * Large: model/upernext.py:215-223 with 256-258 / 384+ channels; pw <= 512): z (M, ld ldz) = the convolution's output incl.
 * bias, written by a VKAS_EPI_NONE launch; hd->proj (n_heads, M, 8) and hd->stats (n_heads, M, 2; NULL = not kept) as
 * VKAS_EPI_HEAD writes them. */
int vkas_head_tail_fwd(const void* z, long ldz, const vkas_head_desc* hd, long M, int dtype, void* stream);
/* backward of the fused tail for all heads of the launch at once: z / dz are the shared (M, sum np) buffers (strides ldz,
 * lddz), hd the descriptor used in the forward (params, stats; proj unused), dproj[h] the (M, 8) fp32 gradients of head
 * h's projection outputs -> dz and dparams[n_heads][6*pw + 8] (same layout as the packed parameters). */
int vkas_head_tail_bwd(const void* z, long ldz, const vkas_head_desc* hd, const float* const* dproj, void* dz, long lddz,
                       float* dparams, float* ws, size_t ws_bytes, long M, int dtype, void* stream);
size_t vkas_head_tail_bwd_ws_bytes(long M, int pw);

/* ---- block_scale / stochastic depth backward: convnext.py:56-58 ----------------------------------- */
/* dz = dout * rowscale[b] * colscale[c]; dscale[c] = sum_m dout*rowscale[b]*z; dbias2[c] = sum_m dz */
int vkas_scale_res_bwd(const void* dout, long lddo, const void* z, long ldz, const float* colscale,
                       const float* rowscale, int rows_per_image, void* dz, long lddz, float* dscale, float* dbias,
                       float* ws, size_t ws_bytes, long M, int Cp, int dtype, void* stream);
size_t vkas_scale_res_bwd_ws_bytes(long M, int Cp);

/* ---- resize: F.interpolate bilinear (upernext.py:79,178-195,237-244), nearest (fpn.py:125-142,197-204) */
/* mode 0 bilinear (align_corners=False), 1 nearest.  accumulate != 0: y += resize(x) (top-down add). */
int vkas_resize_fwd(const void* x, long ldx, void* y, long ldy, int B, int Hin, int Win, int Hout, int Wout,
                    int Cp, int mode, int accumulate, int dtype, void* stream);
/* dx (+)= resize^T(dy) */
int vkas_resize_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int Hin, int Win, int Hout, int Wout,
                    int Cp, int mode, int accumulate, int dtype, void* stream);
/* the same with a workspace (fp32, vkas_resize_bwd_ws_bytes; 0 = none needed): large ratios - the necks' x4 / x8 resize to
   level-0 size, upernext.py:191-195 - then run as two separable gathers (along x into the workspace, then along y) */
size_t vkas_resize_bwd_ws_bytes(int B, int Hin, int Win, int Hout, int Wout, int Cp);
int vkas_resize_bwd_ws(const void* dy, long lddy, void* dx, long lddx, float* ws, size_t ws_bytes, int B, int Hin, int Win,
                       int Hout, int Wout, int Cp, int mode, int accumulate, int dtype, void* stream);

/* ---- nn.AdaptiveAvgPool2d(s): upernext.py:62 ------------------------------------------------------ */
int vkas_adaptive_avgpool_fwd(const void* x, long ldx, void* y, long ldy, int B, int H, int W, int s, int Cp,
                              int dtype, void* stream);
int vkas_adaptive_avgpool_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int H, int W, int s, int Cp,
                              int accumulate, int dtype, void* stream);

/* ---- elementwise ------------------------------------------------------------------------------------ */
/* y[m][0..Cp) = x[m][0..Cp) (+ y if accumulate): channel-slice copies for concat / its backward */
int vkas_copy_channels(const void* x, long ldx, void* y, long ldy, long M, int Cp, int accumulate, int dtype,
                       void* stream);
/* y[m][c_dst + c] = x[m][c_src + c] for c in [0, C), then zero_tail zero channels behind them; channel offsets are
   element-granular (no 8-channel alignment): torch.cat of parts whose widths are not multiples of 8 - e.g.
   FpnNeck(..., out_channels=400) -> 4 x 100 channels, fpn.py:75,144; upernext.py:82,144,197 - and its backward */
int vkas_copy_channel_range(const void* x, long ldx, int c_src, void* y, long ldy, int c_dst, long M, int C,
                            int zero_tail, int dtype, void* stream);
/* nn.Softplus() on fp32 maps: adaptive_scaling.py:101,140 */
int vkas_softplus_fwd(const float* x, float* y, long n, void* stream);
int vkas_softplus_bwd(const float* x, const float* dy, float* dx, long n, void* stream);

/* ---- dense losses: loss_function/adaptive_scaling.py -------------------------------------------------- */
typedef struct vkas_rough_loss_cfg {
  float focal_factor, dice_factor, l1_factor, score_min, height_min; /* :29-35 */
  float focal_alpha, focal_gamma;                                    /* focal_with_logits.py:21-23 */
  float out_scale;                                                   /* train.py:413 (1/2), x 1/world */
} vkas_rough_loss_cfg;
/* mask_feat/height_feat (B,1,H,W) fp32; gt_mask/gt_score (B,CH,CW) fp32; box = up,left of the crop.
 * sums: 8 doubles workspace kept for backward; loss: 1 float (already multiplied by out_scale). */
int vkas_rough_loss_fwd(const float* mask_feat, const float* height_feat, const float* gt_mask,
                        const float* gt_score, int B, int H, int W, int up, int left, int CH, int CW,
                        const vkas_rough_loss_cfg* cfg, double* sums, float* loss, void* stream);
int vkas_rough_loss_bwd(const float* mask_feat, const float* height_feat, const float* gt_mask,
                        const float* gt_score, int B, int H, int W, int up, int left, int CH, int CW,
                        const vkas_rough_loss_cfg* cfg, const double* sums, const float* dloss, float* d_mask_feat,
                        float* d_height_feat, void* stream);

typedef struct vkas_precise_loss_cfg {
  float pos_l2, neg_l2, offset_l1, reg_l1, angle_ce, dist_l1, loss_factor; /* :136-145 */
  float smooth_beta;                                                       /* 2.5, :159-165 */
  float out_scale;
} vkas_precise_loss_cfg;
/* margin[0] = min over the n label points of min(py, H-1-py, px, W-1-px): negative iff a point lies outside the (H, W) map.
 * The reference's advanced indexing (loss_function/adaptive_scaling.py:235-262) raises for such a point; the host reads this
 * value asynchronously instead of synchronising inside the step. */
int vkas_points_margin(const int64_t* py, const int64_t* px, long n, int H, int W, int64_t* margin, void* stream);
/* prob (B,1,H,W), offset (B,2,H,W), angle (B,4,H,W), dist (B,4,H,W) fp32 NCHW; py/px (B,P) int64;
 * gt_offsets (B,P,2), gt_angles (B,P,4), gt_dists (B,P,3) fp32. */
int vkas_precise_loss_fwd(const float* prob, const float* offset, const float* angle, const float* dist,
                          const float* gt_score, const float* gt_mask, const int64_t* py, const int64_t* px,
                          const float* gt_offsets, const float* gt_angles, const float* gt_dists, int B, int H,
                          int W, int up, int left, int CH, int CW, int P, const vkas_precise_loss_cfg* cfg,
                          double* sums, float* loss, void* stream);
int vkas_precise_loss_bwd(const float* prob, const float* offset, const float* angle, const float* dist,
                          const float* gt_score, const float* gt_mask, const int64_t* py, const int64_t* px,
                          const float* gt_offsets, const float* gt_angles, const float* gt_dists, int B, int H,
                          int W, int up, int left, int CH, int CW, int P, const vkas_precise_loss_cfg* cfg,
                          const double* sums, const float* dloss, float* d_prob, float* d_offset, float* d_angle,
                          float* d_dist, void* stream);

/* ---- primitive loss callables: loss_function/__init__.py:12-18 ------------------------------------------- */
#define VKAS_LOSS_FOCAL 0     /* focal_with_logits.py:18-47: p0 = alpha (< 0: no alpha weighting), p1 = gamma */
#define VKAS_LOSS_DICE 1      /* dice.py:17-35: pred are probabilities; 1 - 2 sum(pg) / (sum p + sum g + eps) */
#define VKAS_LOSS_L1 2        /* l1.py:19-47, smooth = False */
#define VKAS_LOSS_SMOOTH_L1 3 /* l1.py:19-47, smooth = True: p0 = smooth_beta */
#define VKAS_LOSS_L2 4        /* l2.py:18-34 */
/* pred, gt, mask (nullable): n fp32 elements each.  mask == NULL: mean over n; else sum(e * mask) / (sum(mask) + eps)
 * (dice: pred and gt are multiplied by the mask).  sums: 4 doubles kept for backward; loss: 1 float. */
int vkas_elementwise_loss_fwd(int kind, const float* pred, const float* gt, const float* mask, long n, float p0, float p1,
                              float eps, double* sums, float* loss, void* stream);
int vkas_elementwise_loss_bwd(int kind, const float* pred, const float* gt, const float* mask, long n, float p0, float p1,
                              float eps, const double* sums, const float* dloss, float* dpred, void* stream);
/* F.cross_entropy(logits (rows, classes), target), mean over rows: cross_entropy_with_logits.py:16-19.
 * hard = 0: target is (rows, classes) fp32 class probabilities; hard = 1: (rows,) int64 class indices in range. */
int vkas_cross_entropy_fwd(const float* logits, const void* target, int hard, long rows, int classes, double* sums,
                           float* loss, void* stream);
int vkas_cross_entropy_bwd(const float* logits, const void* target, int hard, long rows, int classes, const float* dloss,
                           float* dlogits, void* stream);

/* ---- label-point backward of the regression heads: loss_function/adaptive_scaling.py:167-179,235-262 -------------------- */
/* The precise loss reads the offset / angle / distance maps only at the (B,P) label points, so d(conv output) of those heads
 * has B*P non-zero rows.  These calls compact them (csrc/points.hip) for vkas_head_tail_bwd and the GEMMs.
 * prepare: py, px (B,P) int64 label points (clamped into the map like the loss does).  map (B*H*W int32 scratch) receives the
 * owner point of every pixel (0x7f7f7f7f = none); pix (Mp >= B*P int32) = pixel index of point i when it owns its pixel,
 * -1 - pixel for a duplicate of an earlier point, INT_MIN for the padding rows i >= B*P. */
int vkas_points_prepare(const long* py, const long* px, int B, int P, int H, int W, int* map, int* pix, long Mp, void* stream);
/* zs (Mp, Ns) = columns [c0, c0+Ns) of the z rows at the points; stats_s (n_heads, Mp, 2) / dproj_s (n_heads, Mp, 8) = the
 * LayerNorm statistics (n_heads, M, 2) and the heads' d(proj) rows (M, 8) there.  Duplicates and padding rows get zero
 * d(proj) (a pixel's gradient is taken once); padding rows are zero throughout.  16-bit activations. */
int vkas_points_gather_rows(const void* z, long ldz, int c0, int Ns, const float* stats, const float* const* dproj,
                            int n_heads, long M, const int* pix, long Mp, void* zs, float* stats_s, float* dproj_s,
                            int dtype, void* stream);
/* xs (Mp, 3, 3, Cp) = the zero-padded 3x3 neighbourhoods x[p + (ky-1, kx-1)] of the owner points (zeros elsewhere): the
 * activation operand of a 1x1 weight-gradient GEMM whose result has the 3x3 convolution's packed (N, ky, kx, Cp) layout. */
int vkas_points_gather_patches(const void* x, long ldx, int Cp, int B, int H, int W, const int* pix, long Mp, void* xs,
                               int dtype, void* stream);
/* D (Mp, 3, 3, Cp) fp32: D[i][ky][kx] is the input-gradient contribution of point i to pixel p_i + (ky-1, kx-1).  Adds them onto
 * dx (B,H,W,Cp; ld lddx): every touched pixel is summed in fp32 by one workgroup and written once. */
int vkas_points_scatter3x3(const float* D, const int* pix, const int* map, long Mp, int B, int H, int W, int Cp, void* dx,
                           long lddx, int dtype, void* stream);

/* (Mp, 8) fp32 rows <-> the (M, 8) projected-channel map of a head at the label points (opt-in label-point forward of
 * the regression heads, ops.HeadsAtPoints): scatter writes the owner rows (pix[i] >= 0) to their pixels (duplicates
 * and padding rows are skipped; dst is expected to be zero elsewhere); gather reads the owners' pixels and zero-fills the other rows. */
int vkas_points_scatter_vec8(const float* src, const int* pix, long Mp, float* dst, void* stream);
int vkas_points_gather_vec8(const float* src, const int* pix, long Mp, float* dst, void* stream);

/* ---- all packed operands of a step in one launch ------------------------------------------------------------------- */
/* One entry per packed image: kind 0 = vkas_pack_conv_weight(_slice) (mode, n_off, Nt, dtype as there; a plain weight has
 * n_off = 0, Nt = Np), kind 1 = vkas_pack_dw_weight (mode = flip; N, KH, KW, Np, n_off, Nt, dtype unused).  total =
 * elements of the image. */
typedef struct {
  const float* w;
  void* out;
  long total;
  int kind, N, C, KH, KW, Np, Cp, mode, n_off, Nt, dtype;
} vkas_pack_desc;
/* descs (count entries) and block_start (count + 1 ints: first workgroup of every entry - an entry takes
 * vkas_pack_many_blocks(&desc) workgroups -; block_start[count] = total_blocks) live in DEVICE memory and stay valid until the launch has run: the table of a model is
 * built once and re-used every step (the parameters and their images keep their addresses). */
int vkas_pack_many(const vkas_pack_desc* descs, const int* block_start, int count, int total_blocks, void* stream);
/* workgroups entry d takes: 32 x 64 tiles of the transposed image for kind 0 / mode 1 (dgrad layout), else 2048 elements each */
int vkas_pack_many_blocks(const vkas_pack_desc* d);

/* ---- inference post-processing on the device: inferencing/adaptive_scaling.py:129-188,318-396 ------------------------ */
/* mask_logit, height (B,1,H,W) fp32 -> out_mask (B,H,W) uint8 = sigmoid >= mask_thr, out_height (B,H,W) fp32 = height where
 * >= height_min else 0; both 0 on rows >= valid_h[b] / columns >= valid_w[b] (the divisible-by-32 padding, in feature
 * pixels; NULL = the whole map is valid). */
int vkas_rough_postprocess(const float* mask_logit, const float* height, int B, int H, int W, const int* valid_h,
                           const int* valid_w, float mask_thr, float height_min, unsigned char* out_mask,
                           float* out_height, void* stream);
/* prob_logit (B,1,H,W), offset (B,2,H,W), angle (B,4,H,W), dist (B,4,H,W) fp32 -> out_prob (B,H,W) = sigmoid (0 in the
 * padding), out_offset (B,H,W,2), out_angle (B,H,W,4) = softmax over the 4 logits, out_dist (B,H,W,4). */
int vkas_precise_postprocess(const float* prob_logit, const float* offset, const float* angle, const float* dist, int B,
                             int H, int W, const int* valid_h, const int* valid_w, float* out_prob, float* out_offset,
                             float* out_angle, float* out_dist, void* stream);

/* ---- optimizer on the flat parameter / gradient buffers: train.py:468-478 ------------------------------ */
/* sumsq (1 double, zeroed by the call) = sum g^2 */
int vkas_l2norm_sq(const float* g, long n, double* sumsq, void* stream);
/* clip coefficient = min(1, max_norm / (sqrt(sumsq) + 1e-6)) [torch clip_grad_norm_]; AdamW (decoupled wd):
 * p -= lr*wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v/(1-b2^t)) + eps) */
int vkas_adamw_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float max_norm,
                    float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VKAS_H */
