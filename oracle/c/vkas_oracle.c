/*
 * ORACLE (test infrastructure only; never linked into or called by the product path).
 *
 * Plain-C, double-precision restatement of the operator semantics on the adaptive-scaling hot path, written from
 * the formulas (no BLAS, no ATen) so that the torch restatement in oracle/torch_oracle.py and the HIP kernels can be
 * checked against something that shares no code with either.  Layout: NCHW, contiguous, as the reference uses.
 * Citations are relative to /root/reference/vkit_open_model/.
 */
#include <math.h>
#include <stddef.h>

#define IDX4(b, c, y, x, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (y)) * (W) + (x))

/* nn.Conv2d(groups=1 or groups=C_in with one filter per channel), zero padding: model/helper.py:25-73.
 * w: (N, C/groups, K, K); depthwise != 0 means groups == C == N. */
void vko_conv2d(const double* x, const double* w, const double* bias, double* y, int B, int C, int H, int W, int N, int K,
                int stride, int pad, int depthwise) {
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n)
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = bias ? bias[n] : 0.0;
          const int c_lo = depthwise ? n : 0, c_hi = depthwise ? n + 1 : C;
          for (int c = c_lo; c < c_hi; ++c)
            for (int ky = 0; ky < K; ++ky) {
              const int iy = oy * stride - pad + ky;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                const size_t wi = depthwise ? ((size_t)n * K + ky) * K + kx : (((size_t)n * C + c) * K + ky) * K + kx;
                acc += x[IDX4(b, c, iy, ix, C, H, W)] * w[wi];
              }
            }
          y[IDX4(b, n, oy, ox, N, Ho, Wo)] = acc;
        }
}

/* nn.Linear on the channel axis (helper.py:18-22): w (N, C) */
void vko_linear(const double* x, const double* w, const double* bias, double* y, int B, int C, int H, int W, int N) {
  vko_conv2d(x, w, bias, y, B, C, H, W, N, 1, 1, 0, 0);
}

/* nn.LayerNorm(C, eps=1e-6) over the channel axis of every pixel, biased variance (helper.py:96-97) */
void vko_layernorm(const double* x, const double* g, const double* bt, double* y, int B, int C, int H, int W) {
  for (int b = 0; b < B; ++b)
    for (int yy = 0; yy < H; ++yy)
      for (int xx = 0; xx < W; ++xx) {
        double mean = 0.0, var = 0.0;
        for (int c = 0; c < C; ++c) mean += x[IDX4(b, c, yy, xx, C, H, W)];
        mean /= C;
        for (int c = 0; c < C; ++c) {
          const double d = x[IDX4(b, c, yy, xx, C, H, W)] - mean;
          var += d * d;
        }
        var /= C;
        const double rstd = 1.0 / sqrt(var + 1e-6);
        for (int c = 0; c < C; ++c)
          y[IDX4(b, c, yy, xx, C, H, W)] = (x[IDX4(b, c, yy, xx, C, H, W)] - mean) * rstd * g[c] + bt[c];
      }
}

/* exact GELU (helper.py:100-101) and Softplus(beta 1, threshold 20) (model/adaptive_scaling.py:101,140) */
void vko_gelu(const double* x, double* y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = 0.5 * x[i] * (1.0 + erf(x[i] * 0.70710678118654752440));
}
void vko_softplus(const double* x, double* y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = x[i] > 20.0 ? x[i] : log1p(exp(x[i]));
}

/* F.interpolate(size=..., mode='bilinear', align_corners=False): upernext.py:79,178-195,237-244 */
void vko_bilinear(const double* x, double* y, int B, int C, int Hi, int Wi, int Ho, int Wo) {
  const double sy = (double)Hi / Ho, sx = (double)Wi / Wo;
  for (int oy = 0; oy < Ho; ++oy) {
    double fy = (oy + 0.5) * sy - 0.5;
    if (fy < 0) fy = 0;
    int y0 = (int)floor(fy);
    if (y0 > Hi - 1) y0 = Hi - 1;
    const int y1 = y0 + 1 < Hi ? y0 + 1 : Hi - 1;
    const double wy = fy - y0;
    for (int ox = 0; ox < Wo; ++ox) {
      double fx = (ox + 0.5) * sx - 0.5;
      if (fx < 0) fx = 0;
      int x0 = (int)floor(fx);
      if (x0 > Wi - 1) x0 = Wi - 1;
      const int x1 = x0 + 1 < Wi ? x0 + 1 : Wi - 1;
      const double wx = fx - x0;
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
          const double top = x[IDX4(b, c, y0, x0, C, Hi, Wi)] * (1 - wx) + x[IDX4(b, c, y0, x1, C, Hi, Wi)] * wx;
          const double bot = x[IDX4(b, c, y1, x0, C, Hi, Wi)] * (1 - wx) + x[IDX4(b, c, y1, x1, C, Hi, Wi)] * wx;
          y[IDX4(b, c, oy, ox, C, Ho, Wo)] = top * (1 - wy) + bot * wy;
        }
    }
  }
}

/* F.interpolate(mode='nearest'): src = min(floor(dst * in / out), in - 1): fpn.py:125-142,197-204 */
void vko_nearest(const double* x, double* y, int B, int C, int Hi, int Wi, int Ho, int Wo) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          int iy = (int)(((long)oy * Hi) / Ho), ix = (int)(((long)ox * Wi) / Wo);
          if (iy > Hi - 1) iy = Hi - 1;
          if (ix > Wi - 1) ix = Wi - 1;
          y[IDX4(b, c, oy, ox, C, Ho, Wo)] = x[IDX4(b, c, iy, ix, C, Hi, Wi)];
        }
}

/* nn.AdaptiveAvgPool2d(s): bin i = [floor(i*H/s), ceil((i+1)*H/s)): upernext.py:62 */
void vko_adaptive_avgpool(const double* x, double* y, int B, int C, int H, int W, int s) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) {
          const int y0 = (i * H) / s, y1 = ((i + 1) * H + s - 1) / s, x0 = (j * W) / s, x1 = ((j + 1) * W + s - 1) / s;
          double acc = 0.0;
          for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) acc += x[IDX4(b, c, yy, xx, C, H, W)];
          y[IDX4(b, c, i, j, C, s, s)] = acc / ((y1 - y0) * (x1 - x0));
        }
}

/* ConvNextBlockLayer.forward (convnext.py:29-59) for one layer: x + mask_b * scale_c * MLP(LN(dw7x7(x))).
 * tmp must hold B*C*H*W + 2*B*4C*H*W doubles. */
void vko_convnext_layer(const double* x, const double* dw_w, const double* dw_b, const double* ln_g, const double* ln_b,
                        const double* w1, const double* b1, const double* w2, const double* b2, const double* scale,
                        const double* mask, double* out, double* tmp, int B, int C, int H, int W) {
  const size_t n = (size_t)B * C * H * W, n4 = 4 * n;
  double* t0 = tmp;
  double* t1 = tmp + n;
  double* t2 = t1 + n4;
  vko_conv2d(x, dw_w, dw_b, t0, B, C, H, W, C, 7, 1, 3, 1);
  vko_layernorm(t0, ln_g, ln_b, t0, B, C, H, W);
  vko_linear(t0, w1, b1, t1, B, C, H, W, 4 * C);
  vko_gelu(t1, t2, n4);
  vko_linear(t2, w2, b2, t0, B, 4 * C, H, W, C);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < H * W; ++p) {
        const size_t i = ((size_t)b * C + c) * H * W + p;
        out[i] = x[i] + (mask ? mask[b] : 1.0) * scale[c] * t0[i];
      }
}

/* rough loss terms (loss_function/adaptive_scaling.py:53-131), default factors; crop already applied by the caller:
 * m, h, gt_mask, gt_score are n cropped elements each.  Returns 5*focal + dice + masked log smooth-L1. */
double vko_rough_loss(const double* m, const double* h, const double* gm, const double* gs, size_t n) {
  double focal = 0, spg = 0, sp = 0, sg = 0, l1 = 0, cnt = 0;
  for (size_t i = 0; i < n; ++i) {
    const double x = m[i], t = gm[i];
    const double p = 1.0 / (1.0 + exp(-x));
    const double ce = (x > 0 ? x : 0) - x * t + log1p(exp(-fabs(x)));
    const double pt = p * t + (1 - p) * (1 - t);
    focal += (0.25 * t + 0.75 * (1 - t)) * ce * (1 - pt) * (1 - pt);
    spg += p * t;
    sp += p;
    sg += t;
    if (h[i] > 1.1 && gs[i] > 1.1 && t != 0.0) {
      const double d = fabs(log(h[i]) - log(gs[i]));
      l1 += d < 1.0 ? 0.5 * d * d : d - 0.5;
      cnt += 1;
    }
  }
  return 5.0 * focal / n + (1.0 - 2.0 * spg / (sp + sg + 1e-6)) + l1 / (cnt + 1e-6);
}
