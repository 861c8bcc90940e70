/*
 * bt709_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see bt709_oracle.h).
 *
 * Every arithmetic step states which reference line it follows (paths relative
 * to /root/reference/Renderer).  Three C rules of the reference are kept on
 * purpose because they decide the last bit:
 *   - pow() is the double libm function; its float arguments are promoted and
 *     the result is narrowed to float only on assignment (sRGB.h:54,70;
 *     BT709.h:77,99,133,147);
 *   - "(1.0f + a) * pow(...) - a" is evaluated entirely in double and narrowed
 *     once (sRGB.h:70, BT709.h:99);
 *   - "v * 255.0f" is a float multiply whose result is then promoted for
 *     round() (BT709.h:881-883).
 * Compile with -ffp-contract=off so none of the float expressions are fused.
 */
#include "bt709_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ consts */

/* BT709.h:40-59 */
static const float kKr = 0.2126f;
static const float kKg = 0.7152f;
static const float kKb = 0.0722f;
static const float kCrSpan = 1.5748f; /* BT709_Er_minus_Ey_Range */
static const float kCbSpan = 1.8556f; /* BT709_Eb_minus_Ey_Range */
enum { kYLo = 16, kYHi = 235, kCLo = 16, kCHi = 240 };

/* sRGB.h:18-27 */
static float clamp01(float v) {
  if (v < 0.0f) return 0.0f;
  if (v > 1.0f) return 1.0f;
  return v;
}

/* sRGB.h:32-36 */
static float byte_norm(int b) { return b * (1.0f / 255.0f); }

/* ------------------------------------------------------- transfer functions */

float bt709o_srgb_to_linear(float v) {
  /* sRGB.h:43-57 */
  if (v <= 0.04045f) return v * (1.0f / 12.92f);
  const float a = 0.055f;
  const float g = 2.4f;
  float base = (v + a) * (1.0f / (1.0f + a));
  return (float)pow((double)base, (double)g);
}

float bt709o_linear_to_srgb(float v) {
  /* sRGB.h:62-74 */
  if (v <= 0.0031308f) return v * 12.92f;
  const float a = 0.055f;
  const float g = 1.0f / 2.4f;
  double r = (double)(1.0f + a) * pow((double)v, (double)g) - (double)a;
  return (float)r;
}

float bt709o_itu709_to_linear(float v) {
  /* BT709.h:68-81 */
  if (v < 0.081f) return v * (1.0f / 4.5f);
  const float a = 0.099f;
  const float g = 1.0f / 0.45f;
  float base = (v + a) * (1.0f / (1.0f + a));
  return (float)pow((double)base, (double)g);
}

float bt709o_linear_to_itu709(float v) {
  /* BT709.h:90-103 */
  if (v < 0.018f) return v * 4.5f;
  const float a = 0.099f;
  const float g = 0.45f;
  double r = (double)(1.0f + a) * pow((double)v, (double)g) - (double)a;
  return (float)r;
}

#define APPLE_G 1.960938f /* BT709.h:122 */

float bt709o_apple196_to_linear(float v) {
  /* BT709.h:125-137 */
  if (v < 0.05583828f) return v * (1.0f / 16.0f);
  return (float)pow((double)v, (double)APPLE_G);
}

float bt709o_linear_to_apple196(float v) {
  /* BT709.h:139-151 */
  if (v < 0.00349f) return v * 16.0f;
  const float g = 1.0f / APPLE_G;
  return (float)pow((double)v, (double)g);
}

int bt709o_quantize(float v) {
  float scaled = v * 255.0f; /* float multiply, BT709.h:881 */
  return (int)round((double)scaled);
}

int bt709o_transfer_to_byte(int gamma, float v) {
  switch (gamma) {
    case BT709O_GAMMA_APPLE: /* BT709.h:856-878 */
      v = bt709o_linear_to_srgb(bt709o_apple196_to_linear(v));
      break;
    case BT709O_GAMMA_SRGB: /* BT709.h:977-983: no curve at all */
      break;
    case BT709O_GAMMA_LINEAR: /* BT709.h:1156-1157 */
      v = bt709o_linear_to_srgb(v);
      break;
    case BT709O_GAMMA_ITU709: /* BT709.h:540-545 + 704-716 */
      v = bt709o_linear_to_srgb(bt709o_itu709_to_linear(v));
      break;
    case BT709O_TABLE_ENCODE_APPLE: /* BT709_from_linear(v, Apple), BT709.h:1158-1159 */
      v = bt709o_linear_to_apple196(v);
      break;
    default:
      return -1;
  }
  return bt709o_quantize(v);
}

/* --------------------------------------------------------------- per pixel */

void bt709o_ycbcr_to_rgbn(int Y, int Cb, int Cr, float rgbn[3]) {
  /* BT709.h:494-500: centre, then scale by 1/255 in float */
  float yn = (Y - 16) * (1.0f / 255.0f);
  float cbn = (Cb - 128) * (1.0f / 255.0f);
  float crn = (Cr - 128) * (1.0f / 255.0f);

  /* BT709.h:386-397 with unscale=1; every product is a float op */
  const float ys = 255.0f / (kYHi - kYLo);
  const float cs = 255.0f / (kCHi - kCLo);
  const float kr_kg = kKr / kKg; /* BT709.h:52 */
  const float kb_kg = kKb / kKg; /* BT709.h:53 */
  const float m[9] = {
      ys, 0.000f, (cs * kCrSpan),
      ys, (-1.0f * cs * kCbSpan * kb_kg), (-1.0f * cs * kCrSpan * kr_kg),
      ys, (cs * kCbSpan), 0.000f,
  };

  /* BT709.h:424-426: ((Y*a)+(Cb*b))+(Cr*c), zero terms included */
  float r = (yn * m[0]) + (cbn * m[1]) + (crn * m[2]);
  float g = (yn * m[3]) + (cbn * m[4]) + (crn * m[5]);
  float b = (yn * m[6]) + (cbn * m[7]) + (crn * m[8]);

  /* BT709.h:444-446 */
  rgbn[0] = clamp01(r);
  rgbn[1] = clamp01(g);
  rgbn[2] = clamp01(b);
}

void bt709o_decode_pixel(int gamma, int Y, int Cb, int Cr, int rgb[3]) {
  float n[3];
  bt709o_ycbcr_to_rgbn(Y, Cb, Cr, n);
  for (int c = 0; c < 3; c++) rgb[c] = bt709o_transfer_to_byte(gamma, n[c]);
}

int bt709o_decode_alpha(int A) {
  float n[3];
  bt709o_ycbcr_to_rgbn(A, 128, 128, n);
  return bt709o_quantize(n[0]);
}

/* BT709.h:199-268: non-linear normalised RGB -> integer YCbCr */
static void rgbn_to_ycbcr(float rn, float gn, float bn, int out[3]) {
  float ey = (kKr * rn) + (kKg * gn) + (kKb * bn);
  float eb = (bn - ey) / kCbSpan;
  float er = (rn - ey) / kCrSpan;
  float ay = (ey * (kYHi - kYLo)) + 16;
  float ab = (eb * (kCHi - kCLo)) + 128;
  float ar = (er * (kCHi - kCLo)) + 128;
  out[0] = (int)round((double)ay);
  out[1] = (int)round((double)ab);
  out[2] = (int)round((double)ar);
}

void bt709o_encode_pixel(int gamma, int R, int G, int B, int ycbcr[3]) {
  float n[3] = {byte_norm(R), byte_norm(G), byte_norm(B)};
  if (gamma != BT709O_GAMMA_SRGB) {
    /* BT709.h:644-646 / 786-788: sRGB bytes -> linear light */
    for (int c = 0; c < 3; c++) n[c] = bt709o_srgb_to_linear(n[c]);
    if (gamma == BT709O_GAMMA_APPLE) { /* BT709.h:801-803 */
      for (int c = 0; c < 3; c++) n[c] = bt709o_linear_to_apple196(n[c]);
    } else if (gamma == BT709O_GAMMA_ITU709) { /* BT709.h:306-308 */
      for (int c = 0; c < 3; c++) n[c] = bt709o_linear_to_itu709(n[c]);
    }
  }
  rgbn_to_ycbcr(n[0], n[1], n[2], ycbcr);
}

void bt709o_encode_linear_pixel(int apply_curve, int R, int G, int B, int ycbcr[3]) {
  float n[3] = {byte_norm(R), byte_norm(G), byte_norm(B)}; /* BT709.h:336-338 */
  if (apply_curve) /* BT709.h:300-315 */
    for (int c = 0; c < 3; c++) n[c] = bt709o_linear_to_itu709(n[c]);
  rgbn_to_ycbcr(n[0], n[1], n[2], ycbcr);
}

void bt709o_decode_to_linear_pixel(int apply_curve, int Y, int Cb, int Cr, int rgb[3]) {
  float n[3];
  bt709o_ycbcr_to_rgbn(Y, Cb, Cr, n);
  for (int c = 0; c < 3; c++) {
    if (apply_curve) n[c] = bt709o_itu709_to_linear(n[c]); /* BT709.h:536-546 */
    rgb[c] = bt709o_quantize(n[c]);                        /* BT709.h:577-579 */
  }
}

/* ------------------------------------------------------------------ frames */

/* The per-channel byte map depends only on the saturated pre-gamma float, and
 * that float only on the three input bytes, so a frame decode is a pure
 * per-pixel function; no state is shared between pixels. */

int bt709o_decode_nv12_rows(int gamma,
                            const uint8_t *y, size_t y_stride,
                            const uint8_t *uv, size_t uv_stride,
                            const uint8_t *alpha, size_t alpha_stride,
                            int width, int row0, int row1,
                            uint8_t *bgra, size_t bgra_stride,
                            int alpha_fill) {
  if ((width & 1) || (row0 & 1)) return -1;
  for (int row = row0; row < row1; row++) {
    const uint8_t *yrow = y + (size_t)row * y_stride;
    const uint8_t *crow = uv + (size_t)(row / 2) * uv_stride; /* .m:268-269 */
    const uint8_t *arow = alpha ? alpha + (size_t)row * alpha_stride : NULL;
    uint8_t *orow = bgra + (size_t)row * bgra_stride;
    for (int col = 0; col < width; col++) {
      int rgb[3];
      int cb = crow[2 * (col / 2)];     /* low byte  = Cb (.m:1083) */
      int cr = crow[2 * (col / 2) + 1]; /* high byte = Cr           */
      bt709o_decode_pixel(gamma, yrow[col], cb, cr, rgb);
      int a = arow ? bt709o_decode_alpha(arow[col]) : alpha_fill;
      orow[4 * col + 0] = (uint8_t)rgb[2];
      orow[4 * col + 1] = (uint8_t)rgb[1];
      orow[4 * col + 2] = (uint8_t)rgb[0];
      orow[4 * col + 3] = (uint8_t)a;
    }
  }
  return 0;
}

int bt709o_decode_nv12(int gamma,
                       const uint8_t *y, size_t y_stride,
                       const uint8_t *uv, size_t uv_stride,
                       const uint8_t *alpha, size_t alpha_stride,
                       int width, int height,
                       uint8_t *bgra, size_t bgra_stride,
                       int alpha_fill) {
  if ((width & 1) || (height & 1)) return -1; /* .m:69-74 */
  return bt709o_decode_nv12_rows(gamma, y, y_stride, uv, uv_stride, alpha,
                                 alpha_stride, width, 0, height, bgra,
                                 bgra_stride, alpha_fill);
}

int bt709o_decode_nv12_half(int gamma,
                            const uint8_t *y, size_t y_stride,
                            const uint8_t *uv, size_t uv_stride,
                            const uint8_t *alpha, size_t alpha_stride,
                            int width, int height,
                            uint8_t *bgra, size_t bgra_stride,
                            int alpha_fill) {
  if ((width & 3) || (height & 3)) return -1;
  float lin[256]; /* sampler-side decode of an sRGB8 texel */
  for (int b = 0; b < 256; b++) lin[b] = bt709o_srgb_to_linear(byte_norm(b));
  for (int orow = 0; orow < height / 2; orow++) {
    uint8_t *out = bgra + (size_t)orow * bgra_stride;
    /* a 2x2 luma block shares exactly one CbCr sample */
    const uint8_t *crow = uv + (size_t)orow * uv_stride;
    const uint8_t *y0 = y + (size_t)(2 * orow) * y_stride;
    const uint8_t *y1 = y0 + y_stride;
    for (int ocol = 0; ocol < width / 2; ocol++) {
      int cb = crow[2 * ocol], cr = crow[2 * ocol + 1];
      int p[4][3];
      bt709o_decode_pixel(gamma, y0[2 * ocol], cb, cr, p[0]);
      bt709o_decode_pixel(gamma, y0[2 * ocol + 1], cb, cr, p[1]);
      bt709o_decode_pixel(gamma, y1[2 * ocol], cb, cr, p[2]);
      bt709o_decode_pixel(gamma, y1[2 * ocol + 1], cb, cr, p[3]);
      int q[3];
      for (int c = 0; c < 3; c++) {
        float s = (((lin[p[0][c]] + lin[p[1][c]]) + lin[p[2][c]]) + lin[p[3][c]]) * 0.25f;
        q[c] = bt709o_quantize(bt709o_linear_to_srgb(s));
      }
      int a = alpha_fill;
      if (alpha) {
        /* the alpha channel of the 8-bit intermediate is a plain unorm (no sRGB curve): the sampler
         * reads byteNorm(a), the filtered value is written back as round(255 v)
         * (AAPLShaders.metal:411-438 writes it; MetalScaleRenderContext.m:55-105 filters it) */
        const uint8_t *a0 = alpha + (size_t)(2 * orow) * alpha_stride, *a1 = a0 + alpha_stride;
        float s = (((byte_norm(bt709o_decode_alpha(a0[2 * ocol])) + byte_norm(bt709o_decode_alpha(a0[2 * ocol + 1]))) +
                    byte_norm(bt709o_decode_alpha(a1[2 * ocol]))) + byte_norm(bt709o_decode_alpha(a1[2 * ocol + 1]))) * 0.25f;
        a = bt709o_quantize(s);
      }
      out[4 * ocol + 0] = (uint8_t)q[2];
      out[4 * ocol + 1] = (uint8_t)q[1];
      out[4 * ocol + 2] = (uint8_t)q[0];
      out[4 * ocol + 3] = (uint8_t)a;
    }
  }
  return 0;
}

int bt709o_decode_nv12_scaled(int gamma,
                              const uint8_t *y, size_t y_stride,
                              const uint8_t *uv, size_t uv_stride,
                              const uint8_t *alpha, size_t alpha_stride,
                              int width, int height,
                              uint8_t *bgra, size_t bgra_stride,
                              int out_width, int out_height, int alpha_fill) {
  if ((width & 1) || (height & 1) || width <= 0 || height <= 0 || out_width <= 0 || out_height <= 0) return -1;
  float lin[256];
  for (int b = 0; b < 256; b++) lin[b] = bt709o_srgb_to_linear(byte_norm(b));
  const float scale_x = (float)width / (float)out_width;
  const float scale_y = (float)height / (float)out_height;
  for (int oy = 0; oy < out_height; oy++) {
    const float sy = ((float)oy + 0.5f) * scale_y - 0.5f; /* texel-centre sampling */
    const float y0f = floorf(sy), fy = sy - y0f, gy = 1.0f - fy;
    int ys[2] = {(int)y0f, (int)y0f + 1};
    for (int i = 0; i < 2; i++) ys[i] = ys[i] < 0 ? 0 : (ys[i] > height - 1 ? height - 1 : ys[i]); /* clamp to edge */
    uint8_t *out = bgra + (size_t)oy * bgra_stride;
    for (int ox = 0; ox < out_width; ox++) {
      const float sx = ((float)ox + 0.5f) * scale_x - 0.5f;
      const float x0f = floorf(sx), fx = sx - x0f, gx = 1.0f - fx;
      int xs[2] = {(int)x0f, (int)x0f + 1};
      for (int i = 0; i < 2; i++) xs[i] = xs[i] < 0 ? 0 : (xs[i] > width - 1 ? width - 1 : xs[i]);
      const float w[4] = {gx * gy, fx * gy, gx * fy, fx * fy};
      float acc[3] = {0.0f, 0.0f, 0.0f}, acc_a = 0.0f;
      for (int t = 0; t < 4; t++) {
        const int xx = xs[t & 1], yy = ys[t >> 1];
        if (alpha) { /* plain unorm channel, same weights and order */
          const float term = w[t] * byte_norm(bt709o_decode_alpha(alpha[(size_t)yy * alpha_stride + xx]));
          acc_a = t ? acc_a + term : term;
        }
        const uint8_t *c = uv + (size_t)(yy / 2) * uv_stride + 2 * (xx / 2);
        int p[3];
        bt709o_decode_pixel(gamma, y[(size_t)yy * y_stride + xx], c[0], c[1], p);
        for (int k = 0; k < 3; k++) {
          const float term = w[t] * lin[p[k]];
          acc[k] = t ? acc[k] + term : term; /* (((w00*a + w01*b) + w10*c) + w11*d) */
        }
      }
      int q[3];
      for (int k = 0; k < 3; k++) q[k] = bt709o_quantize(bt709o_linear_to_srgb(acc[k]));
      out[4 * ox + 0] = (uint8_t)q[2];
      out[4 * ox + 1] = (uint8_t)q[1];
      out[4 * ox + 2] = (uint8_t)q[0];
      out[4 * ox + 3] = (uint8_t)(alpha ? bt709o_quantize(clamp01(acc_a)) : alpha_fill);
    }
  }
  return 0;
}

/* ---- RGBA16Float render targets and the stand-alone pass 2 ------------------------------- */

/* IEEE binary32 -> binary16, round to nearest even, by exact integer arithmetic on the value:
 * the nearest multiple of the target's unit in the last place, ties to the even one. */
uint16_t bt709o_float_to_half(float v) {
  if (v != v) return 0x7e00;
  const uint16_t sign = signbit(v) ? 0x8000u : 0u;
  const double a = fabs((double)v);
  if (a >= 65520.0) return (uint16_t)(sign | 0x7c00u); /* halfway to 2^16 and beyond rounds to infinity */
  int e;
  (void)frexp(a, &e);                 /* a = m * 2^e, m in [0.5, 1) */
  int ulp_exp = (e - 1) - 10;         /* unit of a normal half with that exponent */
  if (ulp_exp < -24) ulp_exp = -24;   /* subnormal halves share the unit 2^-24 */
  const double q = nearbyint(ldexp(a, -ulp_exp)); /* default rounding mode: nearest, ties to even */
  const double r = ldexp(q, ulp_exp);             /* the rounded value, exact */
  if (r == 0.0) return sign;
  int re;
  const double rm = frexp(r, &re);    /* r may have carried into the next binade */
  if (re - 1 < -14) return (uint16_t)(sign | (uint16_t)ldexp(r, 24)); /* subnormal: r / 2^-24 */
  return (uint16_t)(sign | ((uint16_t)(re - 1 + 15) << 10) | ((uint16_t)ldexp(rm, 11) & 0x3ffu));
}

float bt709o_half_to_float(uint16_t h) {
  const int e = (h >> 10) & 0x1f, m = h & 0x3ff;
  double v;
  if (e == 0) v = ldexp((double)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexp((double)(m | 0x400), e - 25);
  return (float)((h & 0x8000) ? -v : v);
}

float bt709o_curve_to_linear(int gamma, float v) {
  switch (gamma) {
    case BT709O_GAMMA_APPLE: return bt709o_apple196_to_linear(v);  /* BT709ToLinearSRGBKernel, AAPLShaders.metal:336-357 */
    case BT709O_GAMMA_SRGB: return bt709o_srgb_to_linear(v);       /* sRGBToLinearSRGBKernel, :361-382 */
    case BT709O_GAMMA_ITU709: return bt709o_itu709_to_linear(v);
    default: return v;                                             /* LinearToLinearSRGBKernel, :387-407 */
  }
}

int bt709o_decode_nv12_rgba16f(int gamma, const uint8_t *y, size_t y_stride, const uint8_t *uv, size_t uv_stride,
                               const uint8_t *alpha, size_t alpha_stride, int width, int height, uint8_t *rgba,
                               size_t rgba_stride) {
  if ((width & 1) || (height & 1)) return -1;
  for (int row = 0; row < height; row++) {
    uint16_t *o = (uint16_t *)(rgba + (size_t)row * rgba_stride);
    for (int col = 0; col < width; col++) {
      const uint8_t *c = uv + (size_t)(row / 2) * uv_stride + 2 * (col / 2);
      float n[3], a = 1.0f;
      bt709o_ycbcr_to_rgbn(y[(size_t)row * y_stride + col], c[0], c[1], n);
      if (alpha) { /* BT709_decodeAlpha: the luma term alone, saturated; linear */
        float an[3];
        bt709o_ycbcr_to_rgbn(alpha[(size_t)row * alpha_stride + col], 128, 128, an);
        a = an[0];
      }
      for (int k = 0; k < 3; k++) o[4 * col + k] = bt709o_float_to_half(bt709o_curve_to_linear(gamma, n[k]));
      o[4 * col + 3] = bt709o_float_to_half(a);
    }
  }
  return 0;
}

int bt709o_render_scaled(int in_format, const uint8_t *in, size_t in_stride, int width, int height, uint8_t *bgra,
                         size_t bgra_stride, int out_width, int out_height) {
  if (width <= 0 || height <= 0 || out_width <= 0 || out_height <= 0 || (in_format != 0 && in_format != 1)) return -1;
  float lin[256];
  for (int b = 0; b < 256; b++) lin[b] = bt709o_srgb_to_linear(byte_norm(b));
  const float scale_x = (float)width / (float)out_width;
  const float scale_y = (float)height / (float)out_height;
  for (int oy = 0; oy < out_height; oy++) {
    const float sy = ((float)oy + 0.5f) * scale_y - 0.5f;
    const float y0f = floorf(sy), fy = sy - y0f, gy = 1.0f - fy;
    int ys[2] = {(int)y0f, (int)y0f + 1};
    for (int i = 0; i < 2; i++) ys[i] = ys[i] < 0 ? 0 : (ys[i] > height - 1 ? height - 1 : ys[i]);
    uint8_t *out = bgra + (size_t)oy * bgra_stride;
    for (int ox = 0; ox < out_width; ox++) {
      const float sx = ((float)ox + 0.5f) * scale_x - 0.5f;
      const float x0f = floorf(sx), fx = sx - x0f, gx = 1.0f - fx;
      int xs[2] = {(int)x0f, (int)x0f + 1};
      for (int i = 0; i < 2; i++) xs[i] = xs[i] < 0 ? 0 : (xs[i] > width - 1 ? width - 1 : xs[i]);
      const float w[4] = {gx * gy, fx * gy, gx * fy, fx * fy};
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f}; /* R, G, B, A */
      for (int t = 0; t < 4; t++) {
        const uint8_t *px = in + (size_t)ys[t >> 1] * in_stride + (size_t)xs[t & 1] * (in_format ? 8 : 4);
        float s[4];
        if (in_format == 0) { /* BGRA8 sRGB texel: rgb decoded by the sampler, alpha a plain unorm */
          s[0] = lin[px[2]], s[1] = lin[px[1]], s[2] = lin[px[0]], s[3] = byte_norm(px[3]);
        } else {              /* RGBA16Float texel: linear light already */
          const uint16_t *h = (const uint16_t *)px;
          for (int k = 0; k < 4; k++) s[k] = bt709o_half_to_float(h[k]);
        }
        for (int k = 0; k < 4; k++) {
          const float term = w[t] * s[k];
          acc[k] = t ? acc[k] + term : term;
        }
      }
      for (int k = 0; k < 3; k++) out[4 * ox + 2 - k] = (uint8_t)bt709o_quantize(bt709o_linear_to_srgb(clamp01(acc[k])));
      out[4 * ox + 3] = (uint8_t)bt709o_quantize(clamp01(acc[3]));
    }
  }
  return 0;
}

int bt709o_unconvert_packed(int gamma, const uint32_t *ycbcr, uint32_t *bgra,
                            int width, int height) {
  if ((width & 1) || (height & 1)) return -1;
  size_t n = (size_t)width * height;
  for (size_t i = 0; i < n; i++) {
    uint32_t p = ycbcr[i];
    int rgb[3];
    bt709o_decode_pixel(gamma, p & 0xFF, (p >> 8) & 0xFF, (p >> 16) & 0xFF, rgb);
    bgra[i] = ((uint32_t)rgb[0] << 16) | ((uint32_t)rgb[1] << 8) | (uint32_t)rgb[2];
  }
  return 0;
}

int bt709o_convert_packed(int gamma, const uint32_t *bgra, uint32_t *ycbcr,
                          int width, int height) {
  if ((width & 1) || (height & 1)) return -1; /* BGRAToBT709Converter.m:41-46 */
  size_t n = (size_t)width * height;
  for (size_t i = 0; i < n; i++) {
    uint32_t p = bgra[i];
    int v[3];
    bt709o_encode_pixel(gamma, (p >> 16) & 0xFF, (p >> 8) & 0xFF, p & 0xFF, v);
    ycbcr[i] = ((uint32_t)v[2] << 16) | ((uint32_t)v[1] << 8) | (uint32_t)v[0];
  }
  return 0;
}

void bt709o_packed_to_nv12(const uint32_t *ycbcr, int width, int height,
                           uint8_t *y, size_t y_stride,
                           uint8_t *uv, size_t uv_stride) {
  for (int row = 0; row < height; row++) {
    uint8_t *yrow = y + (size_t)row * y_stride;
    uint8_t *crow = uv + (size_t)(row / 2) * uv_stride;
    for (int col = 0; col < width; col++) {
      uint32_t p = ycbcr[(size_t)row * width + col];
      yrow[col] = (uint8_t)(p & 0xFF);
      if ((col & 1) == 0) { /* .m:1078-1084: every row overwrites row/2 */
        crow[col] = (uint8_t)((p >> 8) & 0xFF);
        crow[col + 1] = (uint8_t)((p >> 16) & 0xFF);
      }
    }
  }
}

void bt709o_nv12_to_packed(const uint8_t *y, size_t y_stride,
                           const uint8_t *uv, size_t uv_stride,
                           int width, int height, uint32_t *ycbcr) {
  for (int row = 0; row < height; row++) {
    const uint8_t *yrow = y + (size_t)row * y_stride;
    const uint8_t *crow = uv + (size_t)(row / 2) * uv_stride;
    for (int col = 0; col < width; col++) {
      uint32_t cb = crow[2 * (col / 2)], cr = crow[2 * (col / 2) + 1];
      ycbcr[(size_t)row * width + col] = (cr << 16) | (cb << 8) | yrow[col];
    }
  }
}

/* ----------------------------------------------------- encode-side helpers */

/* BT709.h:1100-1146 (BT709_tolinearNorm) for one channel */
static float to_linear(int gamma, int byte) {
  float n = byte_norm(byte);
  if (gamma == BT709O_GAMMA_SRGB) return bt709o_srgb_to_linear(n);
  if (gamma == BT709O_GAMMA_APPLE) return bt709o_apple196_to_linear(n);
  return n;
}

/* BT709.h:1150-1167 (BT709_from_linear) */
static int from_linear(int gamma, float v) {
  float nl = v;
  if (gamma == BT709O_GAMMA_SRGB) nl = bt709o_linear_to_srgb(v);
  else if (gamma == BT709O_GAMMA_APPLE) nl = bt709o_linear_to_apple196(v);
  return bt709o_quantize(nl);
}

void bt709o_subsample_block(const int rgb[12], int in_gamma, int out_gamma,
                            int y4[4], int *cb, int *cr) {
  float lin[4][3];
  for (int i = 0; i < 4; i++)
    for (int c = 0; c < 3; c++) lin[i][c] = to_linear(in_gamma, rgb[3 * i + c]);

  /* BT709.h:1404-1406 via 1171-1190: ((a+b)+c)+d then /4.0f */
  int avg[3];
  for (int c = 0; c < 3; c++) {
    float sum = (lin[0][c] + lin[1][c] + lin[2][c] + lin[3][c]);
    avg[c] = from_linear(out_gamma, sum / 4.0f); /* BT709.h:1412-1414 */
  }
  int v[3];
  bt709o_encode_pixel(BT709O_GAMMA_SRGB, avg[0], avg[1], avg[2], v); /* :1418 */
  *cb = v[1];
  *cr = v[2];

  for (int i = 0; i < 4; i++) { /* BT709.h:1423-1487 */
    int e[3];
    for (int c = 0; c < 3; c++) e[c] = from_linear(out_gamma, lin[i][c]);
    bt709o_encode_pixel(BT709O_GAMMA_SRGB, e[0], e[1], e[2], v);
    y4[i] = v[0];
  }
}

int bt709o_encode_nv12(const uint32_t *bgra, int width, int height,
                       int in_gamma, int out_gamma,
                       uint8_t *y, size_t y_stride,
                       uint8_t *uv, size_t uv_stride) {
  if ((width & 1) || (height & 1)) return -1;
  for (int row = 0; row < height; row += 2) {
    for (int col = 0; col < width; col += 2) {
      /* CVPixelBufferUtils.h:301-322: p1,p2 top row; p3,p4 bottom row */
      uint32_t p[4] = {bgra[(size_t)row * width + col], bgra[(size_t)row * width + col + 1],
                       bgra[(size_t)(row + 1) * width + col],
                       bgra[(size_t)(row + 1) * width + col + 1]};
      int rgb[12];
      for (int i = 0; i < 4; i++) {
        rgb[3 * i + 0] = (p[i] >> 16) & 0xFF;
        rgb[3 * i + 1] = (p[i] >> 8) & 0xFF;
        rgb[3 * i + 2] = p[i] & 0xFF;
      }
      int y4[4], cb, cr;
      bt709o_subsample_block(rgb, in_gamma, out_gamma, y4, &cb, &cr);
      y[(size_t)row * y_stride + col] = (uint8_t)y4[0];
      y[(size_t)row * y_stride + col + 1] = (uint8_t)y4[1];
      y[(size_t)(row + 1) * y_stride + col] = (uint8_t)y4[2];
      y[(size_t)(row + 1) * y_stride + col + 1] = (uint8_t)y4[3];
      uv[(size_t)(row / 2) * uv_stride + col] = (uint8_t)cb;
      uv[(size_t)(row / 2) * uv_stride + col + 1] = (uint8_t)cr;
    }
  }
  return 0;
}

/* --------------------------------------------------------------- threading */

typedef void (*range_fn)(void *ctx, uint64_t lo, uint64_t hi, int tid);
typedef struct {
  range_fn fn;
  void *ctx;
  uint64_t lo, hi;
  int tid;
} job_t;

static void *job_main(void *p) {
  job_t *j = (job_t *)p;
  j->fn(j->ctx, j->lo, j->hi, j->tid);
  return NULL;
}

/* Run fn over [lo,hi) split into nthreads contiguous slices. */
static void parallel_range(range_fn fn, void *ctx, uint64_t lo, uint64_t hi, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  job_t jobs[64];
  pthread_t th[64];
  uint64_t span = hi - lo;
  for (int t = 0; t < nthreads; t++) {
    jobs[t].fn = fn;
    jobs[t].ctx = ctx;
    jobs[t].lo = lo + span * (uint64_t)t / (uint64_t)nthreads;
    jobs[t].hi = lo + span * (uint64_t)(t + 1) / (uint64_t)nthreads;
    jobs[t].tid = t;
  }
  for (int t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, job_main, &jobs[t]);
  job_main(&jobs[0]);
  for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------- exhaustive helpers */

typedef struct {
  int gamma;
  uint8_t *table;
} table_ctx;

static void table_range(void *p, uint64_t lo, uint64_t hi, int tid) {
  (void)tid;
  table_ctx *c = (table_ctx *)p;
  for (uint64_t i = lo; i < hi; i++) {
    int rgb[3];
    bt709o_decode_pixel(c->gamma, (int)(i >> 16), (int)((i >> 8) & 0xFF), (int)(i & 0xFF), rgb);
    c->table[3 * i + 0] = (uint8_t)rgb[0];
    c->table[3 * i + 1] = (uint8_t)rgb[1];
    c->table[3 * i + 2] = (uint8_t)rgb[2];
  }
}

void bt709o_decode_table(int gamma, uint8_t *table, int nthreads) {
  table_ctx c = {gamma, table};
  parallel_range(table_range, &c, 0, 1u << 24, nthreads);
}

typedef struct {
  int gamma;
  uint64_t hist[64][11];
} hist_ctx;

static void hist_range(void *p, uint64_t lo, uint64_t hi, int tid) {
  hist_ctx *c = (hist_ctx *)p;
  for (uint64_t i = lo; i < hi; i++) {
    int R = (int)(i >> 16), G = (int)((i >> 8) & 0xFF), B = (int)(i & 0xFF);
    int v[3], d[3];
    bt709o_encode_pixel(c->gamma, R, G, B, v);
    bt709o_decode_pixel(c->gamma, v[0], v[1], v[2], d);
    int e = abs(d[0] - R);
    if (abs(d[1] - G) > e) e = abs(d[1] - G);
    if (abs(d[2] - B) > e) e = abs(d[2] - B);
    if (e > 10) e = 10;
    c->hist[tid][e]++;
  }
}

void bt709o_roundtrip_histogram(int gamma, uint64_t hist[11], int nthreads) {
  hist_ctx *c = (hist_ctx *)calloc(1, sizeof(hist_ctx));
  c->gamma = gamma;
  parallel_range(hist_range, c, 0, 1u << 24, nthreads);
  for (int d = 0; d < 11; d++) {
    hist[d] = 0;
    for (int t = 0; t < 64; t++) hist[d] += c->hist[t][d];
  }
  free(c);
}

static float bits_to_float(uint32_t u) {
  float f;
  memcpy(&f, &u, sizeof f);
  return f;
}

void bt709o_thresholds(int gamma, float t[255]) {
  const uint32_t one = 0x3f800000u;
  for (int k = 1; k <= 255; k++) {
    /* smallest bit pattern in [0, one] whose byte is >= k */
    uint32_t lo = 0, hi = one;
    if (bt709o_transfer_to_byte(gamma, bits_to_float(hi)) < k) {
      t[k - 1] = INFINITY;
      continue;
    }
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (bt709o_transfer_to_byte(gamma, bits_to_float(mid)) >= k) hi = mid;
      else lo = mid + 1;
    }
    t[k - 1] = bits_to_float(lo);
  }
}

typedef struct {
  int gamma;
  float t[255];
  uint64_t bad[64];
} chk_ctx;

static void chk_range(void *p, uint64_t lo, uint64_t hi, int tid) {
  chk_ctx *c = (chk_ctx *)p;
  uint64_t bad = 0;
  int prev = -1;
  int k = 0; /* running count of thresholds <= x; x only grows */
  for (uint64_t u = lo; u < hi; u++) {
    float x = bits_to_float((uint32_t)u);
    int v = bt709o_transfer_to_byte(c->gamma, x);
    if (v < prev) bad++;
    prev = v;
    while (k < 255 && x >= c->t[k]) k++;
    if (k != v) bad++;
  }
  c->bad[tid] = bad;
}

uint64_t bt709o_check_thresholds(int gamma, uint32_t lo_bits, uint32_t hi_bits, int nthreads) {
  chk_ctx *c = (chk_ctx *)calloc(1, sizeof(chk_ctx));
  c->gamma = gamma;
  bt709o_thresholds(gamma, c->t);
  parallel_range(chk_range, c, lo_bits, (uint64_t)hi_bits + 1, nthreads);
  uint64_t bad = 0;
  for (int t = 0; t < 64; t++) bad += c->bad[t];
  free(c);
  return bad;
}
