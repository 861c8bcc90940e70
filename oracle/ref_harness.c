/*
 * ref_harness.c -- thin exported wrappers around the REFERENCE's own header-only
 * colour math, compiled from where it lies (-I/root/reference/Renderer).
 * TEST INFRASTRUCTURE ONLY.  Output goes to oracle/_ref/ (git-ignored).
 *
 * Nothing of the reference is copied here: this file only #includes
 * Renderer/BT709.h (which includes Renderer/sRGB.h) and forwards to its
 * `static inline` functions so that Python (ctypes) can call them.
 *
 * The one thing the headers need that plain C lacks is the Objective-C BOOL /
 * TRUE / FALSE spelling of int/1/0 (BT709.h:1051, 1180 -- debug flags only);
 * oracle/Makefile passes them as -D macros.  DEBUG is left undefined so the
 * headers' range asserts stay off, as in the reference's Release build.
 *
 * Must be compiled as C (not C++): the headers rely on pow(double,double).
 */
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>
#include <fcntl.h>

#include "BT709.h"

/* gamma ids shared with bt709_oracle.h: 0 Apple, 1 sRGB, 2 Linear, 3 ITU709 */

void ref_decode_pixel(int gamma, int Y, int Cb, int Cr, int rgb[3]) {
  int R = -1, G = -1, B = -1;
  if (gamma == 0) {
    Apple196_to_sRGB_convertYCbCrToRGB(Y, Cb, Cr, &R, &G, &B, 1);
  } else if (gamma == 1) {
    sRGB_to_sRGB_convertYCbCrToRGB(Y, Cb, Cr, &R, &G, &B, 1);
  } else if (gamma == 2) {
    float Rn, Gn, Bn;
    BT709_convertYCbCrToNonLinearRGB(Y, Cb, Cr, &Rn, &Gn, &Bn);
    R = BT709_from_linear(Rn, BT709GammaSrgb);
    G = BT709_from_linear(Gn, BT709GammaSrgb);
    B = BT709_from_linear(Bn, BT709GammaSrgb);
  } else {
    BT709_to_sRGB_convertYCbCrToRGB(Y, Cb, Cr, &R, &G, &B, 1);
  }
  rgb[0] = R;
  rgb[1] = G;
  rgb[2] = B;
}

void ref_encode_pixel(int gamma, int R, int G, int B, int ycbcr[3]) {
  int Y = -1, Cb = -1, Cr = -1;
  if (gamma == 0) {
    Apple196_from_sRGB_convertRGBToYCbCr(R, G, B, &Y, &Cb, &Cr);
  } else if (gamma == 1) {
    sRGB_from_sRGB_convertRGBToYCbCr(R, G, B, &Y, &Cb, &Cr);
  } else if (gamma == 2) {
    float Rn = sRGB_nonLinearNormToLinear(byteNorm(R));
    float Gn = sRGB_nonLinearNormToLinear(byteNorm(G));
    float Bn = sRGB_nonLinearNormToLinear(byteNorm(B));
    BT709_convertNonLinearRGBToYCbCr(Rn, Gn, Bn, &Y, &Cb, &Cr);
  } else {
    BT709_from_sRGB_convertRGBToYCbCr(R, G, B, &Y, &Cb, &Cr, 1);
  }
  ycbcr[0] = Y;
  ycbcr[1] = Cb;
  ycbcr[2] = Cr;
}

void ref_ycbcr_to_rgbn(int Y, int Cb, int Cr, float rgbn[3]) {
  BT709_convertYCbCrToNonLinearRGB(Y, Cb, Cr, &rgbn[0], &rgbn[1], &rgbn[2]);
}

int ref_decode_alpha(int A) {
  float Rn, Gn, Bn;
  BT709_convertYCbCrToNonLinearRGB(A, 128, 128, &Rn, &Gn, &Bn);
  return (int)round(Rn * 255.0f);
}

/* composite per-channel map applied after matrix+saturate, built only from
 * reference functions */
int ref_transfer_to_byte(int gamma, float v) {
  if (gamma == 0) v = sRGB_linearNormToNonLinear(Apple196_nonLinearNormToLinear(v));
  else if (gamma == 2) v = sRGB_linearNormToNonLinear(v);
  else if (gamma == 3) v = sRGB_linearNormToNonLinear(BT709_nonLinearNormToLinear(v));
  return (int)round(v * 255.0f);
}

float ref_srgb_to_linear(float v) { return sRGB_nonLinearNormToLinear(v); }
float ref_linear_to_srgb(float v) { return sRGB_linearNormToNonLinear(v); }
float ref_itu709_to_linear(float v) { return BT709_nonLinearNormToLinear(v); }
float ref_linear_to_itu709(float v) { return BT709_linearNormToNonLinear(v); }
float ref_apple196_to_linear(float v) { return Apple196_nonLinearNormToLinear(v); }
float ref_linear_to_apple196(float v) { return Apple196_linearNormToNonLinear(v); }

/* index (Y<<16)+(Cb<<8)+Cr, 3 bytes R,G,B per entry; rows [y0,y1) of Y */
void ref_decode_table(int gamma, uint8_t *table, int y0, int y1) {
  for (int Y = y0; Y < y1; Y++)
    for (int Cb = 0; Cb < 256; Cb++)
      for (int Cr = 0; Cr < 256; Cr++) {
        int rgb[3];
        ref_decode_pixel(gamma, Y, Cb, Cr, rgb);
        size_t i = ((size_t)Y << 16) + ((size_t)Cb << 8) + (size_t)Cr;
        table[3 * i + 0] = (uint8_t)rgb[0];
        table[3 * i + 1] = (uint8_t)rgb[1];
        table[3 * i + 2] = (uint8_t)rgb[2];
      }
}

/* worst-channel round-trip error histogram over R in [r0,r1) */
void ref_roundtrip_histogram(int gamma, uint64_t hist[11], int r0, int r1) {
  for (int d = 0; d < 11; d++) hist[d] = 0;
  for (int R = r0; R < r1; R++)
    for (int G = 0; G < 256; G++)
      for (int B = 0; B < 256; B++) {
        int v[3], o[3];
        ref_encode_pixel(gamma, R, G, B, v);
        ref_decode_pixel(gamma, v[0], v[1], v[2], o);
        int e = abs(o[0] - R);
        if (abs(o[1] - G) > e) e = abs(o[1] - G);
        if (abs(o[2] - B) > e) e = abs(o[2] - B);
        hist[e > 10 ? 10 : e]++;
      }
}

static int to_ref_gamma(int g) {
  return g == 0 ? BT709GammaApple : (g == 1 ? BT709GammaSrgb : BT709GammaLinear);
}

/* BT709_average_pixel_values has `debug = 1` left on (BT709.h:1373) and prints
 * several lines per call; park stdout on /dev/null while a batch runs. */
static int saved_stdout = -1;
void ref_quiet_begin(void) {
  fflush(stdout);
  saved_stdout = dup(1);
  int nul = open("/dev/null", O_WRONLY);
  dup2(nul, 1);
  close(nul);
}
void ref_quiet_end(void) {
  fflush(stdout);
  if (saved_stdout >= 0) {
    dup2(saved_stdout, 1);
    close(saved_stdout);
    saved_stdout = -1;
  }
}

/* rgb = {R1,G1,B1,...,R4,G4,B4}; out = {Y1,Y2,Y3,Y4,Cb,Cr} */
void ref_subsample_block(const int rgb[12], int in_gamma, int out_gamma, int out[6]) {
  BT709_average_pixel_values(rgb[0], rgb[1], rgb[2], rgb[3], rgb[4], rgb[5],
                             rgb[6], rgb[7], rgb[8], rgb[9], rgb[10], rgb[11],
                             &out[0], &out[1], &out[2], &out[3], &out[4], &out[5],
                             to_ref_gamma(in_gamma), to_ref_gamma(out_gamma));
}

/* BGRA words -> tight NV12 via repeated BT709_average_pixel_values, walking
 * blocks the way cvpbu_ycbcr_subsample does (that function itself needs
 * CoreVideo and cannot be built here). */
void ref_encode_nv12(const uint32_t *bgra, int width, int height, int in_gamma,
                     int out_gamma, uint8_t *y, uint8_t *uv) {
  ref_quiet_begin();
  for (int row = 0; row < height; row += 2)
    for (int col = 0; col < width; col += 2) {
      int rgb[12], out[6];
      const uint32_t p[4] = {bgra[row * width + col], bgra[row * width + col + 1],
                             bgra[(row + 1) * width + col], bgra[(row + 1) * width + col + 1]};
      for (int i = 0; i < 4; i++) {
        rgb[3 * i] = (p[i] >> 16) & 0xFF;
        rgb[3 * i + 1] = (p[i] >> 8) & 0xFF;
        rgb[3 * i + 2] = p[i] & 0xFF;
      }
      ref_subsample_block(rgb, in_gamma, out_gamma, out);
      y[row * width + col] = (uint8_t)out[0];
      y[row * width + col + 1] = (uint8_t)out[1];
      y[(row + 1) * width + col] = (uint8_t)out[2];
      y[(row + 1) * width + col + 1] = (uint8_t)out[3];
      uv[(row / 2) * width + col] = (uint8_t)out[4];
      uv[(row / 2) * width + col + 1] = (uint8_t)out[5];
    }
  ref_quiet_end();
}

/* NV12 rows [row0,row1) through the reference's per-pixel function, walking pixels
 * the way unconvertSoftware does (BGRAToBT709Converter.m:146-198) with chroma taken
 * at row/2, col/2 (.m:267-277).  Used as the "reference" CPU baseline by bench.py. */
void ref_decode_nv12_rows(int gamma, const uint8_t *y, size_t y_stride, const uint8_t *uv,
                          size_t uv_stride, int width, int row0, int row1, uint8_t *bgra,
                          size_t bgra_stride, int alpha_fill) {
  for (int row = row0; row < row1; row++) {
    const uint8_t *yr = y + (size_t)row * y_stride;
    const uint8_t *cr = uv + (size_t)(row / 2) * uv_stride;
    uint32_t *o = (uint32_t *)(bgra + (size_t)row * bgra_stride);
    for (int col = 0; col < width; col++) {
      int rgb[3];
      ref_decode_pixel(gamma, yr[col], cr[2 * (col / 2)], cr[2 * (col / 2) + 1], rgb);
      o[col] = ((uint32_t)alpha_fill << 24) | ((uint32_t)rgb[0] << 16) | ((uint32_t)rgb[1] << 8) | (uint32_t)rgb[2];
    }
  }
}

/* ---- pass 2 (MetalScaleRenderContext -renderScaled: + samplingShader), composed ONLY of the
 * reference's own inlines.  The reference runs pass 2 on sampler hardware and has no CPU twin, so
 * the composition (its normative definition) is written here once, next to the
 * functions it is made of, and its outputs pin both the oracle's restatement and the GPU kernels:
 *   tap    = 8-bit sRGB bytes of pass 1 (ref_decode_pixel above: Apple196_to_sRGB_... etc.)
 *   sample = sRGB_nonLinearNormToLinear(byteNorm(byte))        Renderer/sRGB.h:32-57 (what sampling an sRGB8 texel yields)
 *   filter = (((a+b)+c)+d)*0.25f for the exact 2:1 ratio; bilinear weights in general
 *   store  = (int)round(sRGB_linearNormToNonLinear(v) * 255.0f)  Renderer/sRGB.h:62-74, BT709.h:881-883
 * The alpha channel of the intermediate is a plain unorm: sample = byteNorm(a), store = round(v*255). */
static int ref_store_srgb(float linear) { return (int)round(sRGB_linearNormToNonLinear(linear) * 255.0f); }
static int ref_store_unorm(float v) { return (int)round(saturatef(v) * 255.0f); }

void ref_decode_nv12_half(int gamma, const uint8_t *y, size_t y_stride, const uint8_t *uv, size_t uv_stride,
                          const uint8_t *alpha, size_t alpha_stride, int width, int height, uint8_t *bgra,
                          size_t bgra_stride, int alpha_fill) {
  for (int orow = 0; orow < height / 2; orow++)
    for (int ocol = 0; ocol < width / 2; ocol++) {
      const int cb = uv[(size_t)orow * uv_stride + 2 * ocol], cr = uv[(size_t)orow * uv_stride + 2 * ocol + 1];
      int p[4][3], a[4];
      for (int t = 0; t < 4; t++) {
        const size_t row = 2 * orow + (t >> 1), col = 2 * ocol + (t & 1);
        ref_decode_pixel(gamma, y[row * y_stride + col], cb, cr, p[t]);
        a[t] = alpha ? ref_decode_alpha(alpha[row * alpha_stride + col]) : alpha_fill;
      }
      uint8_t *o = bgra + (size_t)orow * bgra_stride + 4 * ocol;
      for (int c = 0; c < 3; c++) {
        const float s = (((sRGB_nonLinearNormToLinear(byteNorm(p[0][c])) + sRGB_nonLinearNormToLinear(byteNorm(p[1][c]))) +
                          sRGB_nonLinearNormToLinear(byteNorm(p[2][c]))) + sRGB_nonLinearNormToLinear(byteNorm(p[3][c]))) * 0.25f;
        o[2 - c] = (uint8_t)ref_store_srgb(s);
      }
      o[3] = alpha ? (uint8_t)ref_store_unorm((((byteNorm(a[0]) + byteNorm(a[1])) + byteNorm(a[2])) + byteNorm(a[3])) * 0.25f)
                   : (uint8_t)alpha_fill;
    }
}

/* Any output size: texel-centre sampling, clamp to edge, weights w00 = (1-fx)(1-fy) ..., summed
 * (((w00*a + w01*b) + w10*c) + w11*d).  The sampling geometry is OUR definition (the reference
 * leaves it to the sampler); every transfer function and the quantisation are the reference's. */
void ref_decode_nv12_scaled(int gamma, const uint8_t *y, size_t y_stride, const uint8_t *uv, size_t uv_stride,
                            const uint8_t *alpha, size_t alpha_stride, int width, int height, uint8_t *bgra,
                            size_t bgra_stride, int out_width, int out_height, int alpha_fill) {
  const float scale_x = (float)width / (float)out_width, scale_y = (float)height / (float)out_height;
  for (int oy = 0; oy < out_height; oy++) {
    const float sy = ((float)oy + 0.5f) * scale_y - 0.5f;
    const float y0f = floorf(sy), fy = sy - y0f, gy = 1.0f - fy;
    int ys[2] = {(int)y0f, (int)y0f + 1};
    for (int i = 0; i < 2; i++) ys[i] = ys[i] < 0 ? 0 : (ys[i] > height - 1 ? height - 1 : ys[i]);
    for (int ox = 0; ox < out_width; ox++) {
      const float sx = ((float)ox + 0.5f) * scale_x - 0.5f;
      const float x0f = floorf(sx), fx = sx - x0f, gx = 1.0f - fx;
      int xs[2] = {(int)x0f, (int)x0f + 1};
      for (int i = 0; i < 2; i++) xs[i] = xs[i] < 0 ? 0 : (xs[i] > width - 1 ? width - 1 : xs[i]);
      const float w[4] = {gx * gy, fx * gy, gx * fy, fx * fy};
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int t = 0; t < 4; t++) {
        const int xx = xs[t & 1], yy = ys[t >> 1];
        const uint8_t *c = uv + (size_t)(yy / 2) * uv_stride + 2 * (xx / 2);
        int p[3];
        ref_decode_pixel(gamma, y[(size_t)yy * y_stride + xx], c[0], c[1], p);
        for (int k = 0; k < 3; k++) {
          const float term = w[t] * sRGB_nonLinearNormToLinear(byteNorm(p[k]));
          acc[k] = t ? acc[k] + term : term;
        }
        if (alpha) {
          const float term = w[t] * byteNorm(ref_decode_alpha(alpha[(size_t)yy * alpha_stride + xx]));
          acc[3] = t ? acc[3] + term : term;
        }
      }
      uint8_t *o = bgra + (size_t)oy * bgra_stride + 4 * ox;
      for (int k = 0; k < 3; k++) o[2 - k] = (uint8_t)ref_store_srgb(acc[k]);
      o[3] = alpha ? (uint8_t)ref_store_unorm(acc[3]) : (uint8_t)alpha_fill;
    }
  }
}

/* ---- RGBA16Float intermediate (Renderer/AAPLRenderer.m:143-170) ---------------------------------
 * Pass 1 writes the shader's linear-light float4 into an RGBA16Float texture.  Composition: the
 * reference's CPU matrix step (BT709_convertYCbCrToNonLinearRGB) and its own curve functions, then
 * the store's float -> half conversion (round to nearest even; C has no portable binary16 type, so
 * the conversion is spelled with exact double arithmetic). */
static uint16_t ref_float_to_half(float v) {
  const double a = fabs((double)v);
  const uint16_t sign = signbit(v) ? 0x8000u : 0u;
  if (a >= 65520.0) return (uint16_t)(sign | 0x7c00u);
  int e;
  (void)frexp(a, &e);
  int ue = e - 11;
  if (ue < -24) ue = -24;
  const double r = ldexp(nearbyint(ldexp(a, -ue)), ue);
  if (r == 0.0) return sign;
  int re;
  const double rm = frexp(r, &re);
  if (re - 1 < -14) return (uint16_t)(sign | (uint16_t)ldexp(r, 24));
  return (uint16_t)(sign | ((uint16_t)(re + 14) << 10) | ((uint16_t)ldexp(rm, 11) & 0x3ffu));
}

static float ref_curve_to_linear(int gamma, float v) {
  if (gamma == 0) return Apple196_nonLinearNormToLinear(v);
  if (gamma == 1) return sRGB_nonLinearNormToLinear(v);
  if (gamma == 3) return BT709_nonLinearNormToLinear(v);
  return v;
}

/* index (Y<<16)+(Cb<<8)+Cr, three halves R,G,B per entry; rows [y0,y1) of Y */
void ref_rgba16f_table(int gamma, uint16_t *table, int y0, int y1) {
  for (int Y = y0; Y < y1; Y++)
    for (int Cb = 0; Cb < 256; Cb++)
      for (int Cr = 0; Cr < 256; Cr++) {
        float n[3];
        BT709_convertYCbCrToNonLinearRGB(Y, Cb, Cr, &n[0], &n[1], &n[2]);
        const size_t i = ((size_t)Y << 16) + ((size_t)Cb << 8) + (size_t)Cr;
        for (int k = 0; k < 3; k++) table[3 * i + k] = ref_float_to_half(ref_curve_to_linear(gamma, n[k]));
      }
}

/* linear alpha sample as a half: BT709_decodeAlpha's CPU analogue (R of (A,128,128)), stored unquantised */
int ref_alpha_half(int A) {
  float Rn, Gn, Bn;
  BT709_convertYCbCrToNonLinearRGB(A, 128, 128, &Rn, &Gn, &Bn);
  return ref_float_to_half(Rn);
}
