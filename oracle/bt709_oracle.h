/*
 * bt709_oracle.h -- CPU oracle for the BT.709 NV12 -> sRGB BGRA decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (metalbt709decoder_amd/,
 * include/, host/) may include, link or execute this file.  It is used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and only as
 * the checker.
 *
 * The functions below restate, in plain C99, the arithmetic of the reference's
 * CPU colour path (paths relative to /root/reference):
 *   Renderer/sRGB.h:18-74            saturatef, byteNorm, sRGB transfer pair
 *   Renderer/BT709.h:40-59           Kr/Kg/Kb, video-range constants
 *   Renderer/BT709.h:68-151          ITU BT.709 and Apple 1.961 transfer pairs
 *   Renderer/BT709.h:199-341         RGB -> YCbCr (encode side, used for fixtures)
 *   Renderer/BT709.h:348-513         YCbCr -> non-linear RGB (matrix + saturate)
 *   Renderer/BT709.h:668-738, 821-908, 948-1003, 1150-1167   per-gamma decode entry points
 *   Renderer/BT709.h:1349-1509       2x2 block averaging used when subsampling
 *   Renderer/BGRAToBT709Converter.m:146-198, 256-287, 1042-1099   frame loops
 *   Renderer/CVPixelBufferUtils.h:241-399   RGB -> NV12 subsample loop
 *
 * Pinning: tests/test_oracle_golden.py checks this oracle against the
 * reference's own test vectors, its three exhaustive 2^24 round-trip histograms
 * and full-table hashes produced by the reference headers themselves
 * (oracle/_ref, built by oracle/Makefile from /root/reference in place).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (never -ffast-math, never C++:
 * the reference relies on C's pow(double,double) promotion).
 */
#ifndef BT709_ORACLE_H
#define BT709_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Transfer-curve selector.  0..2 mirror MetalBT709Gamma
 * (Renderer/MetalBT709Decoder.h:15-19); 3 is the true ITU curve that the
 * reference keeps as a dead branch (BGRAToBT709Converter.m:175-183). */
enum {
  BT709O_GAMMA_APPLE = 0,
  BT709O_GAMMA_SRGB = 1,
  BT709O_GAMMA_LINEAR = 2,
  BT709O_GAMMA_ITU709 = 3,
  BT709O_GAMMA_COUNT = 4,
  /* not a decode mode: the encoder-side composite round(255*Apple196enc(v)) =
   * BT709_from_linear(v, BT709GammaApple) (BT709.h:1150-1167); accepted by
   * bt709o_transfer_to_byte / bt709o_thresholds / bt709o_check_thresholds only */
  BT709O_TABLE_ENCODE_APPLE = 4
};

/* ---- scalar transfer functions (normalised float in, normalised float out) */
float bt709o_srgb_to_linear(float v);     /* sRGB.h:43-57   */
float bt709o_linear_to_srgb(float v);     /* sRGB.h:62-74   */
float bt709o_itu709_to_linear(float v);   /* BT709.h:68-81  */
float bt709o_linear_to_itu709(float v);   /* BT709.h:90-103 */
float bt709o_apple196_to_linear(float v); /* BT709.h:125-137 */
float bt709o_linear_to_apple196(float v); /* BT709.h:139-151 */

/* Quantise a normalised float to a byte: (int)round(v * 255.0f)
 * (BT709.h:881-883 and every sibling entry point). */
int bt709o_quantize(float v);

/* Composite "pre-gamma float -> output byte" map of one decode mode, i.e. what
 * happens to each channel after the matrix+saturate step. */
int bt709o_transfer_to_byte(int gamma, float v);

/* ---- per pixel */
/* BT709.h:466-513 + 348-460 (unscale=1): integer YCbCr -> saturated
 * non-linear normalised RGB. */
void bt709o_ycbcr_to_rgbn(int Y, int Cb, int Cr, float rgbn[3]);

/* Full per-pixel decode for one gamma mode; rgb[] receives R,G,B in 0..255.
 * APPLE  = Apple196_to_sRGB_convertYCbCrToRGB   (BT709.h:821-908)
 * SRGB   = sRGB_to_sRGB_convertYCbCrToRGB       (BT709.h:948-1003)
 * LINEAR = BT709_convertYCbCrToNonLinearRGB + BT709_from_linear(Srgb) (466-513, 1150-1167)
 * ITU709 = BT709_to_sRGB_convertYCbCrToRGB      (BT709.h:668-738) */
void bt709o_decode_pixel(int gamma, int Y, int Cb, int Cr, int rgb[3]);

/* Linear alpha plane sample -> alpha byte: the R channel of
 * BT709_convertYCbCrToNonLinearRGB(A,128,128) quantised, which is what
 * BT709_decodeAlpha (AAPLShaders.metal:249-271) feeds the 8-bit alpha channel. */
int bt709o_decode_alpha(int A);

/* Encode one pixel.  gamma selects the encoder twin:
 * APPLE  = Apple196_from_sRGB_convertRGBToYCbCr (BT709.h:743-817)
 * SRGB   = sRGB_from_sRGB_convertRGBToYCbCr     (BT709.h:914-944)
 * ITU709 = BT709_from_sRGB_convertRGBToYCbCr(...,1) (BT709.h:610-664)
 * LINEAR = sRGB -> linear -> matrix (no video curve). */
void bt709o_encode_pixel(int gamma, int R, int G, int B, int ycbcr[3]);

/* BT709_convertRGBToYCbCr / BT709_convertYCbCrToRGB (BT709.h:327-341, 558-605):
 * bytes are treated as LINEAR light; apply_curve selects the ITU curve on the
 * way in / out.  Off the decoder's path; kept because the reference's direct
 * C tests (CoreImageMetalFilterTests.m:94-357) pin the ITU transfer pair
 * through them. */
void bt709o_encode_linear_pixel(int apply_curve, int R, int G, int B, int ycbcr[3]);
void bt709o_decode_to_linear_pixel(int apply_curve, int Y, int Cb, int Cr, int rgb[3]);

/* ---- frames */
/* NV12 (+ optional linear alpha plane) -> BGRA8.  Chroma is replicated
 * (row/2, col/2: BGRAToBT709Converter.m:267-277).  Output word is
 * (A<<24)|(R<<16)|(G<<8)|B; A = decoded alpha plane when alpha != NULL, else
 * alpha_fill (0xFF = Metal opaque path, AAPLShaders.metal:243; 0x00 =
 * unconvertSoftware, BGRAToBT709Converter.m:187-193).
 * Strides are in bytes.  width and height must be even; returns 0, or -1 on
 * odd dimensions (BGRAToBT709Converter.m:69-74). */
int bt709o_decode_nv12(int gamma,
                       const uint8_t *y, size_t y_stride,
                       const uint8_t *uv, size_t uv_stride,
                       const uint8_t *alpha, size_t alpha_stride,
                       int width, int height,
                       uint8_t *bgra, size_t bgra_stride,
                       int alpha_fill);

/* Same, rows [row0,row1) only (row0 even) -- lets a caller partition a frame
 * across host threads for the cpu_baseline timing. */
int bt709o_decode_nv12_rows(int gamma,
                            const uint8_t *y, size_t y_stride,
                            const uint8_t *uv, size_t uv_stride,
                            const uint8_t *alpha, size_t alpha_stride,
                            int width, int row0, int row1,
                            uint8_t *bgra, size_t bgra_stride,
                            int alpha_fill);

/* Two-pass-equivalent decode + exact 2:1 bilinear downscale
 * (MetalBT709Decoder pass 1 + MetalScaleRenderContext.m:55-105 /
 * AAPLShaders.metal:73-85 at a 2:1 ratio).  The reference has no CPU twin and
 * no test for pass 2: PARITY UNPINNED for this function; definition (SURVEY
 * 8a row 9): decode the 4 source pixels to 8-bit sRGB, linearise each byte
 * with sRGB_nonLinearNormToLinear(byteNorm(b)), (((a+b)+c)+d)*0.25f,
 * sRGB_linearNormToNonLinear, quantise.  width/height are SOURCE dimensions
 * and must be multiples of 4 (so the output is even).
 * alpha (may be NULL): linear alpha plane of an alpha clip.  Pass 2 reads the alpha channel of the
 * 8-bit intermediate as a plain unorm (AAPLShaders.metal:411-438 writes it un-premultiplied;
 * MetalScaleRenderContext.m:55-105 filters all four channels): out alpha =
 * quantise((((byteNorm(a0)+byteNorm(a1))+byteNorm(a2))+byteNorm(a3))*0.25f), a_i = decoded alpha bytes. */
int bt709o_decode_nv12_half(int gamma,
                            const uint8_t *y, size_t y_stride,
                            const uint8_t *uv, size_t uv_stride,
                            const uint8_t *alpha, size_t alpha_stride,
                            int width, int height,
                            uint8_t *bgra, size_t bgra_stride,
                            int alpha_fill);

/* Decode + bilinear rescale to any output size (pass 1 + MetalScaleRenderContext
 * -renderScaled: with its linear-filter sampler, AAPLShaders.metal:73-85).
 * PARITY UNPINNED (no CPU twin, no test, sampler arithmetic is hardware's); our
 * definition: texel-centre sampling sx = (ox+0.5f)*(W/OW) - 0.5f, clamp to edge,
 * taps decoded to 8-bit sRGB and linearised like bt709o_decode_nv12_half, weights
 * w00 = (1-fx)(1-fy) ..., v = (((w00*a + w01*b) + w10*c) + w11*d), sRGB-encode,
 * quantise.  At an exact 2:1 ratio all weights are 0.25 and the result equals
 * bt709o_decode_nv12_half bit for bit. */
int bt709o_decode_nv12_scaled(int gamma,
                              const uint8_t *y, size_t y_stride,
                              const uint8_t *uv, size_t uv_stride,
                              const uint8_t *alpha, size_t alpha_stride, /* NULL, or as in bt709o_decode_nv12_half */
                              int width, int height,
                              uint8_t *bgra, size_t bgra_stride,
                              int out_width, int out_height, int alpha_fill);

/* ---- RGBA16Float render targets (Renderer/AAPLRenderer.m:143-170: the intermediate the reference
 * falls back to where sRGB texture writes are unavailable) and the stand-alone pass 2.
 * PARITY: the reference has no CPU twin of either; the definition below is the shader's arithmetic
 * with BT709.h's CPU constants and transfer functions, pinned through oracle/ref_harness.c. */
uint16_t bt709o_float_to_half(float v); /* IEEE binary32 -> binary16, round to nearest even (a float4 stored to RGBA16Float) */
float bt709o_half_to_float(uint16_t h);
float bt709o_curve_to_linear(int gamma, float v); /* what BT709ToLinearSRGBKernel & siblings apply before the store */
/* Pass 1 into an RGBA16Float target: per pixel R,G,B = half(curve_to_linear(saturated non-linear
 * value)), A = half(linear alpha) or 1.0; 8 bytes per pixel, memory order R,G,B,A. */
int bt709o_decode_nv12_rgba16f(int gamma, const uint8_t *y, size_t y_stride, const uint8_t *uv, size_t uv_stride,
                               const uint8_t *alpha, size_t alpha_stride, int width, int height, uint8_t *rgba,
                               size_t rgba_stride);
/* Pass 2 alone (MetalScaleRenderContext.m:55-105 + samplingShader, AAPLShaders.metal:73-85):
 * in_format 0 = BGRA8 sRGB intermediate (taps linearised like the sRGB8 sampler, alpha a plain unorm),
 * 1 = RGBA16Float (taps already linear); out = BGRA8 sRGB of any size.  Sampling geometry, weights and
 * summation order as bt709o_decode_nv12_scaled; the sum is saturated before the store as a unorm
 * render target does. */
int bt709o_render_scaled(int in_format, const uint8_t *in, size_t in_stride, int width, int height, uint8_t *bgra,
                         size_t bgra_stride, int out_width, int out_height);

/* unconvertSoftware (BGRAToBT709Converter.m:146-198): packed
 * (Cr<<16)|(Cb<<8)|Y words -> (R<<16)|(G<<8)|B words, alpha byte 0. */
int bt709o_unconvert_packed(int gamma, const uint32_t *ycbcr, uint32_t *bgra,
                            int width, int height);
/* convertSoftware (BGRAToBT709Converter.m:89-144). */
int bt709o_convert_packed(int gamma, const uint32_t *bgra, uint32_t *ycbcr,
                          int width, int height);

/* copyBT709ToCoreVideo (BGRAToBT709Converter.m:1042-1099): packed -> NV12.
 * CbCr of every even column of EVERY row is written to row/2, so the odd row's
 * value survives. */
void bt709o_packed_to_nv12(const uint32_t *ycbcr, int width, int height,
                           uint8_t *y, size_t y_stride,
                           uint8_t *uv, size_t uv_stride);
/* De-subsample loop of convertVimage (BGRAToBT709Converter.m:256-287). */
void bt709o_nv12_to_packed(const uint8_t *y, size_t y_stride,
                           const uint8_t *uv, size_t uv_stride,
                           int width, int height, uint32_t *ycbcr);

/* BT709_average_pixel_values (BT709.h:1349-1509): one 2x2 block of gamma
 * encoded RGB -> 4 Y + averaged Cb,Cr.  in_gamma/out_gamma use this file's
 * enum (APPLE/SRGB/LINEAR).  rgb = {R1,G1,B1,...,R4,G4,B4}. */
void bt709o_subsample_block(const int rgb[12], int in_gamma, int out_gamma,
                            int y4[4], int *cb, int *cr);
/* cvpbu_ycbcr_subsample (CVPixelBufferUtils.h:241-399): BGRA frame -> NV12. */
int bt709o_encode_nv12(const uint32_t *bgra, int width, int height,
                       int in_gamma, int out_gamma,
                       uint8_t *y, size_t y_stride,
                       uint8_t *uv, size_t uv_stride);

/* ---- exhaustive helpers */
/* Full decode table: index (Y<<16)+(Cb<<8)+Cr, 3 bytes R,G,B per entry
 * (50 331 648 bytes).  nthreads <= 1 runs single-threaded. */
void bt709o_decode_table(int gamma, uint8_t *table, int nthreads);

/* Encode->decode round trip over all 2^24 sRGB colours; hist[d] counts colours
 * whose worst channel error max(|dR|,|dG|,|dB|) equals d (0 = "exact",
 * 1..9 = "off1".."off9", 10 = "offMore9").  Mirrors the exhaustive tests
 * CoreImageMetalFilterTests.m:420-537, 557-674, 696-813 (helpers 44-89). */
void bt709o_roundtrip_histogram(int gamma, uint64_t hist[11], int nthreads);

/* 255 thresholds t[k-1] = smallest float x in [0,1] with
 * transfer_to_byte(gamma,x) >= k, k = 1..255, found by bisection on the float
 * bit pattern (valid because the composite is monotone; see
 * bt709o_check_monotone). */
void bt709o_thresholds(int gamma, float t[255]);

/* Sweep every float bit pattern in [lo_bits, hi_bits] and count (a) places
 * where transfer_to_byte decreases and (b) disagreements with the threshold
 * table.  Returns violations + mismatches (0 = table is exact on the range). */
uint64_t bt709o_check_thresholds(int gamma, uint32_t lo_bits, uint32_t hi_bits,
                                 int nthreads);

#ifdef __cplusplus
}
#endif
#endif /* BT709_ORACLE_H */
