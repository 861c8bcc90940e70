/*
 * bt709hip.h -- C ABI of the MI355X (gfx950) BT.709 NV12 -> sRGB BGRA decode path: the calls that have a twin
 * in the reference (mdejong/MetalBT709Decoder; paths below are relative to that repository).
 *
 *   Renderer/MetalRenderContext.h:17-105       device / queue holder, texture alloc, upload, read-back
 *                                              -> bt709hip_context_*, _stream_*, _malloc/_free, _upload/_download
 *   Renderer/MetalBT709Decoder.h:15-19, 27-48  MetalBT709Gamma, gamma / hasAlphaChannel properties
 *                                              -> bt709hip_gamma, bt709hip_decoder_create
 *   Renderer/MetalBT709Decoder.h:56            -setupMetal                      -> bt709hip_decoder_setup
 *   Renderer/MetalBT709Decoder.h:65-72         -decodeBT709:alphaPixelBuffer:bgraSRGBTexture:commandBuffer:
 *                                              renderPassDescriptor:renderWidth:renderHeight:waitUntilCompleted:
 *                                              -> bt709hip_decode (bt709hip_decode_batch: the same over N frames)
 *   Renderer/MetalScaleRenderContext.h:34-40   -renderScaled:... (pass 2)       -> bt709hip_render_scaled; fused with
 *                                              pass 1: bt709hip_decode_half (exact 2:1), bt709hip_decode_scaled (any size)
 *   Renderer/BGRAToBT709Converter.h:34-46      +unconvert:...                   -> bt709hip_unconvert
 *   Renderer/BGRAToBT709Converter.h:73-76      +convertIntoCoreVideoBuffer:...  -> bt709hip_encode
 *   Renderer/y4m_writer.h:194-241              planar U, V <-> NV12 CbCr        -> bt709hip_(de)interleave_cbcr
 *
 * Everything WITHOUT a reference twin -- frame rings and their placement hunt, ring sets, the frame sharder, in-flight
 * pools, the coalescing submit, graphs, events, options, batched forms of the side paths, introspection -- is declared
 * in bt709hip_ext.h (same library, same ABI number).
 *
 * Plain C types only: pointers are DEVICE pointers unless a parameter says "host"; a stream is an opaque hipStream_t
 * passed as void* (NULL = the context's default stream); sizes and strides are bytes.  Every function returns a
 * bt709hip_status (0 = success) unless noted; the reference's BOOL is ok = (status == 0).
 *
 * Ownership (differs from the reference on purpose): the caller owns every buffer and stream; a decoder
 * holds only its lookup tables, keeps no per-frame state and may be used from several streams and threads at once.
 * Buffers must stay alive until the stream has passed the call that uses them.
 */
#ifndef BT709HIP_H
#define BT709HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of bt709hip.h + bt709hip_ext.h together: bumped whenever a struct layout or a signature changes or an
 * export is added.  Bindings compare it with bt709hip_abi_version() and refuse an older library. */
#define BT709HIP_VERSION 502

typedef struct bt709hip_context bt709hip_context; /* ~ MetalRenderContext */
typedef struct bt709hip_decoder bt709hip_decoder; /* ~ MetalBT709Decoder  */

typedef enum {
  BT709HIP_OK = 0,
  BT709HIP_ERR_INVALID_ARG = -1,    /* NULL pointer, unknown enum, negative size */
  BT709HIP_ERR_NOT_SETUP = -2,      /* decoder has no context (setupMetal would return FALSE, MetalBT709Decoder.m:48-54) */
  BT709HIP_ERR_SIZE_MISMATCH = -3,  /* out != in, render != in, alpha != in (.m:272-306) */
  BT709HIP_ERR_ODD_DIMENSIONS = -4, /* 4:2:0 needs even W,H (BGRAToBT709Converter.m:69-74) */
  BT709HIP_ERR_MATRIX = -5,         /* YCbCr matrix tag is not ITU_R_709_2 (.m:311-318) */
  BT709HIP_ERR_TRANSFER = -6,       /* transfer tag does not match the decoder's gamma (.m:320-353) */
  BT709HIP_ERR_ALPHA_TRANSFER = -7, /* alpha buffer is not tagged linear (.m:357-368) */
  BT709HIP_ERR_STRIDE = -8,         /* stride smaller than a row / misaligned output */
  BT709HIP_ERR_HIP = -9,            /* a HIP call failed: see bt709hip_last_hip_error */
  BT709HIP_ERR_NO_DEVICE = -10,     /* no such GPU */
  BT709HIP_ERR_UNSUPPORTED = -11    /* e.g. batch larger than BT709HIP_MAX_BATCH */
} bt709hip_status;

/* MetalBT709Gamma (MetalBT709Decoder.h:15-19).  ITU709 is an extension: the true ITU curve the reference keeps as a
 * dead branch (BGRAToBT709Converter.m:175-183). */
typedef enum {
  BT709HIP_GAMMA_APPLE = 0, /* default */
  BT709HIP_GAMMA_SRGB = 1,
  BT709HIP_GAMMA_LINEAR = 2,
  BT709HIP_GAMMA_ITU709 = 3
} bt709hip_gamma;

/* kCVImageBufferYCbCrMatrixKey values the reference looks at (.m:311-318) */
typedef enum {
  BT709HIP_MATRIX_UNSPECIFIED = 0,
  BT709HIP_MATRIX_ITU_R_709_2 = 1,
  BT709HIP_MATRIX_ITU_R_601_4 = 2,
  BT709HIP_MATRIX_SMPTE_240M = 3
} bt709hip_matrix_tag;

/* kCVImageBufferTransferFunctionKey values the reference looks at (.m:320-353) */
typedef enum {
  BT709HIP_TRANSFER_UNSPECIFIED = 0,
  BT709HIP_TRANSFER_ITU_R_709_2 = 1, /* required by GAMMA_APPLE and GAMMA_ITU709 */
  BT709HIP_TRANSFER_SRGB = 2,        /* required by GAMMA_SRGB */
  BT709HIP_TRANSFER_LINEAR = 3       /* required by GAMMA_LINEAR and by every alpha buffer */
} bt709hip_transfer_tag;

/* One 4:2:0 biplanar video-range frame ("420v", createCoreVideoYCbCrBuffer: BGRAToBT709Converter.m:471-494) plus the
 * two colour attachments the decoder validates.  Replaces CVPixelBufferRef.  Of an alpha frame only the y plane is
 * read (cvpbu_wrap_y_plane_as_metal_texture, CVPixelBufferUtils.h:82-116); cbcr may be NULL there. */
typedef struct {
  const void *y;      /* W x H bytes, row pitch y_stride               */
  size_t y_stride;
  const void *cbcr;   /* (W/2) x (H/2) byte pairs Cb,Cr; pitch cbcr_stride */
  size_t cbcr_stride;
  int32_t width;      /* luma width, even  */
  int32_t height;     /* luma height, even */
  int32_t matrix;     /* bt709hip_matrix_tag   */
  int32_t transfer;   /* bt709hip_transfer_tag */
} bt709hip_frame;

/* MTLPixelFormat of a render target: BGRA8Unorm_sRGB, or RGBA16Float holding LINEAR light where sRGB texture writes
 * are unavailable (Renderer/AAPLRenderer.m:143-170). */
typedef enum {
  BT709HIP_FORMAT_BGRA8_SRGB = 0, /* default: 4 bytes per pixel, gamma-encoded sRGB */
  BT709HIP_FORMAT_RGBA16F = 1     /* 8 bytes per pixel: IEEE binary16 R,G,B,A in that memory order, linear light */
} bt709hip_format;

/* Render target; replaces id<MTLTexture>.  BGRA8_SRGB: memory order B,G,R,A = little-endian word
 * (A<<24)|(R<<16)|(G<<8)|B (MetalBT709DecoderTests.m:47-52); 4-byte aligned, stride a multiple of 4 (16-byte alignment
 * of both enables the wide-store kernel).  RGBA16F: 8-byte aligned, stride a multiple of 8; accepted by
 * bt709hip_decode[_batch] as output and by bt709hip_render_scaled as input. */
typedef struct {
  void *bgra;
  size_t stride;
  int32_t width;
  int32_t height;
  int32_t format;   /* bt709hip_format; 0 = BGRA8_SRGB */
  int32_t reserved; /* must be 0 */
} bt709hip_surface;

#define BT709HIP_MAX_BATCH 32

/* ------------------------------------------------------------------ context */
/* MetalRenderContext -setupMetal (MetalRenderContext.m:36-74): bind HIP device `device_ordinal`, create the default stream. */
int bt709hip_context_create(int device_ordinal, bt709hip_context **out);
int bt709hip_context_destroy(bt709hip_context *ctx);
int bt709hip_device_count(void); /* visible GPUs (none is initialised by the call), or < 0 */
int bt709hip_abi_version(void);  /* BT709HIP_VERSION the library was compiled against */

/* Streams ~ MTLCommandQueue / -commandBuffer (MetalRenderContext.h:20): one per in-flight frame. */
int bt709hip_stream_create(bt709hip_context *ctx, void **stream);
int bt709hip_stream_destroy(bt709hip_context *ctx, void *stream);
int bt709hip_stream_synchronize(bt709hip_context *ctx, void *stream);

/* Device memory ~ make*Texture / fill* / get*TexturePixels (MetalRenderContext.h:62-105).  upload / download copy
 * `row_bytes` x `rows` (pitches in bytes) on `stream` (hipMemcpy2DAsync).  PAGEABLE host memory (malloc, a std::vector,
 * numpy): the call waits for the copy, like the reference's fill... / get...Pixels (MetalRenderContext.m:122-160).
 * PINNED host memory (bt709hip_host_alloc, bt709hip_ext.h): asynchronous -- the buffer must stay unchanged until the
 * stream has passed the copy.  Nothing is waited for while `stream` is being captured into a graph. */
int bt709hip_malloc(bt709hip_context *ctx, size_t bytes, void **dptr);
int bt709hip_free(bt709hip_context *ctx, void *dptr);
int bt709hip_memset(bt709hip_context *ctx, void *dptr, int value, size_t bytes, void *stream);
int bt709hip_upload(bt709hip_context *ctx, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch,
                    size_t row_bytes, size_t rows, void *stream);
int bt709hip_download(bt709hip_context *ctx, void *dst_host, size_t dst_pitch, const void *src_dev, size_t src_pitch,
                      size_t row_bytes, size_t rows, void *stream);

/* ------------------------------------------------------------------ decoder */
/* alloc/init + property assignment.  has_alpha != 0 forces gamma to SRGB exactly as -setupMetalRenderPipeline does
 * (MetalBT709Decoder.m:165-169).  ctx may be NULL (a decoder without a render context): setup / decode then fail with
 * ERR_NOT_SETUP, mirroring .m:48-54; attach one with bt709hip_decoder_set_context. */
int bt709hip_decoder_create(bt709hip_context *ctx, int gamma, int has_alpha, bt709hip_decoder **out);
int bt709hip_decoder_destroy(bt709hip_decoder *dec);
int bt709hip_decoder_set_context(bt709hip_decoder *dec, bt709hip_context *ctx);
/* Alpha byte written when the decoder has no alpha channel.  Default 0xFF (Metal opaque path, AAPLShaders.metal:243);
 * 0x00 reproduces unconvertSoftware's words (BGRAToBT709Converter.m:187-193). */
int bt709hip_decoder_set_alpha_fill(bt709hip_decoder *dec, int alpha_byte);
int bt709hip_decoder_get_gamma(const bt709hip_decoder *dec);
int bt709hip_decoder_has_alpha(const bt709hip_decoder *dec);             /* 1 / 0, or < 0 */
bt709hip_context *bt709hip_decoder_context(const bt709hip_decoder *dec); /* the render context it was given, or NULL */
/* -setupMetal: builds the exact transfer tables of the decoder's gamma in device memory.  Idempotent
 * (MetalBT709Decoder.m:66-70); implied by every decode (not while a stream is being captured: set up first). */
int bt709hip_decoder_setup(bt709hip_decoder *dec);

/* -decodeBT709:... (MetalBT709Decoder.h:65-72).  Enqueues ONE fused kernel (chroma replicate + YCbCr->RGB matrix +
 * exact transfer + 8-bit pack) on `stream`.  `alpha` is NULL unless the decoder has an alpha channel.
 * render_width / height must equal the frame size (pass 1 never scales: .m:284-290).  wait_until_completed != 0
 * synchronises the stream before returning (.m:486-489). */
int bt709hip_decode(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                    const bt709hip_surface *out, int render_width, int render_height, void *stream, int wait_until_completed);
/* The same operator over `count` independent frames of one geometry (same size, strides, tags) in ONE launch -- how a
 * stream of frames stays off the launch-latency floor (a 4K frame is a ~7.6 us kernel).  count <= BT709HIP_MAX_BATCH
 * in general (the plane pointers travel in the kernel arguments); when frames, alphas and outputs are EVENLY SPACED
 * in memory -- frame i at frame 0 + i * (frame 1 - frame 0), as in a ring carved from one allocation -- any count
 * up to 65535.  alphas may be NULL. */
int bt709hip_decode_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                          const bt709hip_surface *outs, void *stream, int wait_until_completed);

/* Pass 1 + pass 2 (-renderScaled:, bilinear) fused for the exact 2:1 ratio: out is (W/2) x (H/2); W, H multiples of 4.
 * Two-pass-equivalent arithmetic: each output channel is the linear-light mean of the four decoded 8-bit sRGB values,
 * re-encoded to sRGB; the alpha channel of an alpha decoder is filtered as a plain unorm. */
int bt709hip_decode_half(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                         const bt709hip_surface *out, void *stream, int wait_until_completed);
/* Pass 1 + pass 2 fused for ANY output size (view-fit, down or up): bilinear in linear light over the decoded 8-bit
 * sRGB values, texel-centre sampling, clamp to edge; at an exact 2:1 ratio bit-identical to bt709hip_decode_half.  The
 * reference leaves this arithmetic to the sampler hardware (AAPLShaders.metal:73-85): the definition is ours.
 * BT709HIP_ERR_UNSUPPORTED: a plane or the output of 2 GiB or more, or more than 65535 output rows. */
int bt709hip_decode_scaled(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                           const bt709hip_surface *out, void *stream, int wait_until_completed);
/* Pass 2 on its own: -[MetalScaleRenderContext renderScaled:...] (.h:34-40, .m:55-105) + samplingShader.  `in` is the
 * intermediate pass 1 rendered -- BGRA8_SRGB (taps linearised as the sRGB8 sampler does) or RGBA16F (linear already)
 * -- `out` a BGRA8_SRGB surface of any size.  Same taps, weights and summation order as bt709hip_decode_scaled: decode +
 * render_scaled through a BGRA8 intermediate equals the fused call bit for bit. */
int bt709hip_render_scaled(bt709hip_context *ctx, const bt709hip_surface *in, const bt709hip_surface *out, void *stream,
                           int wait_until_completed);

/* ---------------------------------------------------------------- converters */
/* +[BGRAToBT709Converter unconvert:outBGRAPixels:width:height:type:] on the GPU (BGRAToBT709Converter.h:34-46; the
 * Software type, .m:61-85 -> unconvertSoftware .m:146-198): `ycbcr_words` are PACKED 4:4:4 pixels, one 32-bit word
 * Y | Cb << 8 | Cr << 16 each, rows in_stride bytes apart; `out` receives (A<<24)|(R<<16)|(G<<8)|B with the decoder's
 * gamma and A = the decoder's alpha fill.  Odd width or height: BT709HIP_ERR_ODD_DIMENSIONS (.m:69-74); a decoder
 * with an alpha channel: BT709HIP_ERR_UNSUPPORTED. */
int bt709hip_unconvert(bt709hip_decoder *dec, const void *ycbcr_words, size_t in_stride, int width, int height,
                       const bt709hip_surface *out, void *stream, int wait_until_completed);
/* The step before the decode, on the GPU: 8-bit BGRA -> NV12 BT.709 video range with the reference's linear-light 2x2
 * chroma averaging.  Replaces +convertIntoCoreVideoBuffer:cvPixelBuffer:inputGamma:outputGamma:
 * (BGRAToBT709Converter.h:73-76, .m:532-569) -> cvpbu_ycbcr_subsample (CVPixelBufferUtils.h:241-399) ->
 * BT709_average_pixel_values (BT709.h:1349-1509).  Gammas are bt709hip_gamma values APPLE, SRGB or LINEAR; the app
 * encodes with (SRGB, APPLE), (SRGB, SRGB) or (LINEAR, LINEAR) (.m:919-935).  `in` is read (alpha ignored), the planes
 * `out` points to are written (its tags are ignored).  Even, equal sizes on both sides. */
int bt709hip_encode(bt709hip_context *ctx, const bt709hip_surface *in, const bt709hip_frame *out, int input_gamma,
                    int output_gamma, void *stream, int wait_until_completed);
/* The reference's on-disk 4:2:0 format is YUV4MPEG2 "C420jpeg": planar Y, U (Cb), V (Cr) per frame
 * (Renderer/y4m_writer.h:194-241).  These move the two chroma planes to / from NV12's interleaved CbCr plane on the
 * device (the Y plane is the same in both layouts).  chroma_width x chroma_height = (W/2) x (H/2). */
int bt709hip_interleave_cbcr(bt709hip_context *ctx, const void *u, size_t u_stride, const void *v, size_t v_stride,
                             void *cbcr, size_t cbcr_stride, int chroma_width, int chroma_height, void *stream,
                             int wait_until_completed);
int bt709hip_deinterleave_cbcr(bt709hip_context *ctx, const void *cbcr, size_t cbcr_stride, void *u, size_t u_stride,
                               void *v, size_t v_stride, int chroma_width, int chroma_height, void *stream,
                               int wait_until_completed);

/* ------------------------------------------------------------------- errors */
const char *bt709hip_strerror(int status);
int bt709hip_last_hip_error(void); /* hipError_t of the most recent failing HIP call on this thread (0 if none) */
const char *bt709hip_last_hip_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* BT709HIP_H */
