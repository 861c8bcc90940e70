/*
 * bt709hip.h -- C ABI of the MI355X (gfx950) BT.709 NV12 -> sRGB BGRA decode path.
 *
 * This is the drop-in boundary for the reference's decode operator
 * (paths relative to the reference repository, mdejong/MetalBT709Decoder):
 *
 *   Renderer/MetalRenderContext.h:17-105   device / queue holder, texture alloc,
 *                                          upload and read-back helpers
 *                                          -> bt709hip_context_*, bt709hip_malloc/free,
 *                                             bt709hip_upload/download, bt709hip_stream_*
 *   Renderer/MetalBT709Decoder.h:15-19     MetalBT709Gamma            -> bt709hip_gamma
 *   Renderer/MetalBT709Decoder.h:27-48     gamma / hasAlphaChannel /
 *                                          useComputeRenderer props   -> bt709hip_decoder_create
 *   Renderer/MetalBT709Decoder.h:56        -setupMetal                -> bt709hip_decoder_setup
 *   Renderer/MetalBT709Decoder.h:65-72     -decodeBT709:alphaPixelBuffer:bgraSRGBTexture:
 *                                           commandBuffer:renderPassDescriptor:renderWidth:
 *                                           renderHeight:waitUntilCompleted:
 *                                                                     -> bt709hip_decode
 *   Renderer/MetalScaleRenderContext.h:34-40  -renderScaled:... (pass 2), fused with pass 1
 *                                          for the exact 2:1 case     -> bt709hip_decode_half
 *
 * Plain C types only: pointers are DEVICE pointers unless a parameter says
 * "host"; a stream is an opaque hipStream_t passed as void*; sizes are bytes.
 * Every function returns a bt709hip_status (0 = success) unless noted.  The
 * reference's BOOL convention is one comparison away: ok = (status == 0).
 *
 * Ownership (differs from the reference on purpose): the caller
 * owns every buffer and stream; a decoder holds only its lookup table, keeps no
 * per-frame state, and may be used from several streams at once.  Buffers must
 * stay alive until the stream has passed the decode.
 */
#ifndef BT709HIP_H
#define BT709HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct layout or a signature in this file changes or an export is added (501:
 * bt709hip_gamma_lookup_decode; upload / download wait for pageable host memory; 502: bt709hip_ring_options.format).  Bindings compare
 * it with bt709hip_abi_version() so that a library older than the header is refused, not mis-called. */
#define BT709HIP_VERSION 502

typedef struct bt709hip_context bt709hip_context; /* ~ MetalRenderContext */
typedef struct bt709hip_decoder bt709hip_decoder; /* ~ MetalBT709Decoder  */
typedef struct bt709hip_pool bt709hip_pool;       /* ~ CVPixelBufferPool + texture cache + in-flight semaphore */

typedef enum {
  BT709HIP_OK = 0,
  BT709HIP_ERR_INVALID_ARG = -1,    /* NULL pointer, unknown enum, negative size */
  BT709HIP_ERR_NOT_SETUP = -2,      /* decoder has no context (setupMetal would return FALSE, .m:48-54) */
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

/* MetalBT709Gamma (MetalBT709Decoder.h:15-19).  ITU709 is an extension: the true
 * ITU curve the reference keeps as a dead branch (BGRAToBT709Converter.m:175-183). */
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

/* One 4:2:0 biplanar video-range frame ("420v", what createCoreVideoYCbCrBuffer
 * makes: BGRAToBT709Converter.m:471-494) plus the two colour attachments the
 * decoder validates.  Replaces CVPixelBufferRef.  For an alpha frame only the
 * y plane is read (cvpbu_wrap_y_plane_as_metal_texture, CVPixelBufferUtils.h:82-116);
 * cbcr may be NULL there. */
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

/* MTLPixelFormat of a render target.  The reference renders pass 1 into BGRA8Unorm_sRGB, or --
 * where sRGB texture writes are unavailable (macOS < 10.14) -- into RGBA16Float holding LINEAR
 * light (Renderer/AAPLRenderer.m:143-170). */
typedef enum {
  BT709HIP_FORMAT_BGRA8_SRGB = 0, /* default: 4 bytes per pixel, gamma-encoded sRGB */
  BT709HIP_FORMAT_RGBA16F = 1     /* 8 bytes per pixel: IEEE binary16 R,G,B,A in that memory order, linear light */
} bt709hip_format;

/* Render target; replaces id<MTLTexture>.
 * BGRA8_SRGB: memory order B,G,R,A i.e. little-endian word (A<<24)|(R<<16)|(G<<8)|B
 * (MetalBT709DecoderTests.m:47-52).  bgra must be 4-byte aligned, stride a
 * multiple of 4 (16-byte alignment of both enables the wide-store kernel).
 * RGBA16F: 8-byte aligned, stride a multiple of 8; accepted by bt709hip_decode[_batch] as output
 * and by bt709hip_render_scaled as input. */
typedef struct {
  void *bgra;
  size_t stride;
  int32_t width;
  int32_t height;
  int32_t format;   /* bt709hip_format; 0 = BGRA8_SRGB */
  int32_t reserved; /* must be 0 */
} bt709hip_surface;

typedef struct {
  int32_t device_ordinal;
  int32_t compute_units;
  int32_t wavefront_size;
  int32_t lds_bytes_per_block;
  int32_t memory_clock_khz;
  int32_t memory_bus_width_bits;
  int32_t l2_bytes;
  int32_t clock_khz;
  uint64_t total_memory_bytes;
  char name[128];
  char arch[64];
  /* Which physical device this is, for callers that must prove N contexts sit on N GPUs (bench.py's per-rank records; device
   * ordinals are per process and say nothing once HIP_VISIBLE_DEVICES differs between ranks): hipDeviceGetPCIBusId
   * ("0000:c1:00.0") and the 16 bytes of hipDeviceGetUuid -- as they are when they are printable text (ROCm: the 16 characters
   * rocm-smi shows as the unique id), as 32 hex digits otherwise.  MTLDevice has registryID for this
   * (the reference keeps one device, Renderer/MetalRenderContext.m:59-63, and never needs it). */
  char pci_bus_id[32];
  char uuid[40];
} bt709hip_device_info;

#define BT709HIP_MAX_BATCH 32

/* ------------------------------------------------------------------ context */
/* MetalRenderContext -setupMetal (MetalRenderContext.m:36-74): bind a device,
 * create the default stream.  device_ordinal is the HIP device index. */
int bt709hip_context_create(int device_ordinal, bt709hip_context **out);
int bt709hip_context_destroy(bt709hip_context *ctx);
int bt709hip_context_info(const bt709hip_context *ctx, bt709hip_device_info *info);
/* Number of visible GPUs; does not initialise any of them.  Returns count or <0. */
int bt709hip_device_count(void);
/* BT709HIP_VERSION the library was compiled against. */
int bt709hip_abi_version(void);

/* Launch-shape knobs of a context (tuning and test hooks; no reference twin).  Values are
 * clamped to their valid range; 0 restores the default. */
typedef enum {
  BT709HIP_CTX_OPT_GRID_MULT = 1,       /* general (unaligned-layout) kernels: workgroups per launch = CUs x 8 x this; default 2 */
  BT709HIP_CTX_OPT_ENCODE_ROW_PAIRS = 2,/* encoder: consecutive row pairs per workgroup; default 0 = sized per launch */
  BT709HIP_CTX_OPT_ENCODE_THREADS = 3,  /* encoder: lanes per workgroup (rounded down to whole waves); default 0 = from the width */
  BT709HIP_CTX_OPT_XCD_BANDS = 4,       /* encoder: 1 (default) XCD-aware work map for launches of a multiple of 8 pictures; 0 plain order */
  BT709HIP_CTX_OPT_STREAMING_TRIES = 5  /* device buffers of 256 MB or more that the library allocates itself (in-flight pool slots, the sharder's lanes) go through bt709hip_malloc_streaming with this many candidates; default 4, 1 = plain allocation */
} bt709hip_context_option;
int bt709hip_context_set_option(bt709hip_context *ctx, int option, int value);

/* Streams ~ MTLCommandQueue/-commandBuffer (MetalRenderContext.h:20): one per
 * in-flight frame.  `stream == NULL` anywhere below means the context's default. */
int bt709hip_stream_create(bt709hip_context *ctx, void **stream);
/* The same with a scheduling priority (hipStreamCreateWithPriority): 0 = normal, negative = higher, positive = lower; clamped
 * to the device's range.  (MTLCommandQueue has no twin; a renderer that decodes ahead of what it presents puts the look-ahead
 * frames on a lower-priority stream.) */
int bt709hip_stream_create_with_priority(bt709hip_context *ctx, int priority, void **stream);
int bt709hip_stream_destroy(bt709hip_context *ctx, void *stream);
int bt709hip_stream_synchronize(bt709hip_context *ctx, void *stream);

/* Events (timing only; no reference twin). elapsed: milliseconds start->stop. */
int bt709hip_event_create(bt709hip_context *ctx, void **event);
int bt709hip_event_destroy(bt709hip_context *ctx, void *event);
int bt709hip_event_record(bt709hip_context *ctx, void *event, void *stream);
int bt709hip_event_synchronize(bt709hip_context *ctx, void *event);
/* Makes `stream` wait for `event` (recorded on another stream): joins the per-frame streams of a
 * pipeline without blocking the host -- MTLCommandBuffer ordering across queues / encodeWaitForEvent:. */
int bt709hip_stream_wait_event(bt709hip_context *ctx, void *stream, void *event);
int bt709hip_event_elapsed_ms(bt709hip_context *ctx, void *start, void *stop, float *ms);

/* Recorded command buffers.  The reference encodes a frame's passes into an MTLCommandBuffer
 * and commits it (MetalBT709Decoder.h:65-72 takes the buffer; AAPLRenderer.m:891-977 builds one
 * per frame); the HIP twin of a command buffer that is recorded once and replayed is a graph.
 * Between begin and end every bt709hip_decode* / _encode* / upload / download / memset issued
 * on `stream` (a created stream, not NULL) is recorded instead of executed; `graph` then
 * replays them all with one launch -- for pipelines of small frames, where the per-launch host
 * cost exceeds the kernel (a 1080p decode is ~2 us of GPU time).  Decoders must have been set up
 * (bt709hip_decoder_setup) before capture begins; do not wait inside a capture. */
int bt709hip_graph_begin_capture(bt709hip_context *ctx, void *stream);
int bt709hip_graph_end_capture(bt709hip_context *ctx, void *stream, void **graph);
int bt709hip_graph_launch(bt709hip_context *ctx, void *graph, void *stream);
int bt709hip_graph_destroy(bt709hip_context *ctx, void *graph);

/* Device memory ~ make*Texture / fill* / get*TexturePixels
 * (MetalRenderContext.h:62-105).  upload/download are enqueued on `stream`
 * (hipMemcpy2DAsync).  With PINNED host memory (bt709hip_host_alloc) they are asynchronous:
 * the buffer must stay allocated and unchanged until the stream has passed the copy
 * (bt709hip_stream_synchronize, an event, or a later call with wait_until_completed on the
 * same stream).  With PAGEABLE host memory (malloc, a std::vector, numpy) the call waits for
 * the copy before it returns, like the reference's fill... / get...Pixels methods
 * (MetalRenderContext.m:122-160) -- the runtime would otherwise keep reading a buffer its
 * owner is free to release, which is a GPU memory access fault, not an error code.  (While
 * `stream` is being captured into a graph nothing is waited for.)  Pitches are bytes;
 * `row_bytes` x `rows` is copied. */
int bt709hip_malloc(bt709hip_context *ctx, size_t bytes, void **dptr);
int bt709hip_free(bt709hip_context *ctx, void *dptr);
/* Free and total device memory of the context's GPU right now (hipMemGetInfo), for callers that size rings or a placement
 * hunt (bt709hip_malloc_streaming) against what is left.  Either pointer may be NULL. */
int bt709hip_mem_info(bt709hip_context *ctx, size_t *free_bytes, size_t *total_bytes);
int bt709hip_host_alloc(bt709hip_context *ctx, size_t bytes, void **hptr);
int bt709hip_host_free(bt709hip_context *ctx, void *hptr);
int bt709hip_memset(bt709hip_context *ctx, void *dptr, int value, size_t bytes, void *stream);
int bt709hip_upload(bt709hip_context *ctx, void *dst_dev, size_t dst_pitch,
                    const void *src_host, size_t src_pitch,
                    size_t row_bytes, size_t rows, void *stream);
int bt709hip_download(bt709hip_context *ctx, void *dst_host, size_t dst_pitch,
                      const void *src_dev, size_t src_pitch,
                      size_t row_bytes, size_t rows, void *stream);

/* ------------------------------------------------------------------ decoder */
/* alloc/init + property assignment.  has_alpha != 0 forces gamma to SRGB exactly
 * as -setupMetalRenderPipeline does (MetalBT709Decoder.m:165-169).  ctx may be
 * NULL (a decoder without a render context): setup/decode then fail with
 * ERR_NOT_SETUP, mirroring .m:48-54; attach one with bt709hip_decoder_set_context. */
int bt709hip_decoder_create(bt709hip_context *ctx, int gamma, int has_alpha,
                            bt709hip_decoder **out);
int bt709hip_decoder_destroy(bt709hip_decoder *dec);
int bt709hip_decoder_set_context(bt709hip_decoder *dec, bt709hip_context *ctx);
/* Alpha byte written when the decoder has no alpha channel.  Default 0xFF (Metal
 * opaque path, AAPLShaders.metal:243); 0x00 reproduces unconvertSoftware's words
 * (BGRAToBT709Converter.m:187-193). */
int bt709hip_decoder_set_alpha_fill(bt709hip_decoder *dec, int alpha_byte);
int bt709hip_decoder_get_gamma(const bt709hip_decoder *dec);
int bt709hip_decoder_has_alpha(const bt709hip_decoder *dec);           /* 1 / 0, or <0 */
bt709hip_context *bt709hip_decoder_context(const bt709hip_decoder *dec); /* the render context it was given, or NULL */
/* Kernel-selection knobs of a decoder (tuning and test hooks; no reference twin).  They may be
 * changed between calls, not during one. */
typedef enum {
  BT709HIP_OPT_NONTEMPORAL = 1,      /* 1 (default): streaming loads / stores in the fast kernels; 0: default cache policy */
  BT709HIP_OPT_HALF_KERNEL = 2,      /* 2:1 rescale: -1 (default) persistent kernel when the launch is large enough, 0 never, 1 always */
  BT709HIP_OPT_HALF_WORKGROUPS = 3,  /* persistent 2:1 kernel: workgroups; 0 (default) = one per compute unit */
  BT709HIP_OPT_HALF_LDS_KB = 4,      /* persistent 2:1 kernel: KiB of LDS a workgroup may fill with table copies; 0 (default) = 160 */
  BT709HIP_OPT_XCD_BANDS = 5,        /* 1 (default): batched 1:1 launches of 64 frames or more give each XCD a contiguous band of the frames (a count that is not a multiple of 8: that map over the multiple of 8, the plain map over the rest); 0: plain (tile, row pair, frame) order */
  BT709HIP_OPT_COALESCE = 6,         /* 0 (default) off; n in 2..32: coalescing submit, see bt709hip_decode */
  BT709HIP_OPT_COALESCE_MAX_AGE_US = 7 /* 0 (default): queued frames wait for their stream's next call, however long; t > 0: a queue whose oldest frame was queued more than t microseconds ago is issued by the next bt709hip_* call that touches ANY stream of the context (or any decode of any decoder of it) */
} bt709hip_decoder_option;
int bt709hip_decoder_set_option(bt709hip_decoder *dec, int option, int value);
int bt709hip_decoder_get_option(const bt709hip_decoder *dec, int option, int *value);
/* -setupMetal: builds the exact transfer table for the decoder's gamma and puts
 * it in device memory.  Idempotent (MetalBT709Decoder.m:66-70); implied by decode. */
int bt709hip_decoder_setup(bt709hip_decoder *dec);

/* -decodeBT709:... (MetalBT709Decoder.h:65-72).  Enqueues ONE fused kernel
 * (chroma replicate + YCbCr->RGB matrix + exact transfer + 8-bit pack) on
 * `stream`.  `alpha` may be NULL.  render_width/height must equal the frame size
 * (pass 1 never scales: .m:284-290).  wait_until_completed != 0 synchronises the
 * stream before returning (.m:486-489). */
int bt709hip_decode(bt709hip_decoder *dec,
                    const bt709hip_frame *frame, const bt709hip_frame *alpha,
                    const bt709hip_surface *out,
                    int render_width, int render_height,
                    void *stream, int wait_until_completed);

/* COALESCING SUBMIT (extension, opt-in: bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, n), n = 2..32).
 * The reference's cadence is one -decodeBT709: call per frame (MetalBT709Decoder.h:65-72, AAPLRenderer.m:914-957), each call
 * encoding into the caller's command buffer; on an MI355X a 4K frame is a ~7.6 us kernel and a launch boundary on one stream
 * costs ~3.8 us of idle GPU, so that cadence reaches 0.49 of the roofline where one launch over many frames reaches 0.75-0.81.
 * With the option on, a 1:1 decode of device-resident frames with wait_until_completed == 0 -- bt709hip_decode, or
 * bt709hip_decode_batch with a count below n -- is VALIDATED at once (its status is the call's status, as before) but only
 * QUEUED: up to n frames of one geometry and target format per stream gather and go out as ONE bt709hip_decode_batch launch
 * (evenly spaced frames as a uniform batch, any others through the pointer table).  The queue of a stream is issued
 *   - when it holds n frames, or a call with another geometry / format / decoder state arrives for that stream,
 *   - by any bt709hip_* call that takes that stream (stream_synchronize, event_record, stream_wait_event, download, upload,
 *     memset, graph capture, copy_probe, a decode with wait_until_completed != 0, every other decode / encode / rescale
 *     entry point -- of THIS decoder or of any other decoder of the context, coalescing or not) -- so the stream keeps its
 *     order for everything issued through this API (a CPU test parses this header's `void *stream` exports and checks
 *     each one of them),
 *   - by bt709hip_decoder_flush, and when the decoder is destroyed or the option is turned off.
 * There is no timer thread: a queue is only ever issued from inside a bt709hip_* call.  A caller that may go idle with frames
 * queued either flushes before it does, or sets BT709HIP_OPT_COALESCE_MAX_AGE_US, which bounds the wait by the time to the
 * context's NEXT call of any kind (a renderer's per-frame bt709hip_stream_synchronize / event poll on another stream is enough).
 * The command-buffer analogy: queued frames are "encoded, not yet committed".  What the caller gives up: work submitted to
 * the raw hipStream_t behind this API's back (its own kernels, hipStreamSynchronize) is not ordered after queued frames --
 * call bt709hip_decoder_flush first.  Frame and surface descriptors are copied at the call; the buffers they point to must
 * stay alive until the stream has passed the launch, as always.  A launch failure at issue time is returned by the call that
 * issued the queue.  Thread safety: as without the option (several threads may share a decoder; each queue is per stream). */
int bt709hip_decoder_flush(bt709hip_decoder *dec, void *stream /* NULL = the context's default stream */);
/* every stream's queue of this decoder */
int bt709hip_decoder_flush_all(bt709hip_decoder *dec);

/* The same operator over `count` independent frames of one geometry (same
 * width/height/strides/tags) in ONE launch: grid.z = frame.  This is how a stream of
 * small frames stays off the launch-latency floor.  count <= BT709HIP_MAX_BATCH in
 * general (the plane pointers travel in the kernel-argument block); when the frames,
 * alphas and outputs are EVENLY SPACED in memory -- frame i at frame 0 + i * (frame 1
 * - frame 0), as in a ring carved from one allocation -- any count <= 65535 is accepted.
 * alphas may be NULL. */
int bt709hip_decode_batch(bt709hip_decoder *dec, int count,
                          const bt709hip_frame *frames, const bt709hip_frame *alphas,
                          const bt709hip_surface *outs,
                          void *stream, int wait_until_completed);

/* +[BGRAToBT709Converter unconvert:outBGRAPixels:width:height:type:] on the GPU
 * (Renderer/BGRAToBT709Converter.h:34-46; the Software type, .m:61-85 -> unconvertSoftware .m:146-198): `ycbcr_words` are
 * PACKED 4:4:4 pixels in device memory, one 32-bit word Y | Cb << 8 | Cr << 16 each (every pixel its own chroma), rows
 * in_stride bytes apart; `out` receives (A << 24) | (R << 16) | (G << 8) | B with the decoder's gamma (the reference
 * hard-selects Apple196, the default) and A = the decoder's alpha fill (bt709hip_decoder_set_alpha_fill(dec, 0)
 * reproduces unconvertSoftware's words).  Odd width or height -> BT709HIP_ERR_ODD_DIMENSIONS (.m:69-74); a decoder with
 * an alpha channel -> BT709HIP_ERR_UNSUPPORTED. */
int bt709hip_unconvert(bt709hip_decoder *dec, const void *ycbcr_words, size_t in_stride, int width, int height,
                       const bt709hip_surface *out, void *stream, int wait_until_completed);
/* The same over `count` frames of one geometry (same width, height and strides) in ONE launch (grid.z = frame), like
 * bt709hip_decode_batch: up to BT709HIP_MAX_BATCH arbitrary buffers, or any number up to 65535 when frame i sits at frame 0 +
 * i * (frame 1 - frame 0) on both sides.  No reference twin (+unconvert: takes one frame); one 4K frame per call runs at 0.58 of
 * the roofline on one stream -- a launch boundary per 14 us kernel -- a batch does not pay it. */
int bt709hip_unconvert_batch(bt709hip_decoder *dec, int count, const void *const *ycbcr_words, size_t in_stride, int width, int height,
                             const bt709hip_surface *outs, void *stream, int wait_until_completed);

/* Pass 1 + pass 2 (MetalScaleRenderContext -renderScaled:, bilinear) fused for the
 * exact 2:1 ratio: out is (W/2) x (H/2).  Frame W,H must be multiples of 4.
 * Two-pass-equivalent arithmetic: each output channel is the linear-light mean of
 * the four decoded 8-bit sRGB values, re-encoded to sRGB ("rescale").
 * `alpha` / `alphas`: NULL unless the decoder has an alpha channel; the reference renders alpha
 * clips through the same two passes (AAPLShaders.metal:411-438 into the intermediate, then
 * MetalScaleRenderContext.m:55-105), where the alpha channel is stored and filtered as a plain
 * unorm: out alpha = round(255 * mean(byteNorm(decoded alpha bytes))). */
int bt709hip_decode_half(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                         const bt709hip_surface *out,
                         void *stream, int wait_until_completed);
int bt709hip_decode_half_batch(bt709hip_decoder *dec, int count,
                               const bt709hip_frame *frames, const bt709hip_frame *alphas,
                               const bt709hip_surface *outs,
                               void *stream, int wait_until_completed);

/* Pass 1 + pass 2 fused for ANY output size (view-fit): out->width x out->height need not be
 * related to the frame size (down- or up-scaling).  Bilinear in linear light over the decoded
 * 8-bit sRGB values, texel-centre sampling, clamp-to-edge; for an exact 2:1 ratio the result is
 * bit-identical to bt709hip_decode_half.  The reference leaves this arithmetic to the sampler
 * hardware (AAPLShaders.metal:73-85), so the definition is ours ("rescale").
 * The batch form takes `count` same-geometry frames and same-sized outputs in one launch
 * (grid.z = frame; same count limits as bt709hip_decode_batch): AAPLRenderer.m:970-976 calls
 * pass 2 once per frame, a 4K -> 1440p frame is a ~22 us kernel (15 us each with 8 per launch).
 * BT709HIP_ERR_UNSUPPORTED: a plane or the output of 2 GiB or more (row offsets are 32-bit), or
 * more than 65535 output rows. */
int bt709hip_decode_scaled(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                           const bt709hip_surface *out, void *stream, int wait_until_completed);
int bt709hip_decode_scaled_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames,
                                 const bt709hip_frame *alphas, const bt709hip_surface *outs, void *stream,
                                 int wait_until_completed);

/* Pass 2 on its own: -[MetalScaleRenderContext renderScaled:...] (MetalScaleRenderContext.h:34-40,
 * .m:55-105) + samplingShader (AAPLShaders.metal:73-85).  `in` is the intermediate pass 1 rendered
 * -- BGRA8_SRGB (each tap linearised as the sRGB8 sampler does) or RGBA16F (taps are linear light
 * already) -- `out` a BGRA8_SRGB surface of any size.  Bilinear, texel-centre sampling, clamp to edge,
 * weights and summation order as bt709hip_decode_scaled, so decode + render_scaled through a BGRA8
 * intermediate equals the fused call bit for bit.  The alpha channel is filtered as a plain unorm.
 * BT709HIP_ERR_UNSUPPORTED: a surface of 2 GiB or more. */
int bt709hip_render_scaled(bt709hip_context *ctx, const bt709hip_surface *in, const bt709hip_surface *out,
                           void *stream, int wait_until_completed);
/* The same pass over `count` intermediates of one geometry and format in ONE launch (grid.z = surface), for a caller that
 * rescales a ring of decoded frames: one 4K -> 1440p pass is a 3.7-Mpixel launch, too small to fill the chip.  The surfaces
 * must be evenly spaced in memory (surface i at surface 0 + i * (surface 1 - surface 0), as in a ring carved from one
 * allocation; BT709HIP_ERR_UNSUPPORTED otherwise), count <= 65535; differing sizes, strides or formats:
 * BT709HIP_ERR_SIZE_MISMATCH.  No reference twin (-renderScaled: takes one texture). */
int bt709hip_render_scaled_batch(bt709hip_context *ctx, int count, const bt709hip_surface *ins, const bt709hip_surface *outs,
                                 void *stream, int wait_until_completed);
/* The tables of bt709hip_render_scaled (and of RGBA16F decodes: bt709hip_decoder_prepare_format) are
 * built on first use; call these before bt709hip_graph_begin_capture.  Idempotent. */
int bt709hip_render_scaled_prepare(bt709hip_context *ctx);
int bt709hip_decoder_prepare_format(bt709hip_decoder *dec, int format);

/* --------------------------------------------------------------- frame pool */
/* Frames that live in HOST memory.  The reference hands the decoder CVPixelBuffers the GPU reads
 * in place (unified memory) and keeps MaxBuffersInFlight = 3 frames in flight behind a semaphore
 * (AAPLRenderer.m:34, 891-977); a discrete GPU needs the copies, so the pool owns, per in-flight
 * slot, one HIP stream, pinned host staging and device buffers:
 *   acquire  -> pointers to the slot's pinned Y / CbCr planes (waits for the slot's previous frame)
 *   submit   -> upload, decode, download enqueued on the slot's stream; returns at once
 *   wait     -> the slot's pinned BGRA rows, valid until the slot is acquired again
 * Slots are handed out round-robin, so `depth` frames overlap their copies and kernels.  The decoder
 * must outlive the pool; a pool is used from one thread at a time (decoders and contexts may be
 * shared between threads, each thread with pools / streams of its own).  For a decoder with an
 * alpha channel every slot also owns an alpha plane: fetch its pinned pointer with
 * bt709hip_pool_alpha_plane after acquire and fill it before submit. */
int bt709hip_pool_create(bt709hip_decoder *dec, int width, int height, int depth, bt709hip_pool **out);
int bt709hip_pool_destroy(bt709hip_pool *pool);
int bt709hip_pool_acquire(bt709hip_pool *pool, int *slot, void **y, size_t *y_stride, void **cbcr,
                          size_t *cbcr_stride);
int bt709hip_pool_alpha_plane(bt709hip_pool *pool, int slot, void **alpha, size_t *alpha_stride);
int bt709hip_pool_submit(bt709hip_pool *pool, int slot);
int bt709hip_pool_wait(bt709hip_pool *pool, int slot, const void **bgra, size_t *stride);
/* Hands an acquired slot back without submitting it (nothing is enqueued).  A FAILED bt709hip_pool_submit
 * hands its slot back by itself: a slot never stays "acquired" behind an error. */
int bt709hip_pool_release(bt709hip_pool *pool, int slot);

/* --------------------------------------------------------------- frame ring */
/* Frames that live in DEVICE memory: a ring of `frames` same-sized NV12 inputs carved from one slab and their outputs (BGRA8, or RGBA16F through
 * bt709hip_ring_options.format) from another (frame i at slab + i * spacing: any count goes out as one launch, bt709hip_decode_batch's "evenly spaced"
 * form) -- what a streaming application keeps resident, and what bench.py times.  The reference's twin is the set of
 * CVPixelBuffers + the render texture it keeps per in-flight frame (AAPLRenderer.m:34, 530-862); unified memory has no
 * placement to choose, a discrete HBM device does: where the two slabs land decides how fast the launch streams (the same
 * 256-frame 4K launch runs at 0.74-0.82 of the HBM roofline on allocations made one after the other by one process, each
 * keeping its rate).  bt709hip_ring_create therefore allocates up to `tries` candidates per slab (0 = the
 * default, 6; 1 = first allocation, no probing; rings under 256 MB never probe), times the DECODER'S OWN LAUNCH over the ring
 * on the pairings (~15 ms each; every output candidate under input 0 first -- twice or three times `tries` of them when they
 * all look alike -- then every input with the `tries` fastest outputs, then the three best pairings and the first-allocated
 * one again, six times as long), keeps the fastest pairing and frees the rest.  TRANSIENT FOOTPRINT: every candidate is device
 * memory held until the choice is made -- unbounded that is up to 6 input + 18 output slabs (172 GB for a 256-frame 4K ring) --
 * so the hunt runs under a BUDGET (bt709hip_ring_options): by default it never holds more than FOUR TIMES the ring (47 GB for that
 * ring; measured to choose as well as holding everything, profiles/r05_hunt_budget.txt) nor more than half of the memory that was
 * free at the call, ring included, and always leaves 4 GiB of the device free; when a new candidate does not fit, the slowest slab
 * seen so far is freed first (the fastest ones stay for the pairing probes), down to the incumbent pair + one candidate.  The
 * duration and the peak footprint are reported (hunt_ms, peak_bytes).  Set-up cost: 1-2 s for a 12 GB ring.  half_scale != 0: outputs are (W/2) x (H/2) and the ring decodes
 * through bt709hip_decode_half_batch.  A decoder with an alpha channel gets an alpha plane per frame (third plane of the
 * input slab).  The memory is NOT cleared.  The decoder must outlive the ring. */
typedef struct bt709hip_ring bt709hip_ring;
typedef struct {
  int32_t tries;                     /* candidates per slab asked for (after clamping) */
  int32_t in_candidates;             /* input slabs allocated */
  int32_t out_candidates;            /* output slabs allocated (up to 3 x tries) */
  int32_t chosen_in, chosen_out;     /* allocation-order index of the slabs kept */
  int32_t probes;                    /* pairings probed */
  float first_GBps;                  /* probe of the first-allocated pairing (input 0, output 0): what tries = 1 keeps */
  float chosen_GBps;                 /* the pairing kept, on the longer confirming probe */
  float best_GBps, worst_GBps;       /* over the pairing probes */
  float out_prescan_GBps[18];        /* output candidates under input 0, allocation order; 0 = none */
  int32_t out_kept[18];              /* allocation-order indices of the outputs that went on to the pairing probes; -1 = none */
  float hunt_ms;                     /* wall-clock time of the whole hunt (allocations, probes, frees); 0 without a hunt */
  int32_t stopped_by;                /* 0: ran to its end; 1: the byte budget cut candidates; 2: the time budget ended it early */
  uint64_t peak_bytes;               /* most device memory this call held at once, the ring's own two slabs included */
  uint64_t budget_bytes;             /* the byte budget it ran under (after defaults and clamping) */
  int32_t evicted;                   /* candidate slabs freed before the choice to make room (0 when the budget held them all) */
  int32_t reserved;
} bt709hip_ring_placement;
/* Budget of the placement hunt.  A zeroed struct (or NULL) = the defaults. */
typedef struct {
  uint64_t max_bytes;   /* most device memory the call may hold at once, the ring's two slabs included; 0 = four times the ring,
                           at most half of the memory free at the call.  A budget that cannot hold the ring plus one more slab leaves nothing to compare: the ring
                           is then allocated without a hunt */
  uint32_t max_ms;      /* wall-clock budget of the hunt in milliseconds (checked before every probe); 0 = none */
  int32_t frugal;       /* != 0: max_bytes = the ring + ONE candidate pair, whatever is free: the incumbent pair and the pair being
                           probed are all that ever lives */
  int32_t format;       /* render target of the ring's outputs: BT709HIP_FORMAT_BGRA8_SRGB (0, the default) or BT709HIP_FORMAT_RGBA16F
                           (8 bytes per pixel, linear-light halves: the reference's fallback intermediate, AAPLRenderer.m:143-170; not
                           with half_scale).  The hunt then probes with THAT launch: the RGBA16F kernel's 84 %-written stream lands
                           in the slow or the fast regime by the same lottery (0.70 against 0.77) (ABI 502) */
  int32_t reserved;
} bt709hip_ring_options;
int bt709hip_ring_create(bt709hip_decoder *dec, int width, int height, int frames, int half_scale, int tries, bt709hip_ring **out);
/* the same with an explicit budget (bt709hip_ring_create = options NULL) */
int bt709hip_ring_create_ex(bt709hip_decoder *dec, int width, int height, int frames, int half_scale, int tries,
                            const bt709hip_ring_options *options, bt709hip_ring **out);
int bt709hip_ring_destroy(bt709hip_ring *ring);
int bt709hip_ring_frames(const bt709hip_ring *ring);
/* Descriptors of frame `index` (any of the three pointers may be NULL; alpha is zeroed for an opaque decoder). */
int bt709hip_ring_frame(const bt709hip_ring *ring, int index, bt709hip_frame *frame, bt709hip_frame *alpha, bt709hip_surface *out);
int bt709hip_ring_placement_info(const bt709hip_ring *ring, bt709hip_ring_placement *info);
/* Frames [first, first + count) in ONE launch on `stream`. */
int bt709hip_ring_decode(bt709hip_ring *ring, int first, int count, void *stream, int wait_until_completed);

/* ----------------------------------------------------------------- ring set */
/* ONE process, SEVERAL GPUs, frames resident in DEVICE memory: a bt709hip_ring per lane, each with a context and a decoder of
 * its own on device_ordinals[lane] (ordinals may repeat), driven by ONE thread -- the reference's shape, one process that drives
 * everything (Renderer/AAPLRenderer.m:874-985), widened to the GPUs of a node.  bt709hip_ringset_decode issues ONE ring launch
 * per lane, in lane order, on each lane's default stream and returns (the launches run concurrently, one per device; a launch
 * call costs ~10 us of host time against ~1.8 ms of kernel for a 256-frame 4K ring); _synchronize waits for every lane.  No
 * collective, no peer access: nothing crosses GPUs.  The host-frame counterpart is the frame sharder below.  Each lane's ring
 * is created like bt709hip_ring_create_ex's (placement hunt per device, same budget semantics, per device).  Fill the rings
 * through bt709hip_ringset_lane_ring + bt709hip_ring_frame + bt709hip_upload on bt709hip_ringset_lane_context. */
typedef struct bt709hip_ringset bt709hip_ringset;
int bt709hip_ringset_create(const int *device_ordinals, int lanes, int gamma, int has_alpha, int width, int height, int frames,
                            int half_scale, int tries, const bt709hip_ring_options *options, bt709hip_ringset **out);
int bt709hip_ringset_destroy(bt709hip_ringset *set);
int bt709hip_ringset_lanes(const bt709hip_ringset *set);
bt709hip_context *bt709hip_ringset_lane_context(bt709hip_ringset *set, int lane);
bt709hip_decoder *bt709hip_ringset_lane_decoder(bt709hip_ringset *set, int lane);
bt709hip_ring *bt709hip_ringset_lane_ring(bt709hip_ringset *set, int lane);
/* frames [first, first + count) of EVERY lane's ring: one launch per lane, issued from the calling thread */
int bt709hip_ringset_decode(bt709hip_ringset *set, int first, int count, int wait_until_completed);
int bt709hip_ringset_synchronize(bt709hip_ringset *set);

/* ------------------------------------------------------------ frame sharder */
/* ONE process driving SEVERAL GPUs: independent frames shard with no exchange step, frame i (in submission
 * order) goes to lane i mod n.  The reference is one process with one device, one queue and N frames in
 * flight (Renderer/MetalRenderContext.m:59-63, Renderer/AAPLRenderer.m:34, 874-985); its counterpart on an
 * 8-GPU node is this dispatcher over n lanes, each lane = its own bt709hip_context + decoder + in-flight pool
 * of `depth` slots (one HIP stream, pinned staging and device buffers per slot) on device_ordinals[lane].
 * Ordinals may repeat (several lanes on one GPU).  No collective, no peer access, nothing crosses GPUs.
 *   acquire -> ticket + pinned Y / CbCr (/ alpha) planes of the next frame's slot (waits for that slot's previous frame)
 *   commit  -> upload, decode, download enqueued on the slot's stream of the ticket's lane; returns at once
 *   submit  -> acquire + copy of caller-owned host planes (any pitch; tags validated as -decodeBT709: does) + commit
 *   wait    -> the frame's pinned BGRA rows, valid until its slot is handed out again: lanes * depth frames later (sooner if
 *              frames of its lane were cancelled or failed in between); BT709HIP_ERR_INVALID_ARG once recycled
 *   cancel  -> hands an acquired, uncommitted ticket back
 * Threading: a shard is driven by ONE thread at a time (like a pool); that thread only enqueues, the lanes'
 * streams run concurrently.  Several feeding threads use a shard each (contexts are per shard).
 * Frames that already live in device memory: bt709hip_ringset_* above. */
typedef struct bt709hip_shard bt709hip_shard;
int bt709hip_shard_create(const int *device_ordinals, int lanes, int gamma, int has_alpha, int width, int height, int depth,
                          bt709hip_shard **out);
int bt709hip_shard_destroy(bt709hip_shard *shard);
int bt709hip_shard_lanes(const bt709hip_shard *shard);
int bt709hip_shard_lane_device(const bt709hip_shard *shard, int lane);
bt709hip_decoder *bt709hip_shard_lane_decoder(bt709hip_shard *shard, int lane); /* for set_option / set_alpha_fill */
int bt709hip_shard_acquire(bt709hip_shard *shard, uint64_t *ticket, void **y, size_t *y_stride, void **cbcr, size_t *cbcr_stride,
                           void **alpha, size_t *alpha_stride);
int bt709hip_shard_commit(bt709hip_shard *shard, uint64_t ticket);
int bt709hip_shard_cancel(bt709hip_shard *shard);
int bt709hip_shard_submit(bt709hip_shard *shard, const bt709hip_frame *host_frame, const bt709hip_frame *host_alpha,
                          uint64_t *ticket);
int bt709hip_shard_wait(bt709hip_shard *shard, uint64_t ticket, const void **bgra, size_t *stride);

/* ------------------------------------------------------------------ encoder */
/* The step before the decode path, on the GPU: 8-bit BGRA -> NV12 BT.709 video range with
 * the reference's linear-light 2x2 chroma averaging.  Replaces
 * +[BGRAToBT709Converter convertIntoCoreVideoBuffer:cvPixelBuffer:inputGamma:outputGamma:]
 * (Renderer/BGRAToBT709Converter.h:73-76, .m:532-569) -> cvpbu_ycbcr_subsample
 * (Renderer/CVPixelBufferUtils.h:241-399) -> BT709_average_pixel_values
 * (Renderer/BT709.h:1349-1509).  input_gamma / output_gamma are bt709hip_gamma values
 * APPLE, SRGB or LINEAR (BT709Gamma, BT709.h:20-25); the app encodes with (SRGB, APPLE),
 * (SRGB, SRGB) or (LINEAR, LINEAR) (BGRAToBT709Converter.m:919-935).
 * `in` is read (alpha ignored), the planes `out` points to are written; out->matrix and
 * out->transfer are ignored.  Width and height must be even and equal on both sides. */
int bt709hip_encode(bt709hip_context *ctx, const bt709hip_surface *in, const bt709hip_frame *out,
                    int input_gamma, int output_gamma, void *stream, int wait_until_completed);

/* The encoder's lookup tables for one (input_gamma, output_gamma) pair are built on first use
 * (device allocation + blocking copies).  That is not allowed while a stream records a graph:
 * call this once per pair before bt709hip_graph_begin_capture (an encode that finds its tables
 * missing during a capture returns BT709HIP_ERR_NOT_SETUP).  Idempotent. */
int bt709hip_encoder_prepare(bt709hip_context *ctx, int input_gamma, int output_gamma);

/* `count` same-sized pictures in ONE launch (the app encodes its frames one call at a time,
 * BGRAToBT709Converter.m:532-569; a 4K frame is a ~12 us kernel, too short to fill the chip).
 * Same limits as bt709hip_decode_batch: up to BT709HIP_MAX_BATCH arbitrary buffers, or any
 * number up to 65535 when picture i sits at picture 0 + i * (picture 1 - picture 0) in all
 * three planes.  All pictures share size and strides (else BT709HIP_ERR_SIZE_MISMATCH). */
int bt709hip_encode_batch(bt709hip_context *ctx, int count, const bt709hip_surface *ins,
                          const bt709hip_frame *outs, int input_gamma, int output_gamma, void *stream,
                          int wait_until_completed);

/* ------------------------------------------------------------ plane layouts */
/* The reference's on-disk 4:2:0 format is YUV4MPEG2 "C420jpeg": planar Y, then U (Cb), then V
 * (Cr) per frame (Renderer/y4m_writer.h:194-241).  These move the two chroma planes to / from
 * NV12's interleaved CbCr plane on the device (the Y plane is identical in both layouts).
 * chroma_width x chroma_height = (W/2) x (H/2); strides in bytes. */
int bt709hip_interleave_cbcr(bt709hip_context *ctx, const void *u, size_t u_stride, const void *v, size_t v_stride,
                             void *cbcr, size_t cbcr_stride, int chroma_width, int chroma_height,
                             void *stream, int wait_until_completed);
int bt709hip_deinterleave_cbcr(bt709hip_context *ctx, const void *cbcr, size_t cbcr_stride, void *u, size_t u_stride,
                               void *v, size_t v_stride, int chroma_width, int chroma_height,
                               void *stream, int wait_until_completed);

/* -------------------------------------------------------------- diagnostics */
/* Streaming copy of `bytes` (a multiple of 16, both pointers 16-byte aligned) with 16-byte
 * non-temporal loads and stores, one launch: the bandwidth a plain copy reaches on this device, for
 * benchmarks that want to report a kernel against the same box's copy rate (no reference twin). */
int bt709hip_copy_probe(bt709hip_context *ctx, void *dst, const void *src, size_t bytes, void *stream);
/* Placement-aware allocation of ONE streaming slab.  Takes up to `tries` candidates of `bytes` ONE AT A TIME against the incumbent
 * (at most two slabs are alive at any moment; freeing and allocating again hands out other physical pages, so holding them all
 * buys nothing), times a streaming copy (lower half onto upper half) plus a fill of the whole slab over each and keeps the fastest;
 * the slab kept has been overwritten by the probe (zero-filled) whenever a probe ran, and is NOT cleared otherwise (tries = 1, a slab
 * under 2 MiB).  rates_GBps (optional, `tries` floats; always fully written: 0 where no probe ran) receives the probe rates,
 * *chosen (optional) the index kept.  tries = 1 is bt709hip_malloc.
 * This is the WEAKER, cheaper probe: it ranks a slab by itself, with a generic kernel.  A frame ring should use
 * bt709hip_ring_create, which probes with the decoder's own launch and chooses the input x output PAIRING (worth a further
 * 1-2 %, profiles/r03_placement_cross.txt).  No reference twin (unified memory has no placement to choose). */
int bt709hip_malloc_streaming(bt709hip_context *ctx, size_t bytes, int tries, void **dptr, float *rates_GBps, int *chosen);

const char *bt709hip_strerror(int status);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int bt709hip_last_hip_error(void);
const char *bt709hip_last_hip_error_string(void);

/* Introspection used by the parity tests (host memory out).
 * thresholds: 255 floats, t[k-1] = smallest x in [0,1] whose output byte is >= k.
 * constants:  8 floats {1/255, M_y, M_cr_r, M_cb_g, M_cr_g, M_cb_b, 16, 128}
 *             (matrix built as BT709.h:386-397). */
int bt709hip_gamma_thresholds(int gamma, float thresholds[255]);
/* The kernels' lookup of one saturated channel value x in [0,1], replayed on the host from the
 * host-built bucket table (same index function, same compare): the byte the GPU would produce, and
 * optionally the bucket count N and the bucket index.  Returns the byte, or <0 on a bad argument. */
int bt709hip_gamma_lookup(int gamma, float x, int *bucket_count, int *bucket_index);
/* The same for the table the 1:1 kernels (decode, +unconvert:) stage: where the uniform table is large because the thresholds
 * crowd near zero (the LINEAR mode: 4 096 buckets) they use a LOG-bucket form of it, bucket = (bits(x + a) >> 16) - first, a = 2^-5
 * (645 buckets); for the other modes this is bt709hip_gamma_lookup.  *log_form (optional) = 1 / 0. */
int bt709hip_gamma_lookup_decode(int gamma, float x, int *bucket_count, int *bucket_index, int *log_form);
int bt709hip_matrix_constants(float constants[8]);
/* RGBA16F targets: the threshold table of the half-float composite H(x) = half(curve_to_linear(x))
 * (host memory out).  thresholds: up to `capacity` floats, T[i] = smallest x with H(x) >= first_code + i;
 * returns the number of entries the table has (0: the gamma has no curve), or <0. */
int bt709hip_half_thresholds(int gamma, float *thresholds, int capacity);
/* The kernels' settlement of one saturated x replayed on the host: the candidate is the exact code
 * H(x) moved by candidate_offset (0 or -1: the kernel's candidate, one fma over a tangent of the curve, lies below the true value and lands on
 * H or H - 1), then corrected against the one threshold above it.  Returns the half code;
 * *table_entries (optional) as above. */
int bt709hip_half_lookup(int gamma, float x, int candidate_offset, int *table_entries);
/* Name of the kernel the last decode on this thread launched (for profiling). */
const char *bt709hip_last_kernel_name(void);
/* Launch shape of the last bt709hip_decode / _decode_batch (1:1, BGRA8 or RGBA16F target) this thread issued: grid and block of its
 * first kernel launch, the number of launches it took (2: the XCD-aware map over a multiple of 8 frames plus the plain map
 * over the rest) and the work map of the first (bt709hip_decoder_option BT709HIP_OPT_XCD_BANDS value actually used; 0 plain). */
typedef struct {
  uint32_t grid[3], block[3];
  int32_t launches, xcd_bands;
} bt709hip_launch_info;
int bt709hip_last_launch_info(bt709hip_launch_info *info);

#ifdef __cplusplus
}
#endif
#endif /* BT709HIP_H */
