/*
 * bt709hip_ext.h -- the part of the C ABI of libbt709hip.so that has NO twin in the reference: what a discrete HBM
 * device needs around the decode operator (frames resident in a ring whose placement is hunted for, rings on several
 * GPUs, frames fed from host memory through an in-flight pool or a multi-GPU sharder), what keeps the reference's
 * one-call-per-frame cadence fast (coalescing submit, graphs), batched forms of the side paths, tuning options,
 * timing events and the introspection calls the parity tests use.  bt709hip.h holds the reference-twinned calls; both
 * headers describe ONE library and ONE ABI number (BT709HIP_VERSION).
 */
#ifndef BT709HIP_EXT_H
#define BT709HIP_EXT_H

#include "bt709hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bt709hip_pool bt709hip_pool; /* ~ CVPixelBufferPool + texture cache + in-flight semaphore */

/* ------------------------------------------------------------ device, options */
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
int bt709hip_context_info(const bt709hip_context *ctx, bt709hip_device_info *info);

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

/* A stream with a scheduling priority (hipStreamCreateWithPriority): 0 = normal, negative = higher, positive = lower; clamped
 * to the device's range.  (MTLCommandQueue has no twin; a renderer that decodes ahead of what it presents puts the look-ahead
 * frames on a lower-priority stream.) */
int bt709hip_stream_create_with_priority(bt709hip_context *ctx, int priority, void **stream);

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

/* Free and total device memory of the context's GPU right now (hipMemGetInfo), for callers that size rings or a placement
 * hunt against what is left.  Either pointer may be NULL. */
int bt709hip_mem_info(bt709hip_context *ctx, size_t *free_bytes, size_t *total_bytes);
/* Pinned host memory: uploads from / downloads into it are asynchronous (bt709hip.h, "Device memory"). */
int bt709hip_host_alloc(bt709hip_context *ctx, size_t bytes, void **hptr);
int bt709hip_host_free(bt709hip_context *ctx, void *hptr);

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
 * issued the queue.  Thread safety: as without the option (several threads may share a decoder; each queue is per stream); while any decoder of a
 * context coalesces, that context's stream-taking calls serialise on one mutex for the length of the launch calls they issue, and a
 * call made while its stream records a graph issues that stream's queue only (aged queues of other streams wait for the next call). */
int bt709hip_decoder_flush(bt709hip_decoder *dec, void *stream /* NULL = the context's default stream */);
/* every stream's queue of this decoder */
int bt709hip_decoder_flush_all(bt709hip_decoder *dec);

/* ------------------------------------------------------------- batched forms */
/* `count` frames / pictures / surfaces of one geometry in ONE launch (grid.z = item), like bt709hip_decode_batch and with
 * its limits: up to BT709HIP_MAX_BATCH arbitrary buffers, or any number up to 65535 when item i sits at item 0 + i * (item 1
 * - item 0) in every plane.  No reference twin (the reference converts, rescales and encodes one frame per call: a 4K frame
 * is a 12-22 us kernel, too short to fill the chip; one 4K +unconvert: per call runs at 0.58 of the roofline).  Differing sizes,
 * strides or formats: BT709HIP_ERR_SIZE_MISMATCH.  bt709hip_render_scaled_batch takes evenly spaced surfaces only
 * (BT709HIP_ERR_UNSUPPORTED otherwise). */
int bt709hip_unconvert_batch(bt709hip_decoder *dec, int count, const void *const *ycbcr_words, size_t in_stride, int width, int height,
                             const bt709hip_surface *outs, void *stream, int wait_until_completed);
int bt709hip_decode_half_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                               const bt709hip_surface *outs, void *stream, int wait_until_completed);
int bt709hip_decode_scaled_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                                 const bt709hip_surface *outs, void *stream, int wait_until_completed);
int bt709hip_render_scaled_batch(bt709hip_context *ctx, int count, const bt709hip_surface *ins, const bt709hip_surface *outs,
                                 void *stream, int wait_until_completed);
int bt709hip_encode_batch(bt709hip_context *ctx, int count, const bt709hip_surface *ins, const bt709hip_frame *outs, int input_gamma,
                          int output_gamma, void *stream, int wait_until_completed);
/* Lookup tables that are built on first use (device allocation + blocking copies) cannot be built while a stream records a
 * graph: call these before bt709hip_graph_begin_capture (a call that finds its tables missing during a capture returns
 * BT709HIP_ERR_NOT_SETUP).  Idempotent.  render_scaled_prepare: pass 2 alone; decoder_prepare_format: RGBA16F targets;
 * encoder_prepare: one (input_gamma, output_gamma) pair of the encoder. */
int bt709hip_render_scaled_prepare(bt709hip_context *ctx);
int bt709hip_decoder_prepare_format(bt709hip_decoder *dec, int format);
int bt709hip_encoder_prepare(bt709hip_context *ctx, int input_gamma, int output_gamma);

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
/* Frames that live in DEVICE memory: a ring of `frames` same-sized NV12 inputs carved from one slab and their outputs (BGRA8, or
 * RGBA16F through bt709hip_ring_options.format) from another (frame i at slab + i * spacing: any count goes out as one launch,
 * bt709hip_decode_batch's "evenly spaced" form) -- what a streaming application keeps resident, and what bench.py times.  The
 * reference's twin is the set of CVPixelBuffers + the render texture it keeps per in-flight frame (AAPLRenderer.m:34, 530-862);
 * unified memory has no placement to choose, a discrete HBM device does: where the two slabs land decides how fast the launch
 * streams (the same 256-frame 4K launch runs at 0.74-0.82 of the HBM roofline on allocations made one after the other by one
 * process, each keeping its rate).  bt709hip_ring_create therefore allocates candidates -- `tries` sets how many
 * (0 = the default, 6; 1 = first allocation, no probing; rings under 256 MB never probe) -- times the DECODER'S OWN LAUNCH over
 * the ring on them (~15 ms each: every output candidate under the first input, 2-3 x `tries` of them; then the pairings with the
 * other inputs; then the best pairings and the first-allocated one again, six times as long), keeps the fastest pairing and frees
 * the rest.  TRANSIENT FOOTPRINT: the hunt runs under a BUDGET (bt709hip_ring_options).  By default (round 6) it is FRUGAL: it
 * never holds more than TWICE the ring -- the incumbent pair + one candidate pair; a new candidate evicts the slower of the two
 * outputs alive -- because freeing and allocating again hands out other physical pages, so holding every candidate buys nothing
 * (the 11.7 GB 4K ring: 0.80-0.82 of the roofline within 23 GB in ~1 s, against 0.81 within 58-148 GB in 2-5 s;
 * profiles/r05_hunt_budget.txt, profiles/r06_hunt_default.txt), and always leaves 4 GiB of the device free.  max_bytes names a
 * wider budget.  Duration and peak footprint are reported (hunt_ms, peak_bytes).  half_scale != 0: outputs are (W/2) x (H/2) and
 * the ring decodes through bt709hip_decode_half_batch.  A decoder with an alpha channel gets an alpha plane per frame (third
 * plane of the input slab).  The memory is NOT cleared.  The decoder must outlive the ring. */
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
  uint64_t max_bytes;   /* most device memory the call may hold at once, the ring's two slabs included; 0 = the default: twice the
                           ring (frugal).  A budget that cannot hold the ring plus one more output slab leaves nothing to compare:
                           the ring is then allocated without a hunt */
  uint32_t max_ms;      /* wall-clock budget of the hunt in milliseconds (checked before every probe); 0 = none */
  int32_t frugal;       /* != 0: twice the ring whatever max_bytes says, one input candidate: the incumbent pair and the pair being
                           probed are all that ever lives (what max_bytes = 0 selects since round 6) */
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

/* -------------------------------------------------------------- diagnostics */
/* Streaming copy of `bytes` (a multiple of 16, both pointers 16-byte aligned) with 16-byte
 * non-temporal loads and stores, one launch: the bandwidth a plain copy reaches on this device, for
 * benchmarks that want to report a kernel against the same box's copy rate (no reference twin). */
int bt709hip_copy_probe(bt709hip_context *ctx, void *dst, const void *src, size_t bytes, void *stream);
/* Placement-aware allocation of ONE streaming slab.  Takes up to `tries` candidates of `bytes` ONE AT A TIME against the incumbent
 * (two slabs are alive while one is probed, three for the moment of an allocation: the slab that lost stays until the next
 * candidate has its memory, so the allocator cannot hand the same block straight back; holding them all buys nothing), times a streaming copy (lower half onto upper half) plus a fill of the whole slab over each and keeps the fastest;
 * the slab kept has been overwritten by the probe (zero-filled) whenever a probe ran, and is NOT cleared otherwise (tries = 1, a slab
 * under 2 MiB).  rates_GBps (optional, `tries` floats; always fully written: 0 where no probe ran) receives the probe rates,
 * *chosen (optional) the index kept.  tries = 1 is bt709hip_malloc.
 * This is the WEAKER, cheaper probe: it ranks a slab by itself, with a generic kernel.  A frame ring should use
 * bt709hip_ring_create, which probes with the decoder's own launch and chooses the input x output PAIRING (worth a further
 * 1-2 %, profiles/r03_placement_cross.txt).  No reference twin (unified memory has no placement to choose). */
int bt709hip_malloc_streaming(bt709hip_context *ctx, size_t bytes, int tries, void **dptr, float *rates_GBps, int *chosen);

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
#endif /* BT709HIP_EXT_H */
