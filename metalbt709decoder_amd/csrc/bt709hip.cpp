// C-ABI shim (include/bt709hip.h): the thin layer that replaces MetalRenderContext /
// MetalBT709Decoder's CoreVideo + Metal plumbing with hipMalloc / hipMemcpy2DAsync /
// HIP streams, and validates a decode call the way -processBT709ToSRGB: does
// (Renderer/MetalBT709Decoder.m:252-492) before launching the fused kernel.
//
// There is no CPU fallback anywhere in this file: without a HIP device every entry
// point that needs one fails with BT709HIP_ERR_NO_DEVICE / BT709HIP_ERR_HIP.
#include "../../include/bt709hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include <mutex>
#include <new>

#include "bt709_constants.h"
#include "bt709_kernels.h"
#include "transfer_tables.h"

using namespace bt709;

struct EncoderTables {  // device copies for one (input gamma, output gamma) pair
  void *d_per_byte = nullptr;
  void *d_from_linear = nullptr;
  uint32_t from_linear_bytes = 0;
  uint32_t from_linear_n = 0;
  float split = 0.0f, coarse_scale = 1.0f;
  uint32_t coarse_offset = 0;
};

struct bt709hip_context {
  int device = 0;
  hipDeviceProp_t props;
  hipStream_t default_stream = nullptr;
  int grid_mult = 2;    // BT709HIP_CTX_OPT_GRID_MULT
  int grid_blocks = 0;  // workgroups a general-path launch aims for (all frames together)
  int encode_row_pairs = 0, encode_threads = 0;  // BT709HIP_CTX_OPT_ENCODE_*: 0 = sized per launch
  int xcd_bands = 1;                             // BT709HIP_CTX_OPT_XCD_BANDS: XCD-aware work map of batched encoder launches
  int streaming_tries = 4;                       // BT709HIP_CTX_OPT_STREAMING_TRIES: placement candidates for buffers of 256 MB and more that the library allocates itself
  std::mutex encoder_mutex;
  EncoderTables encoders[3][3];  // [input gamma][output gamma], built on first use
  // bt709hip_render_scaled (pass 2 alone): built on first use under encoder_mutex
  void *d_render_encode = nullptr, *d_render_lin = nullptr;
  uint32_t render_encode_bytes = 0, render_encode_log_first = 0;
  float render_encode_log_add = 0.0f;
  // decoders of this context with BT709HIP_OPT_COALESCE on: every entry point that takes a stream issues their queued
  // frames for that stream first (flush_stream), so the stream keeps its order
  std::atomic<int> n_coalescing{0};
  std::mutex coalescing_mutex;
  std::vector<bt709hip_decoder *> coalescing;
};

// BT709HIP_OPT_COALESCE: frames validated and queued for one stream, not yet launched (include/bt709hip.h, COALESCING SUBMIT)
struct PendingQueue {
  hipStream_t stream = nullptr;
  bool with_alphas = false;  // the calls passed alpha descriptors
  int64_t oldest_us = 0;     // steady-clock time at which the oldest frame it holds was queued (BT709HIP_OPT_COALESCE_MAX_AGE_US)
  std::vector<bt709hip_frame> frames, alphas;
  std::vector<bt709hip_surface> outs;
};

struct bt709hip_decoder {
  bt709hip_context *ctx = nullptr;
  int gamma = BT709HIP_GAMMA_APPLE;
  int has_alpha = 0;
  // Options: atomics, because a thread may change one while others are inside a decode (tests/native/shim_stress.cpp does, under
  // TSan); a call reads each option once and runs with what it read.
  std::atomic<uint32_t> alpha_fill{0xFF};
  std::atomic<bool> nontemporal{true};  // BT709HIP_OPT_NONTEMPORAL
  std::atomic<int> half_rep{-1};        // BT709HIP_OPT_HALF_KERNEL: persistent conflict-free rescale kernel: -1 = when the launch is large enough, 0 never, 1 always
  std::atomic<int> half_workgroups{0};  // BT709HIP_OPT_HALF_WORKGROUPS: 0 = one per compute unit
  std::atomic<int> half_lds_kb{0};      // BT709HIP_OPT_HALF_LDS_KB: 0 = all 160
  std::atomic<int> xcd_bands{1};        // BT709HIP_OPT_XCD_BANDS: XCD-aware work map of the batched 1:1 kernels (frames a multiple of 8)
  std::atomic<int> coalesce{0};         // BT709HIP_OPT_COALESCE: 0 off, else frames gathered per launch (2..32)
  std::atomic<int> coalesce_max_age_us{0};  // BT709HIP_OPT_COALESCE_MAX_AGE_US: 0 = no age limit
  std::mutex queue_mutex;   // guards queues
  std::vector<PendingQueue> queues;  // one per stream that has (had) queued frames
  std::mutex setup_mutex;
  std::atomic<bool> ready{false};  // release-stored after the tables below are published, acquire-loaded by every decode
  // device copies (transfer_tables.h)
  uint32_t table_n = 0;            // bucket count N of the decoder's gamma
  float unit1_magic = 0.0f;        // index function of d_table_unit (DecodeParams::unit1_*): uniform or log-bucket form
  uint32_t unit1_first = 0, unit1_shift = 0;
  void *d_table_unit = nullptr;    // TransferBucket[N + 1] (decode kernels)
  uint32_t table_unit_bytes = 0;
  void *d_table_linear = nullptr;  // TransferBucketLinear[N + 1] (rescale kernels, decode side)
  uint32_t table_linear_bytes = 0;
  void *d_encode = nullptr;        // the sRGB-encode composite's log-bucket TransferBucket[] (rescale kernels, encode side)
  uint32_t encode_bytes = 0;
  float encode_log_add = 0.0f;
  uint32_t encode_log_first = 0;
  void *d_encode_u = nullptr;      // the same composite as a UniformTable (persistent 2:1 kernel)
  uint32_t encode_u_bytes = 0, encode_u_n = 0;
  // RGBA16F targets: threshold table of the half-float composite (transfer_tables.h HalfTable), built on
  // first use under setup_mutex; half.table_bytes == 0 with half_ready: the gamma has no curve
  bool half_ready = false;  // under setup_mutex
  HalfParams half = {};
};

struct bt709hip_pool {
  struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *h_in = nullptr, *h_out = nullptr;  // pinned; h_in = Y, CbCr (and the alpha plane behind them)
    uint8_t *d_in = nullptr, *d_out = nullptr;
    bool busy = false;       // submitted, not yet waited for
    bool acquired = false;   // handed out, not yet submitted
  };
  bt709hip_decoder *dec = nullptr;
  int width = 0, height = 0;
  size_t in_bytes = 0, out_bytes = 0;
  std::vector<Slot> slots;
  size_t next = 0;
};

namespace {

thread_local hipError_t tl_hip_error = hipSuccess;
thread_local const char *tl_kernel_name = "";

int hip_fail(hipError_t e) {
  tl_hip_error = e;
  // The runtime keeps the error as this thread's "last error" until somebody reads it, and every launch here ends in
  // hipGetLastError(): without this, a failed allocation (reported to its caller, as it should be) would also fail the NEXT
  // decode of the thread with a stale out-of-memory (round 4: found by a test that asks for a ring the device cannot hold).
  (void)hipGetLastError();
  return BT709HIP_ERR_HIP;
}

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return hip_fail(_e); \
  } while (0)

int bind(const bt709hip_context *ctx) {
  if (ctx == nullptr) return BT709HIP_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  return BT709HIP_OK;
}

hipStream_t pick(const bt709hip_context *ctx, void *stream) {
  return stream ? static_cast<hipStream_t>(stream) : ctx->default_stream;
}

bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

int clamp_int(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// gridDim.y / .z limit of a HIP launch: the row-pair dimension of every kernel lives there
constexpr int kMaxGridYZ = 65535;

// transfer tag the configured gamma insists on (MetalBT709Decoder.m:335-353)
int required_transfer(int gamma) {
  switch (gamma) {
    case BT709HIP_GAMMA_SRGB: return BT709HIP_TRANSFER_SRGB;
    case BT709HIP_GAMMA_LINEAR: return BT709HIP_TRANSFER_LINEAR;
    default: return BT709HIP_TRANSFER_ITU_R_709_2;  // APPLE and the ITU709 extension
  }
}

// Validation of one frame/alpha/surface triple in the reference's order
// (MetalBT709Decoder.m:265-368), then the checks the texture wrappers imply.
int validate(const bt709hip_decoder *dec, const bt709hip_frame *f, const bt709hip_frame *a,
             const bt709hip_surface *o, int out_w, int out_h, int render_w, int render_h) {
  if (f == nullptr || o == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (f->width < 0 || f->height < 0) return BT709HIP_ERR_INVALID_ARG;
  if (o->width != out_w || o->height != out_h) return BT709HIP_ERR_SIZE_MISMATCH;          // .m:272-282
  if (render_w != out_w || render_h != out_h) return BT709HIP_ERR_SIZE_MISMATCH;            // .m:284-290
  if (a != nullptr && (a->width != f->width || a->height != f->height)) return BT709HIP_ERR_SIZE_MISMATCH;  // .m:294-306
  if (f->matrix != BT709HIP_MATRIX_ITU_R_709_2) return BT709HIP_ERR_MATRIX;                // .m:311-318
  if (f->transfer != required_transfer(dec->gamma)) return BT709HIP_ERR_TRANSFER;          // .m:320-353
  if (a != nullptr && a->transfer != BT709HIP_TRANSFER_LINEAR) return BT709HIP_ERR_ALPHA_TRANSFER;  // .m:357-368
  if ((f->width & 1) || (f->height & 1)) return BT709HIP_ERR_ODD_DIMENSIONS;
  if (dec->has_alpha && a == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (f->width == 0 || f->height == 0) return BT709HIP_OK;
  if (f->y == nullptr || f->cbcr == nullptr || o->bgra == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dec->has_alpha && a->y == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (f->y_stride < static_cast<size_t>(f->width) || f->cbcr_stride < static_cast<size_t>(f->width))
    return BT709HIP_ERR_STRIDE;
  if (dec->has_alpha && a->y_stride < static_cast<size_t>(a->width)) return BT709HIP_ERR_STRIDE;
  if ((o->format != BT709HIP_FORMAT_BGRA8_SRGB && o->format != BT709HIP_FORMAT_RGBA16F) || o->reserved != 0)
    return BT709HIP_ERR_INVALID_ARG;
  const size_t px = o->format == BT709HIP_FORMAT_RGBA16F ? 8 : 4;  // bytes per output pixel
  if (o->stride < static_cast<size_t>(out_w) * px || (o->stride & (px - 1)) || !aligned(o->bgra, px))
    return BT709HIP_ERR_STRIDE;
  if (f->y_stride > 0xffffffffu || f->cbcr_stride > 0xffffffffu || o->stride > 0xffffffffu)
    return BT709HIP_ERR_STRIDE;
  return BT709HIP_OK;
}

// Device copy of a host table.  *dst is written only when the copy has succeeded, so a field that
// doubles as the "already built" marker never points at uninitialised memory.
int upload_table(const void *src, size_t bytes, void **dst) {
  void *d = nullptr;
  HIP_TRY(hipMalloc(&d, bytes));
  const hipError_t e = hipMemcpy(d, src, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return hip_fail(e);
  }
  *dst = d;
  return BT709HIP_OK;
}

// hipMalloc + blocking hipMemcpy are illegal while the calling thread records a graph; lazily
// built tables must exist before bt709hip_graph_begin_capture.
bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// -decodeBT709 calls -setupMetal first (.m:228-231).  The first setup allocates and copies, which a
// recording stream must not see: set the decoder up before bt709hip_graph_begin_capture.
int ensure_setup(bt709hip_decoder *dec, void *stream) {
  if (dec->ready.load(std::memory_order_acquire)) return BT709HIP_OK;  // setup itself is serialised by its mutex
  if (dec->ctx != nullptr && capturing(static_cast<hipStream_t>(stream))) return BT709HIP_ERR_NOT_SETUP;
  return bt709hip_decoder_setup(dec);
}

// Threshold table of the RGBA16F composite, built on the first decode into such a target (or by
// bt709hip_decoder_prepare_format; not while recording a graph).
int ensure_half_table(bt709hip_decoder *dec, void *stream) {
  std::lock_guard<std::mutex> lock(dec->setup_mutex);
  if (dec->half_ready) return BT709HIP_OK;
  if (capturing(static_cast<hipStream_t>(stream))) return BT709HIP_ERR_NOT_SETUP;
  HalfTable t;
  if (!build_half_table(dec->gamma, &t)) return BT709HIP_ERR_UNSUPPORTED;
  HalfParams hp = {};
  hp.split = t.split;
  hp.low_scale = t.low_scale;
  hp.index_scale = t.index_scale;
  hp.h_min = t.h_min;
  if (t.split <= 1.0f) {  // a curve: the table covers [h_min, H(1.0)]; its last real entry is followed by +inf
    size_t real = t.thresholds.size();
    while (real > 0 && t.thresholds[real - 1] == std::numeric_limits<float>::infinity()) --real;
    hp.h_max = t.h_min + static_cast<uint32_t>(real) - 1;
    // device image: a guard entry below (T[h_min - 1] = 0: no x is "below" it) and two +inf above
    // (T[h_max + 1], T[h_max + 2]), so a candidate one code off either end needs no clamp; 16-byte multiple
    std::vector<float> image;
    image.push_back(0.0f);
    image.insert(image.end(), t.thresholds.begin(), t.thresholds.begin() + static_cast<std::ptrdiff_t>(real));
    image.push_back(std::numeric_limits<float>::infinity());
    image.push_back(std::numeric_limits<float>::infinity());
    while (image.size() % 4 != 0) image.push_back(std::numeric_limits<float>::infinity());
    hp.cand_offset = static_cast<uint32_t>(image.size() * sizeof(float));  // the candidate entries ride behind the thresholds
    image.insert(image.end(), t.cand.begin(), t.cand.end());
    while (image.size() % 4 != 0) image.push_back(0.0f);
    hp.table_bytes = static_cast<uint32_t>(image.size() * sizeof(float));
    // the kernel's LDS plan (bt709_kernels.h kHalfCandLds): thresholds below the fixed start of the candidates, all under 40 KiB
    if (hp.cand_offset > kHalfCandLds || kHalfCandLds + (hp.table_bytes - hp.cand_offset) > 40u * 1024u) return BT709HIP_ERR_UNSUPPORTED;
    void *d = nullptr;
    if (int rc = upload_table(image.data(), hp.table_bytes, &d)) return rc;
    hp.table = d;
  }
  dec->half = hp;
  dec->half_ready = true;
  return BT709HIP_OK;
}

// Table pointers and lookup constants of a launch.
void set_tables(DecodeParams *p, const bt709hip_decoder *dec) {
  p->table_unit = dec->d_table_unit;
  p->table_unit_bytes = dec->table_unit_bytes;
  p->table_linear = dec->d_table_linear;
  p->table_linear_bytes = dec->table_linear_bytes;
  p->table_encode_u = dec->d_encode_u;
  p->table_encode_u_bytes = dec->encode_u_bytes;
  p->encode_u_n = static_cast<float>(dec->encode_u_n);
  p->table_encode = dec->d_encode;
  p->table_encode_bytes = dec->encode_bytes;
  p->encode_log_add = dec->encode_log_add;
  p->encode_log_first = dec->encode_log_first;
  p->unit_magic = 8388608.0f / static_cast<float>(dec->table_n);  // 2^23 / N, exact: N is a power of two
  p->unit1_magic = dec->unit1_magic;
  p->unit1_first = dec->unit1_first;
  p->unit1_shift = dec->unit1_shift;
}

int64_t byte_step(const void *a, const void *b) {
  return static_cast<int64_t>(reinterpret_cast<intptr_t>(b) - reinterpret_cast<intptr_t>(a));
}

// True when frame i sits at frame 0 + i * (frame 1 - frame 0) for every plane: a ring or pool
// carved from one allocation.  Such a batch needs no per-frame pointer table in the kernarg.
bool evenly_spaced(int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                   const bt709hip_surface *outs) {
  if (count < 2 || frames == nullptr || outs == nullptr) return false;
  const int64_t dy = byte_step(frames[0].y, frames[1].y), dc = byte_step(frames[0].cbcr, frames[1].cbcr);
  const int64_t dout = byte_step(outs[0].bgra, outs[1].bgra);
  const int64_t da = alphas ? byte_step(alphas[0].y, alphas[1].y) : 0;
  for (int i = 2; i < count; ++i) {
    if (byte_step(frames[0].y, frames[i].y) != dy * i || byte_step(frames[0].cbcr, frames[i].cbcr) != dc * i ||
        byte_step(outs[0].bgra, outs[i].bgra) != dout * i)
      return false;
    if (alphas && byte_step(alphas[0].y, alphas[i].y) != da * i) return false;
  }
  return true;
}

// Builds (once) the device tables of one (input gamma, output gamma) encoder pair.  Both uploads go
// into locals and are published together: a failed second upload leaves the pair unbuilt, not half
// built.  Refused while `s` records a graph (hipMalloc / blocking copies are illegal there): call
// bt709hip_encoder_prepare before bt709hip_graph_begin_capture.
int encoder_tables(bt709hip_context *ctx, int input_gamma, int output_gamma, hipStream_t s) {
  EncoderTables &t = ctx->encoders[input_gamma][output_gamma];
  std::lock_guard<std::mutex> lock(ctx->encoder_mutex);
  if (t.d_per_byte != nullptr) return BT709HIP_OK;
  if (capturing(s)) return BT709HIP_ERR_NOT_SETUP;
  EncodeTables host;
  SplitTable fl;
  if (!build_encode_tables(input_gamma, output_gamma, &host) || !build_split_table(host.from_linear_kind, &fl))
    return BT709HIP_ERR_UNSUPPORTED;
  void *d_fl = nullptr, *d_pb = nullptr;
  const uint32_t fl_bytes = static_cast<uint32_t>(fl.buckets.size() * sizeof(TransferBucket));
  if (int rc = upload_table(fl.buckets.data(), fl_bytes, &d_fl)) return rc;
  if (int rc = upload_table(host.per_byte, sizeof host.per_byte, &d_pb)) {
    (void)hipFree(d_fl);
    return rc;
  }
  t.from_linear_n = fl.n_fine;
  t.split = fl.split;
  t.coarse_scale = fl.coarse_scale;
  t.coarse_offset = fl.coarse_offset;
  t.from_linear_bytes = fl_bytes;
  t.d_from_linear = d_fl;
  t.d_per_byte = d_pb;  // the "built" marker: last
  return BT709HIP_OK;
}

uint32_t grid_x_for(const bt709hip_context *ctx, uint32_t rows, int frames) {
  uint32_t per_frame = static_cast<uint32_t>(ctx->grid_blocks / (frames > 0 ? frames : 1));
  if (per_frame < 1) per_frame = 1;
  return rows < per_frame ? rows : per_frame;
}

int decode_batch_now(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                     const bt709hip_surface *outs, void *stream, int wait_until_completed);

// Launches what `q` holds (dec->queue_mutex held).  The queue is emptied first: a failed launch is reported once, to the
// call that issued it, and never re-issued.
int issue_queue(bt709hip_decoder *dec, PendingQueue &q) {
  if (q.frames.empty()) return BT709HIP_OK;
  std::vector<bt709hip_frame> frames, alphas;
  std::vector<bt709hip_surface> outs;
  frames.swap(q.frames);
  alphas.swap(q.alphas);
  outs.swap(q.outs);
  return decode_batch_now(dec, static_cast<int>(frames.size()), frames.data(), q.with_alphas ? alphas.data() : nullptr, outs.data(),
                          q.stream, 0);
}

int64_t now_us() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// one decoder: the queue of stream `s`, or every queue (all = true); aged_only: only queues older than the decoder's age limit
int flush_decoder(bt709hip_decoder *dec, hipStream_t s, bool all, bool aged_only = false) {
  std::lock_guard<std::mutex> lock(dec->queue_mutex);
  int rc = BT709HIP_OK;
  const int64_t limit = aged_only ? now_us() - dec->coalesce_max_age_us : 0;
  for (PendingQueue &q : dec->queues) {
    if (q.frames.empty()) continue;
    const bool aged = dec->coalesce_max_age_us > 0 && q.oldest_us <= limit;
    if (aged_only ? (aged || q.stream == s) : (all || q.stream == s))
      if (int e = issue_queue(dec, q)) rc = rc ? rc : e;
  }
  return rc;
}

// Every coalescing decoder of `ctx` except `skip`: called by each entry point that takes a stream, before it touches the stream
// -- the queue of THAT stream goes out (stream order), and so does any queue of any stream that has outlived its decoder's
// BT709HIP_OPT_COALESCE_MAX_AGE_US (nothing here runs on a timer: an idle caller's frames wait for the context's next call).
// coalescing_mutex is held across the loop (lock order: coalescing_mutex, then a decoder's queue_mutex): a decoder that is being
// destroyed leaves the list under the same mutex (set_coalescing), so none of the pointers can dangle.
int flush_stream(bt709hip_context *ctx, hipStream_t s, const bt709hip_decoder *skip = nullptr) {
  if (ctx == nullptr || ctx->n_coalescing.load(std::memory_order_acquire) == 0) return BT709HIP_OK;
  std::lock_guard<std::mutex> lock(ctx->coalescing_mutex);
  int rc = BT709HIP_OK;
  for (bt709hip_decoder *d : ctx->coalescing)
    if (d != skip)
      if (int e = flush_decoder(d, s, false, true)) rc = rc ? rc : e;
  return rc;
}

#define FLUSH_STREAM(ctx, stream)                                                        \
  do {                                                                                   \
    if ((ctx) != nullptr)                                                                \
      if (int _rc = flush_stream((ctx), pick((ctx), (stream)))) return _rc;              \
  } while (0)

// Turns the coalescing submit of `dec` on (n > 1), off (0) or changes its count.  Order matters when other threads are inside a
// decode: ON registers the decoder with its context BEFORE the count becomes visible (a frame queued from then on is seen by
// every flush_stream), OFF issues what is queued and clears the count under the queue_mutex -- a submit that is waiting for that
// mutex re-reads the count behind it and launches instead of queueing -- and only then leaves the context's list.  Lock order:
// the context's coalescing_mutex and a decoder's queue_mutex are never held together here.  Returns the flush's status.
int set_coalescing(bt709hip_decoder *dec, int n) {
  bt709hip_context *ctx = dec->ctx;
  const bool now = n > 1;
  if (now && ctx != nullptr) {
    std::lock_guard<std::mutex> lock(ctx->coalescing_mutex);
    if (std::find(ctx->coalescing.begin(), ctx->coalescing.end(), dec) == ctx->coalescing.end()) ctx->coalescing.push_back(dec);
    ctx->n_coalescing.store(static_cast<int>(ctx->coalescing.size()), std::memory_order_release);
  }
  int rc = BT709HIP_OK;
  {
    std::lock_guard<std::mutex> lock(dec->queue_mutex);
    if (!now)  // off: nothing may stay queued behind the switch
      for (PendingQueue &q : dec->queues)
        if (int e = issue_queue(dec, q)) rc = rc ? rc : e;
    dec->coalesce.store(now ? n : 0);
  }
  if (!now && ctx != nullptr) {
    std::lock_guard<std::mutex> lock(ctx->coalescing_mutex);
    auto it = std::find(ctx->coalescing.begin(), ctx->coalescing.end(), dec);
    if (it != ctx->coalescing.end()) ctx->coalescing.erase(it);
    ctx->n_coalescing.store(static_cast<int>(ctx->coalescing.size()), std::memory_order_release);
  }
  return rc;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ context

int bt709hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    tl_hip_error = e;
    (void)hipGetLastError();  // read and cleared: see hip_fail
    return e == hipErrorNoDevice ? 0 : BT709HIP_ERR_HIP;
  }
  return n;
}

int bt709hip_context_create(int device_ordinal, bt709hip_context **out) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    tl_hip_error = e;
    return BT709HIP_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= n) return BT709HIP_ERR_NO_DEVICE;
  bt709hip_context *ctx = new (std::nothrow) bt709hip_context();
  if (ctx == nullptr) return BT709HIP_ERR_INVALID_ARG;
  ctx->device = device_ordinal;
  e = hipSetDevice(device_ordinal);
  if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->props, device_ordinal);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->default_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = prepare_kernels();
  if (e == hipSuccess) e = prepare_rescale_kernels();
  if (e == hipSuccess) e = prepare_encode_kernels();
  if (e == hipSuccess) e = prepare_rgba16f_kernels();
  if (e != hipSuccess) {
    delete ctx;
    return hip_fail(e);
  }
  // Workgroups per launch: enough to fill every CU at 8 resident blocks, times a
  // small factor so the tail is short; row pairs are grid-strided beyond that.
  ctx->grid_blocks = ctx->props.multiProcessorCount * 8 * ctx->grid_mult;
  *out = ctx;
  return BT709HIP_OK;
}

int bt709hip_abi_version(void) { return BT709HIP_VERSION; }

int bt709hip_context_set_option(bt709hip_context *ctx, int option, int value) {
  if (ctx == nullptr) return BT709HIP_ERR_INVALID_ARG;
  switch (option) {
    case BT709HIP_CTX_OPT_GRID_MULT:
      ctx->grid_mult = value <= 0 ? 2 : clamp_int(value, 1, 64);
      ctx->grid_blocks = ctx->props.multiProcessorCount * 8 * ctx->grid_mult;
      return BT709HIP_OK;
    case BT709HIP_CTX_OPT_ENCODE_ROW_PAIRS:
      ctx->encode_row_pairs = clamp_int(value, 0, 64);
      return BT709HIP_OK;
    case BT709HIP_CTX_OPT_ENCODE_THREADS:
      ctx->encode_threads = clamp_int(value, 0, 1024) / 64 * 64;
      return BT709HIP_OK;
    case BT709HIP_CTX_OPT_XCD_BANDS:
      ctx->xcd_bands = value != 0;
      return BT709HIP_OK;
    case BT709HIP_CTX_OPT_STREAMING_TRIES:
      ctx->streaming_tries = value <= 0 ? 4 : clamp_int(value, 1, 32);
      return BT709HIP_OK;
    default:
      return BT709HIP_ERR_INVALID_ARG;
  }
}

int bt709hip_context_destroy(bt709hip_context *ctx) {
  if (ctx == nullptr) return BT709HIP_OK;
  if (hipSetDevice(ctx->device) == hipSuccess && ctx->default_stream) {
    (void)hipStreamSynchronize(ctx->default_stream);
    (void)hipStreamDestroy(ctx->default_stream);
    for (auto &row : ctx->encoders)
      for (EncoderTables &t : row) {
        if (t.d_per_byte) (void)hipFree(t.d_per_byte);
        if (t.d_from_linear) (void)hipFree(t.d_from_linear);
      }
    if (ctx->d_render_encode) (void)hipFree(ctx->d_render_encode);
    if (ctx->d_render_lin) (void)hipFree(ctx->d_render_lin);
  }
  delete ctx;
  return BT709HIP_OK;
}

int bt709hip_context_info(const bt709hip_context *ctx, bt709hip_device_info *info) {
  if (ctx == nullptr || info == nullptr) return BT709HIP_ERR_INVALID_ARG;
  std::memset(info, 0, sizeof *info);
  info->device_ordinal = ctx->device;
  info->compute_units = ctx->props.multiProcessorCount;
  info->wavefront_size = ctx->props.warpSize;
  info->lds_bytes_per_block = static_cast<int32_t>(ctx->props.sharedMemPerBlock);
  info->memory_clock_khz = ctx->props.memoryClockRate;
  info->memory_bus_width_bits = ctx->props.memoryBusWidth;
  info->l2_bytes = ctx->props.l2CacheSize;
  info->clock_khz = ctx->props.clockRate;
  info->total_memory_bytes = ctx->props.totalGlobalMem;
  std::snprintf(info->name, sizeof info->name, "%s", ctx->props.name);
  std::snprintf(info->arch, sizeof info->arch, "%s", ctx->props.gcnArchName);
  // which physical device: the bus id as the runtime prints it, the UUID as 32 hex digits (both empty if the runtime has none)
  if (hipDeviceGetPCIBusId(info->pci_bus_id, static_cast<int>(sizeof info->pci_bus_id), ctx->device) != hipSuccess) {
    (void)hipGetLastError();
    std::snprintf(info->pci_bus_id, sizeof info->pci_bus_id, "%04x:%02x:%02x.0", ctx->props.pciDomainID, ctx->props.pciBusID, ctx->props.pciDeviceID);
  }
  hipUUID uuid;
  if (hipDeviceGetUuid(&uuid, ctx->device) == hipSuccess) {
    // ROCm hands out 16 ASCII characters ("4a6a3df9d3b4a52e", what rocm-smi prints as the unique id); anything else: 32 hex digits
    bool text = true;
    for (int i = 0; i < 16; ++i) text = text && uuid.bytes[i] >= 0x21 && uuid.bytes[i] <= 0x7e;
    for (int i = 0; i < 16; ++i) {
      if (text) info->uuid[i] = uuid.bytes[i], info->uuid[i + 1] = 0;
      else std::snprintf(info->uuid + 2 * i, 3, "%02x", static_cast<unsigned>(static_cast<unsigned char>(uuid.bytes[i])));
    }
  } else {
    (void)hipGetLastError();
  }
  return BT709HIP_OK;
}

int bt709hip_stream_create(bt709hip_context *ctx, void **stream) {
  if (stream == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return BT709HIP_OK;
}

int bt709hip_stream_create_with_priority(bt709hip_context *ctx, int priority, void **stream) {
  if (stream == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  int least = 0, greatest = 0;  // numerically: least priority >= greatest priority
  HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const int p = priority > least ? least : (priority < greatest ? greatest : priority);
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p));
  *stream = s;
  return BT709HIP_OK;
}

int bt709hip_stream_destroy(bt709hip_context *ctx, void *stream) {
  if (stream == nullptr) return BT709HIP_OK;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);
  HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return BT709HIP_OK;
}

int bt709hip_stream_synchronize(bt709hip_context *ctx, void *stream) {
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);
  HIP_TRY(hipStreamSynchronize(pick(ctx, stream)));
  return BT709HIP_OK;
}

int bt709hip_event_create(bt709hip_context *ctx, void **event) {
  if (event == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  hipEvent_t ev = nullptr;
  HIP_TRY(hipEventCreate(&ev));
  *event = ev;
  return BT709HIP_OK;
}

int bt709hip_event_destroy(bt709hip_context *ctx, void *event) {
  if (event == nullptr) return BT709HIP_OK;
  if (int rc = bind(ctx)) return rc;
  HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return BT709HIP_OK;
}

int bt709hip_event_record(bt709hip_context *ctx, void *event, void *stream) {
  if (event == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);
  HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), pick(ctx, stream)));
  return BT709HIP_OK;
}

int bt709hip_event_synchronize(bt709hip_context *ctx, void *event) {
  if (event == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return BT709HIP_OK;
}

int bt709hip_stream_wait_event(bt709hip_context *ctx, void *stream, void *event) {
  if (event == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);  // what this stream queued before the wait runs before it
  HIP_TRY(hipStreamWaitEvent(pick(ctx, stream), static_cast<hipEvent_t>(event), 0));
  return BT709HIP_OK;
}

int bt709hip_event_elapsed_ms(bt709hip_context *ctx, void *start, void *stop, float *ms) {
  if (start == nullptr || stop == nullptr || ms == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return BT709HIP_OK;
}

int bt709hip_graph_begin_capture(bt709hip_context *ctx, void *stream) {
  if (stream == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);  // frames queued before the recording are not part of it
  HIP_TRY(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
  return BT709HIP_OK;
}

int bt709hip_graph_end_capture(bt709hip_context *ctx, void *stream, void **graph) {
  if (stream == nullptr || graph == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *graph = nullptr;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);  // frames queued during the recording are recorded
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture(static_cast<hipStream_t>(stream), &g));
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return hip_fail(e);
  *graph = exec;
  return BT709HIP_OK;
}

int bt709hip_graph_launch(bt709hip_context *ctx, void *graph, void *stream) {
  if (graph == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);
  HIP_TRY(hipGraphLaunch(static_cast<hipGraphExec_t>(graph), pick(ctx, stream)));
  return BT709HIP_OK;
}

int bt709hip_graph_destroy(bt709hip_context *ctx, void *graph) {
  if (graph == nullptr) return BT709HIP_OK;
  if (int rc = bind(ctx)) return rc;
  HIP_TRY(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph)));
  return BT709HIP_OK;
}

int bt709hip_malloc(bt709hip_context *ctx, size_t bytes, void **dptr) {
  if (dptr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *dptr = nullptr;
  if (int rc = bind(ctx)) return rc;
  if (bytes == 0) return BT709HIP_OK;
  HIP_TRY(hipMalloc(dptr, bytes));
  return BT709HIP_OK;
}

int bt709hip_free(bt709hip_context *ctx, void *dptr) {
  if (dptr == nullptr) return BT709HIP_OK;
  if (int rc = bind(ctx)) return rc;
  HIP_TRY(hipFree(dptr));
  return BT709HIP_OK;
}

int bt709hip_mem_info(bt709hip_context *ctx, size_t *free_bytes, size_t *total_bytes) {
  if (int rc = bind(ctx)) return rc;
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return BT709HIP_OK;
}

int bt709hip_host_alloc(bt709hip_context *ctx, size_t bytes, void **hptr) {
  if (hptr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *hptr = nullptr;
  if (int rc = bind(ctx)) return rc;
  if (bytes == 0) return BT709HIP_OK;
  HIP_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
  return BT709HIP_OK;
}

int bt709hip_host_free(bt709hip_context *ctx, void *hptr) {
  if (hptr == nullptr) return BT709HIP_OK;
  if (int rc = bind(ctx)) return rc;
  HIP_TRY(hipHostFree(hptr));
  return BT709HIP_OK;
}

int bt709hip_memset(bt709hip_context *ctx, void *dptr, int value, size_t bytes, void *stream) {
  if (int rc = bind(ctx)) return rc;
  if (bytes == 0) return BT709HIP_OK;
  if (dptr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  FLUSH_STREAM(ctx, stream);
  HIP_TRY(hipMemsetAsync(dptr, value, bytes, pick(ctx, stream)));
  return BT709HIP_OK;
}

extern "C++" {
namespace {
// Host memory that HIP has not pinned or registered (malloc, numpy, a std::vector): an asynchronous copy from or to it keeps
// reading or writing it after the call has returned -- the runtime pins the pages on the fly -- and a caller that frees the
// buffer meanwhile takes a GPU memory access fault (round 5: tools/ab_libs.py did, profiles/r05_ab_rgba16f_packed.txt 6).
bool pageable(const void *host) {
  hipPointerAttribute_t attr = {};
  if (hipPointerGetAttributes(&attr, host) != hipSuccess) {
    (void)hipGetLastError();  // older runtimes report an unregistered pointer as an error
    return true;
  }
  return attr.type == hipMemoryTypeUnregistered;
}
}  // namespace
}  // extern "C++"

int bt709hip_upload(bt709hip_context *ctx, void *dst_dev, size_t dst_pitch, const void *src_host,
                    size_t src_pitch, size_t row_bytes, size_t rows, void *stream) {
  if (int rc = bind(ctx)) return rc;
  if (row_bytes == 0 || rows == 0) return BT709HIP_OK;
  if (dst_dev == nullptr || src_host == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dst_pitch < row_bytes || src_pitch < row_bytes) return BT709HIP_ERR_STRIDE;
  FLUSH_STREAM(ctx, stream);
  hipStream_t s = pick(ctx, stream);
  HIP_TRY(hipMemcpy2DAsync(dst_dev, dst_pitch, src_host, src_pitch, row_bytes, rows, hipMemcpyHostToDevice, s));
  // pageable source: the copy is complete when the call returns (the reference's fill* methods are synchronous,
  // MetalRenderContext.m:122-160); pinned memory (bt709hip_host_alloc) stays asynchronous
  if (!capturing(s) && pageable(src_host)) HIP_TRY(hipStreamSynchronize(s));
  return BT709HIP_OK;
}

int bt709hip_download(bt709hip_context *ctx, void *dst_host, size_t dst_pitch, const void *src_dev,
                      size_t src_pitch, size_t row_bytes, size_t rows, void *stream) {
  if (int rc = bind(ctx)) return rc;
  if (row_bytes == 0 || rows == 0) return BT709HIP_OK;
  if (dst_host == nullptr || src_dev == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dst_pitch < row_bytes || src_pitch < row_bytes) return BT709HIP_ERR_STRIDE;
  FLUSH_STREAM(ctx, stream);
  hipStream_t s = pick(ctx, stream);
  HIP_TRY(hipMemcpy2DAsync(dst_host, dst_pitch, src_dev, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost, s));
  if (!capturing(s) && pageable(dst_host)) HIP_TRY(hipStreamSynchronize(s));  // as above: a pageable target is filled on return
  return BT709HIP_OK;
}

// ------------------------------------------------------------------ decoder

int bt709hip_decoder_create(bt709hip_context *ctx, int gamma, int has_alpha, bt709hip_decoder **out) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (gamma < 0 || gamma >= kGammaCount) return BT709HIP_ERR_INVALID_ARG;
  bt709hip_decoder *dec = new (std::nothrow) bt709hip_decoder();
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  dec->ctx = ctx;
  dec->has_alpha = has_alpha ? 1 : 0;
  // RGBA render supports only the sRGB gamma function (MetalBT709Decoder.m:165-169)
  dec->gamma = has_alpha ? BT709HIP_GAMMA_SRGB : gamma;
  *out = dec;
  return BT709HIP_OK;
}

int bt709hip_decoder_destroy(bt709hip_decoder *dec) {
  if (dec == nullptr) return BT709HIP_OK;
  // queued frames go out; the context forgets the decoder (under its coalescing_mutex: a flush_stream that is walking the list
  // right now finishes first)
  if (dec->ctx != nullptr && hipSetDevice(dec->ctx->device) != hipSuccess) (void)hipGetLastError();
  (void)set_coalescing(dec, 0);
  if (dec->ctx != nullptr && hipSetDevice(dec->ctx->device) == hipSuccess) {
    if (dec->d_table_unit) (void)hipFree(dec->d_table_unit);
    if (dec->d_table_linear) (void)hipFree(dec->d_table_linear);
    if (dec->d_encode) (void)hipFree(dec->d_encode);
    if (dec->d_encode_u) (void)hipFree(dec->d_encode_u);
    if (dec->half.table) (void)hipFree(const_cast<void *>(dec->half.table));
  }
  delete dec;
  return BT709HIP_OK;
}

int bt709hip_decoder_set_context(bt709hip_decoder *dec, bt709hip_context *ctx) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(dec->setup_mutex);
  if (dec->ready) return dec->ctx == ctx ? BT709HIP_OK : BT709HIP_ERR_INVALID_ARG;
  const int n = dec->coalesce;
  (void)set_coalescing(dec, 0);  // registered with the context it belongs to
  dec->ctx = ctx;
  (void)set_coalescing(dec, n);
  return BT709HIP_OK;
}

int bt709hip_decoder_set_alpha_fill(bt709hip_decoder *dec, int alpha_byte) {
  if (dec == nullptr || alpha_byte < 0 || alpha_byte > 255) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bt709hip_decoder_flush_all(dec)) return rc;  // queued frames were submitted under the old value
  dec->alpha_fill = static_cast<uint32_t>(alpha_byte);
  return BT709HIP_OK;
}

int bt709hip_decoder_get_gamma(const bt709hip_decoder *dec) {
  return dec ? dec->gamma : BT709HIP_ERR_INVALID_ARG;
}

int bt709hip_decoder_has_alpha(const bt709hip_decoder *dec) { return dec ? dec->has_alpha : BT709HIP_ERR_INVALID_ARG; }

bt709hip_context *bt709hip_decoder_context(const bt709hip_decoder *dec) { return dec ? dec->ctx : nullptr; }

int bt709hip_decoder_flush(bt709hip_decoder *dec, void *stream) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dec->ctx == nullptr) return BT709HIP_OK;  // never had a stream to queue on
  if (int rc = bind(dec->ctx)) return rc;
  return flush_decoder(dec, pick(dec->ctx, stream), false);
}

int bt709hip_decoder_flush_all(bt709hip_decoder *dec) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dec->ctx == nullptr) return BT709HIP_OK;
  if (int rc = bind(dec->ctx)) return rc;
  return flush_decoder(dec, nullptr, true);
}

int bt709hip_decoder_set_option(bt709hip_decoder *dec, int option, int value) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bt709hip_decoder_flush_all(dec)) return rc;  // queued frames were submitted under the old options
  switch (option) {
    case BT709HIP_OPT_NONTEMPORAL: dec->nontemporal = value != 0; return BT709HIP_OK;
    case BT709HIP_OPT_HALF_KERNEL: dec->half_rep = clamp_int(value, -1, 1); return BT709HIP_OK;
    case BT709HIP_OPT_HALF_WORKGROUPS: dec->half_workgroups = clamp_int(value, 0, 1 << 20); return BT709HIP_OK;
    case BT709HIP_OPT_HALF_LDS_KB: dec->half_lds_kb = clamp_int(value, 0, 160); return BT709HIP_OK;
    case BT709HIP_OPT_XCD_BANDS: dec->xcd_bands = clamp_int(value, 0, 2); return BT709HIP_OK;
    case BT709HIP_OPT_COALESCE: return set_coalescing(dec, value <= 1 ? 0 : clamp_int(value, 2, kMaxBatch));
    case BT709HIP_OPT_COALESCE_MAX_AGE_US: dec->coalesce_max_age_us = value < 0 ? 0 : value; return BT709HIP_OK;
    default: return BT709HIP_ERR_INVALID_ARG;
  }
}

int bt709hip_decoder_get_option(const bt709hip_decoder *dec, int option, int *value) {
  if (dec == nullptr || value == nullptr) return BT709HIP_ERR_INVALID_ARG;
  switch (option) {
    case BT709HIP_OPT_NONTEMPORAL: *value = dec->nontemporal ? 1 : 0; return BT709HIP_OK;
    case BT709HIP_OPT_HALF_KERNEL: *value = dec->half_rep; return BT709HIP_OK;
    case BT709HIP_OPT_HALF_WORKGROUPS: *value = dec->half_workgroups; return BT709HIP_OK;
    case BT709HIP_OPT_HALF_LDS_KB: *value = dec->half_lds_kb; return BT709HIP_OK;
    case BT709HIP_OPT_XCD_BANDS: *value = dec->xcd_bands; return BT709HIP_OK;
    case BT709HIP_OPT_COALESCE: *value = dec->coalesce; return BT709HIP_OK;
    case BT709HIP_OPT_COALESCE_MAX_AGE_US: *value = dec->coalesce_max_age_us; return BT709HIP_OK;
    default: return BT709HIP_ERR_INVALID_ARG;
  }
}

int bt709hip_decoder_setup(bt709hip_decoder *dec) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(dec->setup_mutex);
  if (dec->ctx == nullptr) return BT709HIP_ERR_NOT_SETUP;  // MetalBT709Decoder.m:48-54
  if (dec->ready) return BT709HIP_OK;                      // second call is a nop (.m:66-70)
  if (int rc = bind(dec->ctx)) return rc;

  TransferTable t;
  TransferTable enc;  // sRGB encoder of the rescale kernels: the LINEAR composite's log-bucket form (5 KiB instead of 33)
  UniformTable enc_u;
  if (!build_transfer_table(dec->gamma, &t) || !build_transfer_table(kGammaLinear, &enc) || enc.buckets_log.empty() ||
      !build_uniform_table(kGammaLinear, 256, &enc_u))
    return BT709HIP_ERR_UNSUPPORTED;
  dec->table_n = t.n;
  // the 1:1 kernels' table: the log-bucket form where the builder found one at most half the size (the LINEAR mode: 5 KiB
  // instead of 33), else the uniform buckets the rescale kernels' table_linear shares its layout with
  const bool log_form = !t.buckets_log.empty();
  const std::vector<TransferBucket> &unit = log_form ? t.buckets_log : t.buckets_unit;
  const float uniform_magic = 8388608.0f / static_cast<float>(t.n);
  uint32_t uniform_first;
  std::memcpy(&uniform_first, &uniform_magic, sizeof uniform_first);
  dec->unit1_magic = log_form ? t.log_add : uniform_magic;
  dec->unit1_first = log_form ? t.log_first : uniform_first;
  dec->unit1_shift = log_form ? 16u : 0u;
  dec->table_unit_bytes = static_cast<uint32_t>(unit.size() * sizeof(TransferBucket));
  dec->table_linear_bytes = static_cast<uint32_t>(t.buckets_linear.size() * sizeof(TransferBucketLinear));
  dec->encode_log_add = enc.log_add;
  dec->encode_log_first = enc.log_first;
  dec->encode_bytes = static_cast<uint32_t>(enc.buckets_log.size() * sizeof(TransferBucket));
  void *d_unit = nullptr, *d_linear = nullptr, *d_enc = nullptr, *d_enc_u = nullptr;
  const uint32_t enc_u_bytes = static_cast<uint32_t>(enc_u.buckets.size() * sizeof(TransferBucket));
  int rc = upload_table(unit.data(), dec->table_unit_bytes, &d_unit);
  if (rc == BT709HIP_OK) rc = upload_table(t.buckets_linear.data(), dec->table_linear_bytes, &d_linear);
  if (rc == BT709HIP_OK) rc = upload_table(enc.buckets_log.data(), dec->encode_bytes, &d_enc);
  if (rc == BT709HIP_OK) rc = upload_table(enc_u.buckets.data(), enc_u_bytes, &d_enc_u);
  if (rc != BT709HIP_OK) {  // a retry starts from scratch: nothing is published, nothing leaks
    if (d_unit) (void)hipFree(d_unit);
    if (d_linear) (void)hipFree(d_linear);
    if (d_enc) (void)hipFree(d_enc);
    return rc;
  }
  dec->d_encode_u = d_enc_u;
  dec->encode_u_bytes = enc_u_bytes;
  dec->encode_u_n = enc_u.n;
  dec->d_table_unit = d_unit;
  dec->d_table_linear = d_linear;
  dec->d_encode = d_enc;
  dec->ready.store(true, std::memory_order_release);
  return BT709HIP_OK;
}

extern "C++" {  // helpers with C++ linkage inside the C-ABI block
namespace {

enum class OutShape { kSame, kHalf, kAny };  // output size relative to the frame: pass 1 / exact 2:1 / view-fit

struct BatchInfo {
  bool uniform = false;    // frames evenly spaced in memory (no pointer table needed)
  uint32_t in_align = 16;  // largest power of two <= 16 dividing every input plane pointer and pitch
  uint32_t out_align = 16; // same for the outputs
  int format = BT709HIP_FORMAT_BGRA8_SRGB;
};

void fold_align(uint32_t *a, uintptr_t v) {
  while (*a > 1 && (v % *a) != 0) *a /= 2;
}

// Validates `count` frames (+ alpha frames) against their outputs in the reference's order
// (MetalBT709Decoder.m:265-368), checks that the batch shares one geometry, and fills the pointer
// table, pitches and frame spacing of `p`.  Returns BT709HIP_OK with p->width == 0 for empty frames.
int gather_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                 const bt709hip_surface *outs, OutShape shape, void *stream, DecodeParams *p, BatchInfo *info) {
  if (dec == nullptr || frames == nullptr || outs == nullptr || count < 0) return BT709HIP_ERR_INVALID_ARG;
  // an alpha buffer handed to an opaque decoder is validated (.m:294-306, 357-368) but not read
  const bt709hip_frame *planes_a = dec->has_alpha ? alphas : nullptr;
  info->uniform = evenly_spaced(count, frames, planes_a, outs);
  if (count > (info->uniform ? kMaxUniformBatch : kMaxBatch)) return BT709HIP_ERR_UNSUPPORTED;
  if (int rc = ensure_setup(dec, stream)) return rc;
  std::memset(p, 0, sizeof *p);
  if (count == 0) return BT709HIP_OK;
  if (int rc = bind(dec->ctx)) return rc;

  const bt709hip_frame &f0 = frames[0];
  const bt709hip_surface &o0 = outs[0];
  info->format = o0.format;
  for (int i = 0; i < count; ++i) {
    const bt709hip_frame &f = frames[i];
    const bt709hip_surface &o = outs[i];
    const bt709hip_frame *a = alphas ? &alphas[i] : nullptr;
    if (shape == OutShape::kHalf && ((f.width & 3) || (f.height & 3))) return BT709HIP_ERR_ODD_DIMENSIONS;
    const int want_w = shape == OutShape::kSame ? f.width : (shape == OutShape::kHalf ? f.width / 2 : o.width);
    const int want_h = shape == OutShape::kSame ? f.height : (shape == OutShape::kHalf ? f.height / 2 : o.height);
    if (shape == OutShape::kAny && (o.width < 0 || o.height < 0)) return BT709HIP_ERR_INVALID_ARG;
    if (int rc = validate(dec, &f, a, &o, want_w, want_h, o.width, o.height)) return rc;
    if (o.format != BT709HIP_FORMAT_BGRA8_SRGB && (shape != OutShape::kSame || o.format != BT709HIP_FORMAT_RGBA16F))
      return BT709HIP_ERR_UNSUPPORTED;
    if (f.width != f0.width || f.height != f0.height || f.y_stride != f0.y_stride || f.cbcr_stride != f0.cbcr_stride ||
        o.stride != o0.stride || o.width != o0.width || o.height != o0.height || o.format != o0.format)
      return BT709HIP_ERR_SIZE_MISMATCH;
    if (planes_a != nullptr && a->y_stride != alphas[0].y_stride) return BT709HIP_ERR_SIZE_MISMATCH;
    if (i < kMaxBatch) {
      p->frames[i].y = static_cast<const uint8_t *>(f.y);
      p->frames[i].cbcr = static_cast<const uint8_t *>(f.cbcr);
      p->frames[i].alpha = planes_a ? static_cast<const uint8_t *>(a->y) : nullptr;
      p->frames[i].out = static_cast<uint8_t *>(o.bgra);
    }
    fold_align(&info->in_align, reinterpret_cast<uintptr_t>(f.y));
    fold_align(&info->in_align, reinterpret_cast<uintptr_t>(f.cbcr));
    if (planes_a) fold_align(&info->in_align, reinterpret_cast<uintptr_t>(a->y));
    fold_align(&info->out_align, reinterpret_cast<uintptr_t>(o.bgra));
  }
  fold_align(&info->in_align, f0.y_stride);
  fold_align(&info->in_align, f0.cbcr_stride);
  if (planes_a) fold_align(&info->in_align, alphas[0].y_stride);
  fold_align(&info->out_align, o0.stride);
  // the row-pair dimension of the 1:1 and 2:1 kernels is gridDim.y; the any-ratio kernel walks strips of OUTPUT rows
  // in a 1-D grid and has its own (documented) limits: 65535 output rows, planes under 2 GiB
  if (shape != OutShape::kAny && (f0.height / 2 > kMaxGridYZ || o0.height > 2 * kMaxGridYZ)) return BT709HIP_ERR_UNSUPPORTED;
  if (f0.width == 0 || f0.height == 0 || o0.width == 0 || o0.height == 0) return BT709HIP_OK;

  if (info->uniform && count > 1) {
    p->uniform = 1;
    p->step_y = byte_step(frames[0].y, frames[1].y);
    p->step_cbcr = byte_step(frames[0].cbcr, frames[1].cbcr);
    p->step_alpha = planes_a ? byte_step(alphas[0].y, alphas[1].y) : 0;
    p->step_out = byte_step(outs[0].bgra, outs[1].bgra);
  }
  set_tables(p, dec);
  p->width = static_cast<uint32_t>(f0.width);
  p->height = static_cast<uint32_t>(f0.height);
  p->y_stride = static_cast<uint32_t>(f0.y_stride);
  p->cbcr_stride = static_cast<uint32_t>(f0.cbcr_stride);
  p->alpha_stride = planes_a ? static_cast<uint32_t>(alphas[0].y_stride) : 0;
  p->out_stride = static_cast<uint32_t>(o0.stride);
  p->out_width = static_cast<uint32_t>(o0.width);
  p->out_height = static_cast<uint32_t>(o0.height);
  p->alpha_word = dec->alpha_fill << 24;
  return BT709HIP_OK;
}

int finish_launch(hipStream_t s, int wait_until_completed) {
  HIP_TRY(hipGetLastError());
  if (wait_until_completed) HIP_TRY(hipStreamSynchronize(s));  // .m:486-489
  return BT709HIP_OK;
}

}  // namespace
}  // extern "C++"

extern "C++" {  // helpers with C++ linkage inside the C-ABI block
namespace {

// the launch itself (no queueing)
int decode_batch_now(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                     const bt709hip_surface *outs, void *stream, int wait_until_completed) {
  DecodeParams p;
  BatchInfo info;
  if (int rc = gather_batch(dec, count, frames, alphas, outs, OutShape::kSame, stream, &p, &info)) return rc;
  if (p.width == 0) return BT709HIP_OK;
  hipStream_t s = pick(dec->ctx, stream);
  if (info.format == BT709HIP_FORMAT_RGBA16F) {  // the reference's pre-10.14 intermediate: linear-light halves
    if (int rc = ensure_half_table(dec, stream)) return rc;
    last_launch_shape() = LaunchShape{};
    tl_kernel_name = launch_decode_rgba16f(p, dec->half, count, dec->has_alpha != 0, info.in_align, info.out_align,
                                           static_cast<uint32_t>(dec->ctx->props.multiProcessorCount), dec->xcd_bands != 0, s);
    return finish_launch(s, wait_until_completed);
  }
  // Fast path: one short-lived workgroup per tile of a row pair, dispatched in address order
  // (see the kernel file's header).  General path keeps the grid-strided shape.
  const bool fast = (p.width % 4) == 0 && info.in_align >= 4 && info.out_align >= 16;
  const uint32_t gx = fast ? quads_tiles(p.width) : grid_x_for(dec->ctx, p.height / 2, count);
  const uint32_t threads = quads_block_threads(p.width);
  last_launch_shape() = LaunchShape{};
  tl_kernel_name = launch_decode(p, count, fast ? kVariantQuads : kVariantBlocks, dec->has_alpha != 0,
                                 dec->gamma == kGammaSRGB, dec->nontemporal, dec->xcd_bands, gx, threads, s);
  return finish_launch(s, wait_until_completed);
}

bool same_shape(const bt709hip_frame &a, const bt709hip_frame &b) {
  return a.width == b.width && a.height == b.height && a.y_stride == b.y_stride && a.cbcr_stride == b.cbcr_stride;
}

// BT709HIP_OPT_COALESCE (include/bt709hip.h, COALESCING SUBMIT): validate now, launch later.
int coalescing_submit(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                      const bt709hip_surface *outs, void *stream, int wait_until_completed) {
  if (dec->ctx == nullptr) return decode_batch_now(dec, count, frames, alphas, outs, stream, wait_until_completed);
  hipStream_t s = pick(dec->ctx, stream);
  // what the context's OTHER coalescing decoders have queued for this stream was submitted before this call: it goes first.
  // Done before this decoder's own queue_mutex is taken (lock order: the context's coalescing_mutex, then a queue_mutex).
  if (int rc = bind(dec->ctx)) return rc;
  if (int rc = flush_stream(dec->ctx, s, dec)) return rc;
  std::lock_guard<std::mutex> lock(dec->queue_mutex);
  PendingQueue *q = nullptr;
  if (dec->coalesce_max_age_us > 0) {  // this decoder's queues of OTHER streams that have waited too long
    const int64_t limit = now_us() - dec->coalesce_max_age_us;
    for (PendingQueue &c : dec->queues)
      if (c.stream != s && !c.frames.empty() && c.oldest_us <= limit)
        if (int rc = issue_queue(dec, c)) return rc;
  }
  for (PendingQueue &c : dec->queues)
    if (c.stream == s) q = &c;
  const int n = dec->coalesce.load();  // read ONCE, behind the mutex: set_coalescing changes it under the same mutex
  const bool eligible = n > 1 && wait_until_completed == 0 && count >= 1 && count < n && frames != nullptr && outs != nullptr;
  if (!eligible) {  // in stream order: what is queued goes first
    if (q != nullptr)
      if (int rc = issue_queue(dec, *q)) return rc;
    return decode_batch_now(dec, count, frames, alphas, outs, stream, wait_until_completed);
  }
  {  // the call's own status: everything -decodeBT709: checks, now
    DecodeParams p;
    BatchInfo info;
    if (int rc = gather_batch(dec, count, frames, alphas, outs, OutShape::kSame, stream, &p, &info)) return rc;
    if (p.width == 0) return BT709HIP_OK;  // empty frames: nothing to launch
  }
  if (q == nullptr) {
    dec->queues.emplace_back();
    q = &dec->queues.back();
    q->stream = s;
  }
  if (!q->frames.empty()) {
    const bool fits = q->frames.size() + static_cast<size_t>(count) <= static_cast<size_t>(n) &&
                      same_shape(q->frames[0], frames[0]) && q->frames[0].transfer == frames[0].transfer &&
                      q->outs[0].stride == outs[0].stride && q->outs[0].format == outs[0].format &&
                      q->with_alphas == (alphas != nullptr) && (alphas == nullptr || q->alphas[0].y_stride == alphas[0].y_stride);
    if (!fits)
      if (int rc = issue_queue(dec, *q)) return rc;
  }
  q->with_alphas = alphas != nullptr;
  if (q->frames.empty()) q->oldest_us = now_us();
  q->frames.insert(q->frames.end(), frames, frames + count);
  if (alphas != nullptr) q->alphas.insert(q->alphas.end(), alphas, alphas + count);
  q->outs.insert(q->outs.end(), outs, outs + count);
  tl_kernel_name = "(queued: coalescing submit)";
  if (q->frames.size() >= static_cast<size_t>(n)) return issue_queue(dec, *q);
  return BT709HIP_OK;
}

}  // namespace
}  // extern "C++"

int bt709hip_decode_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames,
                          const bt709hip_frame *alphas, const bt709hip_surface *outs, void *stream,
                          int wait_until_completed) {
  if (dec != nullptr && dec->coalesce > 1) return coalescing_submit(dec, count, frames, alphas, outs, stream, wait_until_completed);
  // frames another (coalescing) decoder of the context queued on this stream were submitted first: they are issued first
  if (dec != nullptr && dec->ctx != nullptr && dec->ctx->n_coalescing.load(std::memory_order_acquire) != 0) {
    if (int rc = bind(dec->ctx)) return rc;
    FLUSH_STREAM(dec->ctx, stream);
  }
  return decode_batch_now(dec, count, frames, alphas, outs, stream, wait_until_completed);
}

int bt709hip_decode(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                    const bt709hip_surface *out, int render_width, int render_height, void *stream,
                    int wait_until_completed) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = ensure_setup(dec, stream)) return rc;
  // render size is a property of this call only; check it here, the rest in the batch path
  if (int rc = validate(dec, frame, alpha, out, frame ? frame->width : 0, frame ? frame->height : 0, render_width,
                        render_height))
    return rc;
  return bt709hip_decode_batch(dec, 1, frame, alpha, out, stream, wait_until_completed);
}

int bt709hip_unconvert_batch(bt709hip_decoder *dec, int count, const void *const *ycbcr_words, size_t in_stride, int width, int height,
                             const bt709hip_surface *outs, void *stream, int wait_until_completed) {
  if (dec == nullptr || outs == nullptr || ycbcr_words == nullptr || width < 0 || height < 0 || count < 0) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = ensure_setup(dec, stream)) return rc;
  FLUSH_STREAM(dec->ctx, stream);
  if (dec->has_alpha) return BT709HIP_ERR_UNSUPPORTED;  // the packed words carry no alpha sample
  if (count == 0) return BT709HIP_OK;
  const size_t row = static_cast<size_t>(width) * 4;
  bool vec = (width % 4) == 0 && (in_stride % 16) == 0;
  bool uniform = count > 1;
  std::vector<void *> out_ptrs(static_cast<size_t>(count));
  for (int i = 0; i < count; ++i) {
    const bt709hip_surface &o = outs[i];
    if (o.width != width || o.height != height) return BT709HIP_ERR_SIZE_MISMATCH;
    if ((width & 1) || (height & 1)) return BT709HIP_ERR_ODD_DIMENSIONS;  // BGRAToBT709Converter.m:69-74
    if (o.format != BT709HIP_FORMAT_BGRA8_SRGB || o.reserved != 0) return BT709HIP_ERR_UNSUPPORTED;
    if (o.stride != outs[0].stride) return BT709HIP_ERR_SIZE_MISMATCH;
    if (width == 0 || height == 0) continue;
    if (ycbcr_words[i] == nullptr || o.bgra == nullptr) return BT709HIP_ERR_INVALID_ARG;
    if (in_stride < row || (in_stride & 3) || !aligned(ycbcr_words[i], 4) || o.stride < row || (o.stride & 3) || !aligned(o.bgra, 4) ||
        in_stride > 0xffffffffu || o.stride > 0xffffffffu)
      return BT709HIP_ERR_STRIDE;
    vec = vec && (o.stride % 16) == 0 && aligned(ycbcr_words[i], 16) && aligned(o.bgra, 16);
    out_ptrs[static_cast<size_t>(i)] = o.bgra;
    if (i >= 2)
      uniform = uniform && byte_step(ycbcr_words[0], ycbcr_words[i]) == byte_step(ycbcr_words[0], ycbcr_words[1]) * i &&
                byte_step(outs[0].bgra, o.bgra) == byte_step(outs[0].bgra, outs[1].bgra) * i;
  }
  if (width == 0 || height == 0) return BT709HIP_OK;
  if (count > (uniform ? kMaxUniformBatch : kMaxBatch)) return BT709HIP_ERR_UNSUPPORTED;
  if (height > kMaxGridYZ) return BT709HIP_ERR_UNSUPPORTED;
  if (int rc = bind(dec->ctx)) return rc;
  DecodeParams t;
  std::memset(&t, 0, sizeof t);
  set_tables(&t, dec);
  t.alpha_word = dec->alpha_fill << 24;
  UnconvertBatch batch;
  batch.count = count;
  batch.uniform = uniform;
  batch.in = ycbcr_words;
  batch.out = out_ptrs.data();
  batch.in_step = uniform ? byte_step(ycbcr_words[0], ycbcr_words[1]) : 0;
  batch.out_step = uniform ? byte_step(outs[0].bgra, outs[1].bgra) : 0;
  hipStream_t s = pick(dec->ctx, stream);
  tl_kernel_name = launch_unconvert(t, batch, in_stride, outs[0].stride, static_cast<uint32_t>(width), static_cast<uint32_t>(height), vec,
                                    dec->gamma == kGammaSRGB, s);
  return finish_launch(s, wait_until_completed);
}

int bt709hip_unconvert(bt709hip_decoder *dec, const void *ycbcr_words, size_t in_stride, int width, int height,
                       const bt709hip_surface *out, void *stream, int wait_until_completed) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  return bt709hip_unconvert_batch(dec, 1, &ycbcr_words, in_stride, width, height, out, stream, wait_until_completed);
}

int bt709hip_decode_half_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames,
                               const bt709hip_frame *alphas, const bt709hip_surface *outs, void *stream,
                               int wait_until_completed) {
  DecodeParams p;
  BatchInfo info;
  if (dec != nullptr) FLUSH_STREAM(dec->ctx, stream);
  if (int rc = gather_batch(dec, count, frames, alphas, outs, OutShape::kHalf, stream, &p, &info)) return rc;
  if (p.width == 0) return BT709HIP_OK;
  hipStream_t s = pick(dec->ctx, stream);
  // wide: same tiling as the 1:1 kernel over the source width; narrow: 256 output pixels per workgroup
  const bool wide = info.in_align >= 4 && info.out_align >= 8;
  const uint32_t gx = wide ? quads_tiles(p.width) : (p.width / 2 + kBlockThreads - 1) / kBlockThreads;
  const uint32_t threads = wide ? quads_block_threads(p.width) : kBlockThreads;
  // Larger launches: persistent workgroups with bank-conflict-free (replicated) LDS tables, one per CU.
  // Staging ~150 KiB of LDS per workgroup pays once a CU has several tile rows to walk (measured
  // with tools/half_threshold.sh: one 8K frame = 17 tile rows per CU is already 19 % faster).
  const uint32_t cus = static_cast<uint32_t>(dec->ctx->props.multiProcessorCount);
  const uint64_t tile_rows = static_cast<uint64_t>((p.width / 4 + kRepBlockThreads - 1) / kRepBlockThreads) *
                             (p.height / 2) * static_cast<uint32_t>(count);
  const bool rep = wide && dec->half_rep != 0 && (dec->half_rep > 0 || tile_rows >= 8ull * cus);
  const uint32_t rep_groups = dec->half_workgroups > 0 ? static_cast<uint32_t>(dec->half_workgroups) : cus;
  const uint32_t rep_lds = (dec->half_lds_kb > 0 ? static_cast<uint32_t>(dec->half_lds_kb) : 160u) * 1024u;
  const char *name = rep ? launch_decode_half_rep(p, count, dec->has_alpha != 0, dec->nontemporal, rep_groups, rep_lds, s) : nullptr;
  tl_kernel_name = name ? name : launch_decode_half(p, count, wide, dec->has_alpha != 0, dec->nontemporal, gx, threads, s);
  return finish_launch(s, wait_until_completed);
}

int bt709hip_decode_half(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                         const bt709hip_surface *out, void *stream, int wait_until_completed) {
  if (frame == nullptr || out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  return bt709hip_decode_half_batch(dec, 1, frame, alpha, out, stream, wait_until_completed);
}

int bt709hip_decode_scaled_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames,
                                 const bt709hip_frame *alphas, const bt709hip_surface *outs, void *stream,
                                 int wait_until_completed) {
  DecodeParams p;
  BatchInfo info;
  // the frames are validated like any decode input; the surfaces may have any (common) size
  if (dec != nullptr) FLUSH_STREAM(dec->ctx, stream);
  if (int rc = gather_batch(dec, count, frames, alphas, outs, OutShape::kAny, stream, &p, &info)) return rc;
  if (p.width == 0) return BT709HIP_OK;
  if (p.out_height > static_cast<uint32_t>(kMaxGridYZ)) return BT709HIP_ERR_UNSUPPORTED;
  p.scale_x = static_cast<float>(p.width) / static_cast<float>(p.out_width);
  p.scale_y = static_cast<float>(p.height) / static_cast<float>(p.out_height);
  hipStream_t s = pick(dec->ctx, stream);
  const char *name = launch_decode_scaled(p, count, dec->has_alpha != 0, info.in_align,
                                          static_cast<uint32_t>(dec->ctx->props.multiProcessorCount), s);
  if (name == nullptr) return BT709HIP_ERR_UNSUPPORTED;  // a plane of 2 GiB or more
  tl_kernel_name = name;
  return finish_launch(s, wait_until_completed);
}

int bt709hip_decode_scaled(bt709hip_decoder *dec, const bt709hip_frame *frame, const bt709hip_frame *alpha,
                           const bt709hip_surface *out, void *stream, int wait_until_completed) {
  if (frame == nullptr || out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  return bt709hip_decode_scaled_batch(dec, 1, frame, alpha, out, stream, wait_until_completed);
}

int bt709hip_decoder_prepare_format(bt709hip_decoder *dec, int format) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bt709hip_decoder_setup(dec)) return rc;
  if (format == BT709HIP_FORMAT_BGRA8_SRGB) return BT709HIP_OK;
  if (format != BT709HIP_FORMAT_RGBA16F) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(dec->ctx)) return rc;
  return ensure_half_table(dec, nullptr);
}

extern "C++" {  // helpers with C++ linkage inside the C-ABI block
namespace {

// Tables of the stand-alone pass 2, built once per context: the two-resolution sRGB-encode buckets
// (as a decoder's) and lin[256] = sRGB_nonLinearNormToLinear(byteNorm(b)) (the sRGB8 sampler's decode).
int render_tables(bt709hip_context *ctx, hipStream_t s) {
  std::lock_guard<std::mutex> lock(ctx->encoder_mutex);
  if (ctx->d_render_lin != nullptr) return BT709HIP_OK;
  if (capturing(s)) return BT709HIP_ERR_NOT_SETUP;
  TransferTable enc;
  if (!build_transfer_table(kGammaLinear, &enc) || enc.buckets_log.empty()) return BT709HIP_ERR_UNSUPPORTED;
  float lin[256];
  for (int b = 0; b < 256; ++b) lin[b] = srgb_to_linear(b * (1.0f / 255.0f));
  void *d_enc = nullptr, *d_lin = nullptr;
  const uint32_t enc_bytes = static_cast<uint32_t>(enc.buckets_log.size() * sizeof(TransferBucket));
  int rc = upload_table(enc.buckets_log.data(), enc_bytes, &d_enc);
  if (rc == BT709HIP_OK) rc = upload_table(lin, sizeof lin, &d_lin);
  if (rc != BT709HIP_OK) {
    if (d_enc) (void)hipFree(d_enc);
    return rc;
  }
  ctx->render_encode_bytes = enc_bytes;
  ctx->render_encode_log_add = enc.log_add;
  ctx->render_encode_log_first = enc.log_first;
  ctx->d_render_encode = d_enc;
  ctx->d_render_lin = d_lin;  // the "built" marker: last
  return BT709HIP_OK;
}

}  // namespace
}  // extern "C++"

int bt709hip_render_scaled_prepare(bt709hip_context *ctx) {
  if (int rc = bind(ctx)) return rc;
  return render_tables(ctx, nullptr);
}

static int render_scaled_launch(bt709hip_context *ctx, int count, const bt709hip_surface *in, const bt709hip_surface *out,
                                int64_t in_step, int64_t out_step, void *stream, int wait_until_completed);

int bt709hip_render_scaled_batch(bt709hip_context *ctx, int count, const bt709hip_surface *in, const bt709hip_surface *out,
                                 void *stream, int wait_until_completed) {
  if (ctx == nullptr || in == nullptr || out == nullptr || count < 0 || count > kMaxUniformBatch) return BT709HIP_ERR_INVALID_ARG;
  if (count == 0) return BT709HIP_OK;
  // one geometry, surfaces evenly spaced in memory (a ring carved from one allocation): surface i = surface 0 + i * step
  int64_t in_step = 0, out_step = 0;
  if (count > 1) {
    in_step = static_cast<const uint8_t *>(in[1].bgra) - static_cast<const uint8_t *>(in[0].bgra);
    out_step = static_cast<const uint8_t *>(out[1].bgra) - static_cast<const uint8_t *>(out[0].bgra);
  }
  for (int i = 1; i < count; ++i) {
    if (in[i].width != in[0].width || in[i].height != in[0].height || in[i].stride != in[0].stride || in[i].format != in[0].format ||
        in[i].reserved != 0 || out[i].width != out[0].width || out[i].height != out[0].height || out[i].stride != out[0].stride ||
        out[i].format != out[0].format || out[i].reserved != 0)
      return BT709HIP_ERR_SIZE_MISMATCH;
    if (static_cast<const uint8_t *>(in[i].bgra) != static_cast<const uint8_t *>(in[0].bgra) + static_cast<int64_t>(i) * in_step ||
        static_cast<const uint8_t *>(out[i].bgra) != static_cast<const uint8_t *>(out[0].bgra) + static_cast<int64_t>(i) * out_step)
      return BT709HIP_ERR_UNSUPPORTED;
  }
  return render_scaled_launch(ctx, count, in, out, in_step, out_step, stream, wait_until_completed);
}

int bt709hip_render_scaled(bt709hip_context *ctx, const bt709hip_surface *in, const bt709hip_surface *out, void *stream,
                           int wait_until_completed) {
  return render_scaled_launch(ctx, 1, in, out, 0, 0, stream, wait_until_completed);
}

static int render_scaled_launch(bt709hip_context *ctx, int count, const bt709hip_surface *in, const bt709hip_surface *out,
                                int64_t in_step, int64_t out_step, void *stream, int wait_until_completed) {
  if (ctx == nullptr || in == nullptr || out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (in->width < 0 || in->height < 0 || out->width < 0 || out->height < 0 || in->reserved != 0 || out->reserved != 0)
    return BT709HIP_ERR_INVALID_ARG;
  if (in->format != BT709HIP_FORMAT_BGRA8_SRGB && in->format != BT709HIP_FORMAT_RGBA16F) return BT709HIP_ERR_INVALID_ARG;
  if (out->format != BT709HIP_FORMAT_BGRA8_SRGB) return BT709HIP_ERR_UNSUPPORTED;  // the view is an 8-bit sRGB drawable
  if (in->width == 0 || in->height == 0 || out->width == 0 || out->height == 0) return BT709HIP_OK;
  if (in->bgra == nullptr || out->bgra == nullptr) return BT709HIP_ERR_INVALID_ARG;
  const size_t ipx = in->format == BT709HIP_FORMAT_RGBA16F ? 8 : 4;
  if (in->stride < static_cast<size_t>(in->width) * ipx || (in->stride & (ipx - 1)) || !aligned(in->bgra, ipx) ||
      out->stride < static_cast<size_t>(out->width) * 4 || (out->stride & 3) || !aligned(out->bgra, 4) ||
      in->stride > 0xffffffffu || out->stride > 0xffffffffu)
    return BT709HIP_ERR_STRIDE;
  if (out->height > kMaxGridYZ) return BT709HIP_ERR_UNSUPPORTED;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);
  hipStream_t s = pick(ctx, stream);
  if (int rc = render_tables(ctx, s)) return rc;
  RenderParams p;
  std::memset(&p, 0, sizeof p);
  p.in = static_cast<const uint8_t *>(in->bgra);
  p.out = static_cast<uint8_t *>(out->bgra);
  p.in_stride = static_cast<uint32_t>(in->stride);
  p.out_stride = static_cast<uint32_t>(out->stride);
  p.width = static_cast<uint32_t>(in->width);
  p.height = static_cast<uint32_t>(in->height);
  p.out_width = static_cast<uint32_t>(out->width);
  p.out_height = static_cast<uint32_t>(out->height);
  p.scale_x = static_cast<float>(in->width) / static_cast<float>(out->width);
  p.scale_y = static_cast<float>(in->height) / static_cast<float>(out->height);
  p.table_encode = ctx->d_render_encode;
  p.table_lin = ctx->d_render_lin;
  p.table_encode_bytes = ctx->render_encode_bytes;
  p.encode_log_add = ctx->render_encode_log_add;
  p.encode_log_first = ctx->render_encode_log_first;
  p.in_step = in_step;
  p.out_step = out_step;
  const char *name = launch_render_scaled(p, count, in->format == BT709HIP_FORMAT_RGBA16F,
                                          static_cast<uint32_t>(ctx->props.multiProcessorCount), s);
  if (name == nullptr) return BT709HIP_ERR_UNSUPPORTED;  // a surface of 2 GiB or more
  tl_kernel_name = name;
  return finish_launch(s, wait_until_completed);
}

// ------------------------------------------------------------------ frame pool

int bt709hip_pool_destroy(bt709hip_pool *pool) {
  if (pool == nullptr) return BT709HIP_OK;
  if (pool->dec && pool->dec->ctx && hipSetDevice(pool->dec->ctx->device) == hipSuccess) {
    for (auto &s : pool->slots) {
      if (s.stream) (void)hipStreamSynchronize(s.stream), (void)hipStreamDestroy(s.stream);
      if (s.h_in) (void)hipHostFree(s.h_in);
      if (s.h_out) (void)hipHostFree(s.h_out);
      if (s.d_in) (void)hipFree(s.d_in);
      if (s.d_out) (void)hipFree(s.d_out);
    }
  }
  delete pool;
  return BT709HIP_OK;
}

extern "C++" {  // helpers with C++ linkage inside the C-ABI block
namespace {
// Device buffers the library allocates for itself (in-flight pool slots, hence the sharder's lanes): anything of 256 MB or
// more streams from HBM, where placement matters, and goes through the placement-aware allocator with the
// context's BT709HIP_CTX_OPT_STREAMING_TRIES candidates (default 4; 1 = plain hipMalloc).  A 4K slot is 33 MB + 12 MB: this
// only triggers for very large frames (e.g. 8K x 8K); the pool is PCIe-bound either way.
hipError_t alloc_pool_buffer(bt709hip_context *ctx, size_t bytes, uint8_t **out) {
  constexpr size_t kStreamingBytes = 256u << 20;
  if (bytes >= kStreamingBytes && ctx->streaming_tries > 1) {
    void *p = nullptr;
    const int rc = bt709hip_malloc_streaming(ctx, bytes, ctx->streaming_tries, &p, nullptr, nullptr);
    *out = static_cast<uint8_t *>(p);
    return rc == BT709HIP_OK ? hipSuccess : (tl_hip_error != hipSuccess ? tl_hip_error : hipErrorOutOfMemory);
  }
  return hipMalloc(reinterpret_cast<void **>(out), bytes);
}
}  // namespace
}  // extern "C++"

int bt709hip_pool_create(bt709hip_decoder *dec, int width, int height, int depth, bt709hip_pool **out) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (dec == nullptr || width <= 0 || height <= 0 || depth <= 0 || depth > 64) return BT709HIP_ERR_INVALID_ARG;
  if ((width & 1) || (height & 1)) return BT709HIP_ERR_ODD_DIMENSIONS;
  if (int rc = bt709hip_decoder_setup(dec)) return rc;
  if (int rc = bind(dec->ctx)) return rc;
  bt709hip_pool *pool = new (std::nothrow) bt709hip_pool();
  if (pool == nullptr) return BT709HIP_ERR_INVALID_ARG;
  pool->dec = dec;
  pool->width = width;
  pool->height = height;
  // Y + CbCr, plus a full-size alpha plane for a decoder with an alpha channel
  pool->in_bytes = static_cast<size_t>(width) * height * 3 / 2 + (dec->has_alpha ? static_cast<size_t>(width) * height : 0);
  pool->out_bytes = static_cast<size_t>(width) * height * 4;
  pool->slots.resize(static_cast<size_t>(depth));
  hipError_t e = hipSuccess;
  for (auto &s : pool->slots) {
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_in), pool->in_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_out), pool->out_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = alloc_pool_buffer(dec->ctx, pool->in_bytes, &s.d_in);
    if (e == hipSuccess) e = alloc_pool_buffer(dec->ctx, pool->out_bytes, &s.d_out);
  }
  if (e != hipSuccess) {
    bt709hip_pool_destroy(pool);
    return hip_fail(e);
  }
  *out = pool;
  return BT709HIP_OK;
}

int bt709hip_pool_acquire(bt709hip_pool *pool, int *slot, void **y, size_t *y_stride, void **cbcr,
                          size_t *cbcr_stride) {
  if (pool == nullptr || slot == nullptr || y == nullptr || cbcr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(pool->dec->ctx)) return rc;
  const size_t i = pool->next;
  bt709hip_pool::Slot &s = pool->slots[i];
  if (s.acquired) return BT709HIP_ERR_INVALID_ARG;  // every slot is out: submit one first
  if (s.busy) {
    HIP_TRY(hipStreamSynchronize(s.stream));  // the in-flight semaphore of the reference
    s.busy = false;
  }
  s.acquired = true;
  pool->next = (i + 1) % pool->slots.size();
  *slot = static_cast<int>(i);
  *y = s.h_in;
  *cbcr = s.h_in + static_cast<size_t>(pool->width) * pool->height;
  if (y_stride) *y_stride = static_cast<size_t>(pool->width);
  if (cbcr_stride) *cbcr_stride = static_cast<size_t>(pool->width);
  return BT709HIP_OK;
}

int bt709hip_pool_alpha_plane(bt709hip_pool *pool, int slot, void **alpha, size_t *alpha_stride) {
  if (pool == nullptr || alpha == nullptr || slot < 0 || static_cast<size_t>(slot) >= pool->slots.size())
    return BT709HIP_ERR_INVALID_ARG;
  if (!pool->dec->has_alpha) return BT709HIP_ERR_UNSUPPORTED;
  bt709hip_pool::Slot &s = pool->slots[static_cast<size_t>(slot)];
  if (!s.acquired) return BT709HIP_ERR_INVALID_ARG;
  *alpha = s.h_in + static_cast<size_t>(pool->width) * pool->height * 3 / 2;
  if (alpha_stride) *alpha_stride = static_cast<size_t>(pool->width);
  return BT709HIP_OK;
}

int bt709hip_pool_submit(bt709hip_pool *pool, int slot) {
  if (pool == nullptr || slot < 0 || static_cast<size_t>(slot) >= pool->slots.size()) return BT709HIP_ERR_INVALID_ARG;
  bt709hip_pool::Slot &s = pool->slots[static_cast<size_t>(slot)];
  if (!s.acquired) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(pool->dec->ctx)) return rc;
  const int w = pool->width, h = pool->height;
  {
    const hipError_t e = hipMemcpyAsync(s.d_in, s.h_in, pool->in_bytes, hipMemcpyHostToDevice, s.stream);
    if (e != hipSuccess) {
      s.acquired = false;  // handed back, see below
      s.busy = true;
      return hip_fail(e);
    }
  }
  bt709hip_frame f;
  std::memset(&f, 0, sizeof f);
  f.y = s.d_in;
  f.y_stride = static_cast<size_t>(w);
  f.cbcr = s.d_in + static_cast<size_t>(w) * h;
  f.cbcr_stride = static_cast<size_t>(w);
  f.width = w;
  f.height = h;
  f.matrix = BT709HIP_MATRIX_ITU_R_709_2;
  f.transfer = required_transfer(pool->dec->gamma);
  bt709hip_surface o;
  std::memset(&o, 0, sizeof o);
  o.bgra = s.d_out;
  o.stride = static_cast<size_t>(w) * 4;
  o.width = w;
  o.height = h;
  bt709hip_frame a = f;  // alpha plane: only y is read (cvpbu_wrap_y_plane_as_metal_texture)
  a.y = s.d_in + static_cast<size_t>(w) * h * 3 / 2;
  a.cbcr = nullptr;
  a.transfer = BT709HIP_TRANSFER_LINEAR;
  // From here on the slot is no longer "acquired" whatever happens: a failed submit hands it back (its staging
  // may hold a partly enqueued frame, so it counts as busy until its stream has drained) instead of leaving a
  // slot that can be neither submitted nor acquired again.
  s.acquired = false;
  s.busy = true;
  if (int rc = bt709hip_decode(pool->dec, &f, pool->dec->has_alpha ? &a : nullptr, &o, w, h, s.stream, 0)) return rc;
  if (int rc = bt709hip_decoder_flush(pool->dec, s.stream)) return rc;  // the raw copy below must follow the decode
  HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, pool->out_bytes, hipMemcpyDeviceToHost, s.stream));
  return BT709HIP_OK;
}

int bt709hip_pool_release(bt709hip_pool *pool, int slot) {
  if (pool == nullptr || slot < 0 || static_cast<size_t>(slot) >= pool->slots.size()) return BT709HIP_ERR_INVALID_ARG;
  bt709hip_pool::Slot &s = pool->slots[static_cast<size_t>(slot)];
  if (!s.acquired) return BT709HIP_ERR_INVALID_ARG;
  s.acquired = false;  // nothing was enqueued: the slot is free at once
  return BT709HIP_OK;
}

int bt709hip_pool_wait(bt709hip_pool *pool, int slot, const void **bgra, size_t *stride) {
  if (pool == nullptr || slot < 0 || static_cast<size_t>(slot) >= pool->slots.size() || bgra == nullptr)
    return BT709HIP_ERR_INVALID_ARG;
  bt709hip_pool::Slot &s = pool->slots[static_cast<size_t>(slot)];
  if (s.acquired) return BT709HIP_ERR_INVALID_ARG;  // acquired but never submitted
  if (int rc = bind(pool->dec->ctx)) return rc;
  if (s.busy) {
    HIP_TRY(hipStreamSynchronize(s.stream));
    s.busy = false;
  }
  *bgra = s.h_out;
  if (stride) *stride = static_cast<size_t>(pool->width) * 4;
  return BT709HIP_OK;
}

// ------------------------------------------------------------------ frame sharder

struct bt709hip_shard {
  struct Lane {
    bt709hip_context *ctx = nullptr;
    bt709hip_decoder *dec = nullptr;
    bt709hip_pool *pool = nullptr;
  };
  std::vector<Lane> lanes;
  int width = 0, height = 0, depth = 0, has_alpha = 0, gamma = 0;
  uint64_t next = 0;       // ticket of the next frame = frames handed out so far
  bool open = false;       // a ticket is acquired and not yet committed
  int open_slot = -1;
  // [lane * depth + pool slot] -> ticket whose pixels the slot holds (kNoTicket: none).  A slot is found by its ticket, not by
  // arithmetic: a cancelled or failed frame advances its lane's pool without taking a ticket, so slots and tickets drift apart.
  std::vector<uint64_t> owner;
  static constexpr uint64_t kNoTicket = ~0ull;
};

int bt709hip_shard_destroy(bt709hip_shard *sh) {
  if (sh == nullptr) return BT709HIP_OK;
  for (auto &l : sh->lanes) {
    if (l.pool) bt709hip_pool_destroy(l.pool);
    if (l.dec) bt709hip_decoder_destroy(l.dec);
    if (l.ctx) bt709hip_context_destroy(l.ctx);
  }
  delete sh;
  return BT709HIP_OK;
}

int bt709hip_shard_create(const int *device_ordinals, int lanes, int gamma, int has_alpha, int width, int height, int depth,
                          bt709hip_shard **out) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (device_ordinals == nullptr || lanes <= 0 || lanes > 64 || depth <= 0 || depth > 64 || width <= 0 || height <= 0)
    return BT709HIP_ERR_INVALID_ARG;
  if (gamma < 0 || gamma >= kGammaCount) return BT709HIP_ERR_INVALID_ARG;
  if ((width & 1) || (height & 1)) return BT709HIP_ERR_ODD_DIMENSIONS;
  bt709hip_shard *sh = new (std::nothrow) bt709hip_shard();
  if (sh == nullptr) return BT709HIP_ERR_INVALID_ARG;
  sh->width = width, sh->height = height, sh->depth = depth, sh->has_alpha = has_alpha ? 1 : 0;
  sh->lanes.resize(static_cast<size_t>(lanes));
  sh->owner.assign(static_cast<size_t>(lanes) * depth, bt709hip_shard::kNoTicket);
  int rc = BT709HIP_OK;
  for (int i = 0; i < lanes && rc == BT709HIP_OK; ++i) {
    bt709hip_shard::Lane &l = sh->lanes[static_cast<size_t>(i)];
    rc = bt709hip_context_create(device_ordinals[i], &l.ctx);
    if (rc == BT709HIP_OK) rc = bt709hip_decoder_create(l.ctx, gamma, has_alpha, &l.dec);
    if (rc == BT709HIP_OK) rc = bt709hip_pool_create(l.dec, width, height, depth, &l.pool);
  }
  if (rc != BT709HIP_OK) {
    bt709hip_shard_destroy(sh);
    return rc;
  }
  sh->gamma = sh->lanes[0].dec->gamma;
  *out = sh;
  return BT709HIP_OK;
}

int bt709hip_shard_lanes(const bt709hip_shard *sh) { return sh ? static_cast<int>(sh->lanes.size()) : BT709HIP_ERR_INVALID_ARG; }

int bt709hip_shard_lane_device(const bt709hip_shard *sh, int lane) {
  if (sh == nullptr || lane < 0 || static_cast<size_t>(lane) >= sh->lanes.size()) return BT709HIP_ERR_INVALID_ARG;
  return sh->lanes[static_cast<size_t>(lane)].ctx->device;
}

bt709hip_decoder *bt709hip_shard_lane_decoder(bt709hip_shard *sh, int lane) {
  if (sh == nullptr || lane < 0 || static_cast<size_t>(lane) >= sh->lanes.size()) return nullptr;
  return sh->lanes[static_cast<size_t>(lane)].dec;
}

int bt709hip_shard_acquire(bt709hip_shard *sh, uint64_t *ticket, void **y, size_t *y_stride, void **cbcr, size_t *cbcr_stride,
                           void **alpha, size_t *alpha_stride) {
  if (sh == nullptr || ticket == nullptr || y == nullptr || cbcr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (sh->has_alpha && alpha == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (sh->open) return BT709HIP_ERR_INVALID_ARG;  // one frame is being filled: commit or cancel it first
  const size_t lane = static_cast<size_t>(sh->next % sh->lanes.size());  // frame i -> lane i mod n
  bt709hip_shard::Lane &l = sh->lanes[lane];
  int slot = -1;
  if (int rc = bt709hip_pool_acquire(l.pool, &slot, y, y_stride, cbcr, cbcr_stride)) return rc;
  sh->owner[lane * sh->depth + static_cast<size_t>(slot)] = bt709hip_shard::kNoTicket;  // the slot's previous frame is gone
  if (alpha != nullptr) {
    *alpha = nullptr;
    if (sh->has_alpha) {
      if (int rc = bt709hip_pool_alpha_plane(l.pool, slot, alpha, alpha_stride)) {
        (void)bt709hip_pool_release(l.pool, slot);
        return rc;
      }
    }
  }
  sh->open = true;
  sh->open_slot = slot;
  *ticket = sh->next;
  return BT709HIP_OK;
}

int bt709hip_shard_cancel(bt709hip_shard *sh) {
  if (sh == nullptr || !sh->open) return BT709HIP_ERR_INVALID_ARG;
  bt709hip_shard::Lane &l = sh->lanes[static_cast<size_t>(sh->next % sh->lanes.size())];
  sh->open = false;
  return bt709hip_pool_release(l.pool, sh->open_slot);
}

int bt709hip_shard_commit(bt709hip_shard *sh, uint64_t ticket) {
  if (sh == nullptr || !sh->open || ticket != sh->next) return BT709HIP_ERR_INVALID_ARG;
  const size_t lane = static_cast<size_t>(ticket % sh->lanes.size());
  bt709hip_shard::Lane &l = sh->lanes[lane];
  sh->open = false;  // pool_submit hands the slot back on failure; the ticket is then void and the lane is reused
  if (int rc = bt709hip_pool_submit(l.pool, sh->open_slot)) return rc;
  sh->owner[lane * sh->depth + static_cast<size_t>(sh->open_slot)] = ticket;
  ++sh->next;
  return BT709HIP_OK;
}

int bt709hip_shard_submit(bt709hip_shard *sh, const bt709hip_frame *frame, const bt709hip_frame *alpha, uint64_t *ticket) {
  if (sh == nullptr || frame == nullptr || ticket == nullptr) return BT709HIP_ERR_INVALID_ARG;
  // the reference's order (MetalBT709Decoder.m:272-368): sizes, matrix tag, transfer tag, alpha's transfer tag
  if (frame->width != sh->width || frame->height != sh->height) return BT709HIP_ERR_SIZE_MISMATCH;
  if (alpha != nullptr && (alpha->width != frame->width || alpha->height != frame->height)) return BT709HIP_ERR_SIZE_MISMATCH;
  if (frame->matrix != BT709HIP_MATRIX_ITU_R_709_2) return BT709HIP_ERR_MATRIX;
  if (frame->transfer != required_transfer(sh->gamma)) return BT709HIP_ERR_TRANSFER;
  if (alpha != nullptr && alpha->transfer != BT709HIP_TRANSFER_LINEAR) return BT709HIP_ERR_ALPHA_TRANSFER;
  if (sh->has_alpha && (alpha == nullptr || alpha->y == nullptr)) return BT709HIP_ERR_INVALID_ARG;
  if (frame->y == nullptr || frame->cbcr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  const size_t w = static_cast<size_t>(sh->width), h = static_cast<size_t>(sh->height);
  if (frame->y_stride < w || frame->cbcr_stride < w || (sh->has_alpha && alpha->y_stride < w)) return BT709HIP_ERR_STRIDE;
  void *y = nullptr, *c = nullptr, *a = nullptr;
  size_t ys = 0, cs = 0, as = 0;
  uint64_t t = 0;
  if (int rc = bt709hip_shard_acquire(sh, &t, &y, &ys, &c, &cs, sh->has_alpha ? &a : nullptr, &as)) return rc;
  for (size_t r = 0; r < h; ++r)
    std::memcpy(static_cast<uint8_t *>(y) + r * ys, static_cast<const uint8_t *>(frame->y) + r * frame->y_stride, w);
  for (size_t r = 0; r < h / 2; ++r)
    std::memcpy(static_cast<uint8_t *>(c) + r * cs, static_cast<const uint8_t *>(frame->cbcr) + r * frame->cbcr_stride, w);
  if (sh->has_alpha)
    for (size_t r = 0; r < h; ++r)
      std::memcpy(static_cast<uint8_t *>(a) + r * as, static_cast<const uint8_t *>(alpha->y) + r * alpha->y_stride, w);
  if (int rc = bt709hip_shard_commit(sh, t)) return rc;
  *ticket = t;
  return BT709HIP_OK;
}

int bt709hip_shard_wait(bt709hip_shard *sh, uint64_t ticket, const void **bgra, size_t *stride) {
  if (sh == nullptr || bgra == nullptr) return BT709HIP_ERR_INVALID_ARG;
  // a frame's rows stay valid until its slot is handed out again: lanes * depth frames later, sooner if frames of its lane
  // were cancelled or failed in between (they consume a slot without a ticket)
  if (ticket >= sh->next) return BT709HIP_ERR_INVALID_ARG;
  const size_t lane = static_cast<size_t>(ticket % sh->lanes.size());
  for (int slot = 0; slot < sh->depth; ++slot)
    if (sh->owner[lane * sh->depth + static_cast<size_t>(slot)] == ticket)
      return bt709hip_pool_wait(sh->lanes[lane].pool, slot, bgra, stride);
  return BT709HIP_ERR_INVALID_ARG;  // never committed, or its slot has been recycled
}

// ------------------------------------------------------------------ encoder

int bt709hip_encode_batch(bt709hip_context *ctx, int count, const bt709hip_surface *ins, const bt709hip_frame *outs,
                          int input_gamma, int output_gamma, void *stream, int wait_until_completed) {
  if (ctx == nullptr || ins == nullptr || outs == nullptr || count < 0) return BT709HIP_ERR_INVALID_ARG;
  if (input_gamma < 0 || input_gamma > 2 || output_gamma < 0 || output_gamma > 2) return BT709HIP_ERR_INVALID_ARG;
  if (count == 0) return BT709HIP_OK;
  const bt709hip_surface &in0 = ins[0];
  const bt709hip_frame &out0 = outs[0];
  bool uniform = count > 1;
  bool fast = (in0.width % 4) == 0 && (in0.stride % 16) == 0 && (out0.y_stride % 4) == 0 && (out0.cbcr_stride % 4) == 0;
  EncodeParams p;
  std::memset(&p, 0, sizeof p);
  for (int i = 0; i < count; ++i) {
    const bt709hip_surface &in = ins[i];
    const bt709hip_frame &out = outs[i];
    if (in.width < 0 || in.height < 0) return BT709HIP_ERR_INVALID_ARG;
    if (in.width != out.width || in.height != out.height) return BT709HIP_ERR_SIZE_MISMATCH;
    if ((in.width & 1) || (in.height & 1)) return BT709HIP_ERR_ODD_DIMENSIONS;  // BGRAToBT709Converter.m:540-541
    if (in.height / 2 > kMaxGridYZ) return BT709HIP_ERR_UNSUPPORTED;            // row pairs live in gridDim.y
    if (in.format != BT709HIP_FORMAT_BGRA8_SRGB || in.reserved != 0) return BT709HIP_ERR_UNSUPPORTED;
    if (in.width != in0.width || in.height != in0.height || in.stride != in0.stride || out.y_stride != out0.y_stride ||
        out.cbcr_stride != out0.cbcr_stride)
      return BT709HIP_ERR_SIZE_MISMATCH;
    if (in.width == 0 || in.height == 0) continue;
    if (in.bgra == nullptr || out.y == nullptr || out.cbcr == nullptr) return BT709HIP_ERR_INVALID_ARG;
    const size_t w = static_cast<size_t>(in.width);
    if (in.stride < 4 * w || (in.stride & 3) || !aligned(in.bgra, 4) || out.y_stride < w || out.cbcr_stride < w)
      return BT709HIP_ERR_STRIDE;
    if (in.stride > 0xffffffffu || out.y_stride > 0xffffffffu || out.cbcr_stride > 0xffffffffu) return BT709HIP_ERR_STRIDE;
    fast = fast && aligned(in.bgra, 16) && aligned(out.y, 4) && aligned(out.cbcr, 4);
    if (i >= 2)
      uniform = uniform && byte_step(ins[0].bgra, in.bgra) == byte_step(ins[0].bgra, ins[1].bgra) * i &&
                byte_step(outs[0].y, out.y) == byte_step(outs[0].y, outs[1].y) * i &&
                byte_step(outs[0].cbcr, out.cbcr) == byte_step(outs[0].cbcr, outs[1].cbcr) * i;
    if (i < kMaxBatch) {
      p.frames[i].bgra = static_cast<const uint8_t *>(in.bgra);
      p.frames[i].y = static_cast<uint8_t *>(const_cast<void *>(out.y));
      p.frames[i].cbcr = static_cast<uint8_t *>(const_cast<void *>(out.cbcr));
    }
  }
  if (count > (uniform ? kMaxUniformBatch : kMaxBatch)) return BT709HIP_ERR_UNSUPPORTED;
  if (in0.width == 0 || in0.height == 0) return BT709HIP_OK;
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);

  hipStream_t s = pick(ctx, stream);
  EncoderTables &t = ctx->encoders[input_gamma][output_gamma];
  if (int rc = encoder_tables(ctx, input_gamma, output_gamma, s)) return rc;

  if (uniform) {
    p.uniform = 1;
    p.step_bgra = byte_step(ins[0].bgra, ins[1].bgra);
    p.step_y = byte_step(outs[0].y, outs[1].y);
    p.step_cbcr = byte_step(outs[0].cbcr, outs[1].cbcr);
  }
  p.per_byte = static_cast<const EncodeByteEntry *>(t.d_per_byte);
  p.from_linear = static_cast<const TransferBucket *>(t.d_from_linear);
  p.from_linear_bytes = t.from_linear_bytes;
  p.from_linear_scale = static_cast<float>(t.from_linear_n);
  p.from_linear_split = t.split;
  p.from_linear_coarse = t.coarse_scale;
  p.from_linear_offset = t.coarse_offset;
  p.row_pairs_per_block = static_cast<uint32_t>(ctx->encode_row_pairs);  // 0: sized per launch
  p.block_threads = static_cast<uint32_t>(ctx->encode_threads);
  p.width = static_cast<uint32_t>(in0.width);
  p.height = static_cast<uint32_t>(in0.height);
  p.bgra_stride = static_cast<uint32_t>(in0.stride);
  p.y_stride = static_cast<uint32_t>(out0.y_stride);
  p.cbcr_stride = static_cast<uint32_t>(out0.cbcr_stride);
  tl_kernel_name = launch_encode(p, count, fast, ctx->xcd_bands != 0, s);
  HIP_TRY(hipGetLastError());
  if (wait_until_completed) HIP_TRY(hipStreamSynchronize(s));
  return BT709HIP_OK;
}

int bt709hip_encoder_prepare(bt709hip_context *ctx, int input_gamma, int output_gamma) {
  if (ctx == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (input_gamma < 0 || input_gamma > 2 || output_gamma < 0 || output_gamma > 2) return BT709HIP_ERR_INVALID_ARG;
  if (int rc = bind(ctx)) return rc;
  return encoder_tables(ctx, input_gamma, output_gamma, nullptr);
}

int bt709hip_encode(bt709hip_context *ctx, const bt709hip_surface *in, const bt709hip_frame *out, int input_gamma,
                    int output_gamma, void *stream, int wait_until_completed) {
  if (in == nullptr || out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  return bt709hip_encode_batch(ctx, 1, in, out, input_gamma, output_gamma, stream, wait_until_completed);
}

// ------------------------------------------------------------ plane layouts

static int planes_call(bt709hip_context *ctx, const void *u, size_t u_stride, const void *v, size_t v_stride,
                       void *cbcr, size_t cbcr_stride, int cw, int ch, bool interleave, void *stream, int wait) {
  if (ctx == nullptr || cw < 0 || ch < 0) return BT709HIP_ERR_INVALID_ARG;
  if (cw == 0 || ch == 0) return BT709HIP_OK;
  if (u == nullptr || v == nullptr || cbcr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  const size_t w = static_cast<size_t>(cw);
  if (u_stride < w || v_stride < w || cbcr_stride < 2 * w) return BT709HIP_ERR_STRIDE;
  if (u_stride > 0xffffffffu || v_stride > 0xffffffffu || cbcr_stride > 0xffffffffu) return BT709HIP_ERR_STRIDE;
  if (ch > kMaxGridYZ) return BT709HIP_ERR_UNSUPPORTED;  // one chroma row per gridDim.y
  if (int rc = bind(ctx)) return rc;
  FLUSH_STREAM(ctx, stream);
  PlaneParams p;
  std::memset(&p, 0, sizeof p);
  p.u = static_cast<const uint8_t *>(u);
  p.v = static_cast<const uint8_t *>(v);
  p.cbcr = static_cast<uint8_t *>(cbcr);
  p.u_stride = static_cast<uint32_t>(u_stride);
  p.v_stride = static_cast<uint32_t>(v_stride);
  p.cbcr_stride = static_cast<uint32_t>(cbcr_stride);
  p.chroma_width = static_cast<uint32_t>(cw);
  p.chroma_height = static_cast<uint32_t>(ch);
  p.wide = (cw % 8) == 0 && (u_stride % 8) == 0 && (v_stride % 8) == 0 && (cbcr_stride % 16) == 0 && aligned(u, 8) &&
           aligned(v, 8) && aligned(cbcr, 16);
  hipStream_t s = pick(ctx, stream);
  tl_kernel_name = launch_planes(p, interleave, s);
  HIP_TRY(hipGetLastError());
  if (wait) HIP_TRY(hipStreamSynchronize(s));
  return BT709HIP_OK;
}

int bt709hip_interleave_cbcr(bt709hip_context *ctx, const void *u, size_t u_stride, const void *v, size_t v_stride,
                             void *cbcr, size_t cbcr_stride, int chroma_width, int chroma_height, void *stream,
                             int wait_until_completed) {
  return planes_call(ctx, u, u_stride, v, v_stride, cbcr, cbcr_stride, chroma_width, chroma_height, true, stream,
                     wait_until_completed);
}

int bt709hip_deinterleave_cbcr(bt709hip_context *ctx, const void *cbcr, size_t cbcr_stride, void *u, size_t u_stride,
                               void *v, size_t v_stride, int chroma_width, int chroma_height, void *stream,
                               int wait_until_completed) {
  return planes_call(ctx, u, u_stride, v, v_stride, const_cast<void *>(cbcr), cbcr_stride, chroma_width, chroma_height,
                     false, stream, wait_until_completed);
}

// -------------------------------------------------------------- diagnostics

int bt709hip_copy_probe(bt709hip_context *ctx, void *dst, const void *src, size_t bytes, void *stream) {
  if (int rc = bind(ctx)) return rc;
  if (bytes == 0) return BT709HIP_OK;
  if (dst == nullptr || src == nullptr || (bytes & 15) || !aligned(dst, 16) || !aligned(src, 16))
    return BT709HIP_ERR_INVALID_ARG;
  FLUSH_STREAM(ctx, stream);
  tl_kernel_name = launch_copy_probe(dst, src, bytes, pick(ctx, stream));
  HIP_TRY(hipGetLastError());
  return BT709HIP_OK;
}

int bt709hip_malloc_streaming(bt709hip_context *ctx, size_t bytes, int tries, void **dptr, float *rates_GBps, int *chosen) {
  if (dptr == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *dptr = nullptr;
  if (chosen) *chosen = -1;
  if (bytes == 0 || tries < 1 || tries > 32) return BT709HIP_ERR_INVALID_ARG;
  if (rates_GBps)
    for (int i = 0; i < tries; ++i) rates_GBps[i] = 0.0f;  // fully written whatever path is taken below
  if (int rc = bind(ctx)) return rc;
  // Candidates are taken ONE AT A TIME against the incumbent (round 5): at most two slabs are alive at any moment -- round 4 held
  // all `tries` of them, 34 GB for a 4K ring's output slab -- and the diversity does not suffer: hipFree + hipMalloc of a slab this
  // size hands out other physical pages (profiles/r05_hunt_budget.txt).
  const size_t half = (bytes / 2) & ~static_cast<size_t>(4095);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t s = ctx->default_stream;
  const bool probing = tries > 1 && half >= (1u << 20) && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  auto probe = [&](void *slab) -> float {
    uint8_t *p = static_cast<uint8_t *>(slab);
    // a streaming copy (lower half onto upper half) and a fill of the whole slab per round: the fill separates the
    // placements more clearly (6.3 against 6.5-6.7 TB/s where the copy shows 5.95 against 6.13, tools/placement_probes.py),
    // and a frame ring is mostly written
    for (int w = 0; w < 2; ++w) {  // warm (clocks, page tables)
      launch_copy_probe(p + half, p, half, s);
      (void)hipMemsetAsync(p, 0, bytes, s);
    }
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < 4; ++r) {
      launch_copy_probe(p + half, p, half, s);
      (void)hipMemsetAsync(p, 0, bytes, s);
    }
    (void)hipEventRecord(e1, s);
    float ms = 0.0f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.0f) ms = 1e9f;
    return static_cast<float>(4.0 * (2.0 * static_cast<double>(half) + static_cast<double>(bytes)) / (ms * 1e-3) / 1e9);
  };
  void *best_p = nullptr;
  int best = -1;
  float best_rate = -1.0f;
  for (int i = 0; i < (probing ? tries : 1); ++i) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      break;  // out of memory: the incumbent (if any) stays
    }
    const float rate = probing ? probe(p) : 0.0f;
    if (rates_GBps) rates_GBps[i] = rate;
    if (best_p == nullptr || rate > best_rate) {
      if (best_p != nullptr) (void)hipFree(best_p);
      best_p = p, best = i, best_rate = rate;
    } else {
      (void)hipFree(p);
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipGetLastError();
  if (best_p == nullptr) return hip_fail(hipErrorOutOfMemory);
  *dptr = best_p;
  if (chosen) *chosen = best;
  return BT709HIP_OK;
}

const char *bt709hip_strerror(int status) {
  switch (status) {
    case BT709HIP_OK: return "ok";
    case BT709HIP_ERR_INVALID_ARG: return "invalid argument";
    case BT709HIP_ERR_NOT_SETUP: return "decoder has no render context (setup failed)";
    case BT709HIP_ERR_SIZE_MISMATCH: return "size mismatch between BT709 input, output surface, render size or alpha";
    case BT709HIP_ERR_ODD_DIMENSIONS: return "width and height must be even (multiples of 4 for half-scale)";
    case BT709HIP_ERR_MATRIX: return "unsupported YCbCrMatrix, only BT.709 matrix is supported";
    case BT709HIP_ERR_TRANSFER: return "TransferFunction tag does not match the decoder's gamma";
    case BT709HIP_ERR_ALPHA_TRANSFER: return "alpha pixel buffer TransferFunction must be linear";
    case BT709HIP_ERR_STRIDE: return "stride smaller than a row or misaligned output";
    case BT709HIP_ERR_HIP: return "HIP runtime error";
    case BT709HIP_ERR_NO_DEVICE: return "no such HIP device";
    case BT709HIP_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

int bt709hip_last_hip_error(void) { return static_cast<int>(tl_hip_error); }
const char *bt709hip_last_hip_error_string(void) { return hipGetErrorString(tl_hip_error); }
const char *bt709hip_last_kernel_name(void) { return tl_kernel_name; }

int bt709hip_last_launch_info(bt709hip_launch_info *info) {
  if (info == nullptr) return BT709HIP_ERR_INVALID_ARG;
  const LaunchShape &s = last_launch_shape();
  for (int i = 0; i < 3; ++i) info->grid[i] = s.grid[i], info->block[i] = s.block[i];
  info->launches = s.launches;
  info->xcd_bands = s.xcd_bands;
  return BT709HIP_OK;
}

int bt709hip_gamma_thresholds(int gamma, float thresholds[255]) {
  if (thresholds == nullptr) return BT709HIP_ERR_INVALID_ARG;
  TransferTable t;
  if (!build_transfer_table(gamma, &t)) return BT709HIP_ERR_INVALID_ARG;
  std::memcpy(thresholds, t.thresholds, sizeof t.thresholds);
  return BT709HIP_OK;
}

extern "C++" {
namespace {
// one table per gamma, built on first use (host only: no device involved); nullptr: unknown gamma
const TransferTable *host_transfer_table(int gamma) {
  static std::mutex mutex;
  static TransferTable tables[kGammaCount];
  static bool built[kGammaCount] = {false, false, false, false};
  if (gamma < 0 || gamma >= kGammaCount) return nullptr;
  std::lock_guard<std::mutex> lock(mutex);
  if (!built[gamma]) {
    if (!build_transfer_table(gamma, &tables[gamma])) return nullptr;
    built[gamma] = true;
  }
  return &tables[gamma];
}
}  // namespace
}  // extern "C++"

int bt709hip_gamma_lookup(int gamma, float x, int *bucket_count, int *bucket_index_out) {
  if (gamma < 0 || gamma >= kGammaCount || !(x >= 0.0f && x <= 1.0f)) return BT709HIP_ERR_INVALID_ARG;
  const TransferTable *tp = host_transfer_table(gamma);
  if (tp == nullptr) return BT709HIP_ERR_UNSUPPORTED;
  const TransferTable &t = *tp;
  const uint32_t q = bucket_index(x, 8388608.0f / static_cast<float>(t.n));
  if (bucket_count) *bucket_count = static_cast<int>(t.n);
  if (bucket_index_out) *bucket_index_out = static_cast<int>(q);
  const TransferBucket &b = t.buckets_unit[q];
  return static_cast<int>(b.base + (x >= b.edge ? 1u : 0u));
}

int bt709hip_gamma_lookup_decode(int gamma, float x, int *bucket_count, int *bucket_index_out, int *log_form) {
  if (gamma < 0 || gamma >= kGammaCount || !(x >= 0.0f && x <= 1.0f)) return BT709HIP_ERR_INVALID_ARG;
  const TransferTable *tp = host_transfer_table(gamma);
  if (tp == nullptr) return BT709HIP_ERR_UNSUPPORTED;
  if (log_form) *log_form = tp->buckets_log.empty() ? 0 : 1;
  if (tp->buckets_log.empty()) return bt709hip_gamma_lookup(gamma, x, bucket_count, bucket_index_out);
  const uint32_t q = bucket_index_log(x, tp->log_add) - tp->log_first;  // as decoder_setup hands it to the kernels (unit1_*)
  if (bucket_count) *bucket_count = static_cast<int>(bucket_index_log(1.0f, tp->log_add) - tp->log_first + 1);
  if (bucket_index_out) *bucket_index_out = static_cast<int>(q);
  const TransferBucket &b = tp->buckets_log[q];
  return static_cast<int>(b.base + (x >= b.edge ? 1u : 0u));
}

extern "C++" {  // helpers with C++ linkage inside the C-ABI block
namespace {
const HalfTable *host_half_table(int gamma) {
  static std::mutex mutex;
  static HalfTable tables[kGammaCount];
  static bool built[kGammaCount] = {false, false, false, false};
  if (gamma < 0 || gamma >= kGammaCount) return nullptr;
  std::lock_guard<std::mutex> lock(mutex);
  if (!built[gamma]) {
    if (!build_half_table(gamma, &tables[gamma])) return nullptr;
    built[gamma] = true;
  }
  return &tables[gamma];
}
}  // namespace
}  // extern "C++"

int bt709hip_half_thresholds(int gamma, float *thresholds, int capacity) {
  const HalfTable *t = host_half_table(gamma);
  if (t == nullptr || capacity < 0 || (thresholds == nullptr && capacity > 0)) return BT709HIP_ERR_INVALID_ARG;
  if (t->split > 1.0f) return 0;
  const int n = static_cast<int>(t->thresholds.size());
  for (int i = 0; i < n && i < capacity; ++i) thresholds[i] = t->thresholds[static_cast<size_t>(i)];
  return n;
}

int bt709hip_half_lookup(int gamma, float x, int candidate_offset, int *table_entries) {
  const HalfTable *t = host_half_table(gamma);
  if (t == nullptr || !(x >= 0.0f && x <= 1.0f) || candidate_offset < -1 || candidate_offset > 0)
    return BT709HIP_ERR_INVALID_ARG;
  if (table_entries) *table_entries = t->split > 1.0f ? 0 : static_cast<int>(t->thresholds.size());
  const int low = float_to_half(x * t->low_scale);
  if (t->split > 1.0f || x < t->split) return low;
  size_t real = t->thresholds.size();
  while (real > 0 && t->thresholds[real - 1] == std::numeric_limits<float>::infinity()) --real;
  const int h_min = static_cast<int>(t->h_min), h_max = h_min + static_cast<int>(real) - 1;
  int h0 = static_cast<int>(float_to_half(curve_to_linear(gamma, x))) + candidate_offset;
  h0 = h0 < h_min - 1 ? h_min - 1 : (h0 > h_max ? h_max : h0);  // x >= split: H(x) >= h_min
  // T[h0 + 1]; T[h_max + 1] = +inf (the device copy's guard entry): the top code is never exceeded
  const size_t above = static_cast<size_t>(h0 + 1 - h_min);
  const float edge = above < real ? t->thresholds[above] : std::numeric_limits<float>::infinity();
  return h0 + (x >= edge ? 1 : 0);
}

int bt709hip_matrix_constants(float c[8]) {
  if (c == nullptr) return BT709HIP_ERR_INVALID_ARG;
  c[0] = kInv255;
  c[1] = kMY;
  c[2] = kMCrR;
  c[3] = kMCbG;
  c[4] = kMCrG;
  c[5] = kMCbB;
  c[6] = 16.0f;
  c[7] = 128.0f;
  return BT709HIP_OK;
}

}  // extern "C"
