// C-ABI shim, core (include/bt709hip.h): the thin layer that replaces MetalRenderContext's device / queue / texture plumbing
// (Renderer/MetalRenderContext.h:17-105) with hipSetDevice / HIP streams / hipMalloc / hipMemcpy2DAsync -- contexts, streams,
// events, recorded command buffers (graphs), device and pinned memory, uploads and read-backs, status strings.
//
// There is no CPU fallback anywhere in the shim: without a HIP device every entry point that needs one fails with
// BT709HIP_ERR_NO_DEVICE / BT709HIP_ERR_HIP.
#include "shim_internal.h"

namespace bt709shim __attribute__((visibility("hidden"))) {

// This thread's last failing HIP call and last launched kernel.  Touched through the functions below ONLY: an `extern
// thread_local` read from another translation unit goes through a TLS wrapper that tests a weak, hidden init symbol, and in
// this -fPIC library clang resolves that symbol to the load address instead of null -- the first kernel launch of round 6's
// split shim jumped there (CPU tests cannot see it: nothing launches without a GPU).
namespace {
thread_local hipError_t tl_hip_error = hipSuccess;
thread_local const char *tl_kernel_name = "";
}  // namespace

void set_kernel_name(const char *name) { tl_kernel_name = name; }
const char *kernel_name() { return tl_kernel_name; }
hipError_t last_hip_error() { return tl_hip_error; }

int hip_fail(hipError_t e) {
  tl_hip_error = e;
  // The runtime keeps the error as this thread's "last error" until somebody reads it, and every launch here ends in
  // hipGetLastError(): without this, a failed allocation (reported to its caller, as it should be) would also fail the NEXT
  // decode of the thread with a stale out-of-memory (round 4: found by a test that asks for a ring the device cannot hold).
  (void)hipGetLastError();
  return BT709HIP_ERR_HIP;
}

int bind(const bt709hip_context *ctx) {
  if (ctx == nullptr) return BT709HIP_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
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

int finish_launch(hipStream_t s, int wait_until_completed) {
  HIP_TRY(hipGetLastError());
  if (wait_until_completed) HIP_TRY(hipStreamSynchronize(s));  // .m:486-489
  return BT709HIP_OK;
}

}  // namespace bt709shim

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

// Waits for what `s` holds up to HERE -- the copy just enqueued -- through an event of its own, not hipStreamSynchronize: work
// another thread enqueues on the stream behind the copy is not waited for (round 5's advisor).
int wait_for_copy(hipStream_t s) {
  hipEvent_t e = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t rc = hipEventRecord(e, s);
  if (rc == hipSuccess) rc = hipEventSynchronize(e);
  (void)hipEventDestroy(e);
  return rc == hipSuccess ? BT709HIP_OK : hip_fail(rc);
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
  if (!capturing(s) && pageable(src_host)) return wait_for_copy(s);
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
  if (!capturing(s) && pageable(dst_host)) return wait_for_copy(s);  // as above: a pageable target is filled on return
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

}  // extern "C"
