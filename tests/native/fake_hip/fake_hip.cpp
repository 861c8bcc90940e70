// TESTS ONLY -- the fake HIP runtime behind tests/native/fake_hip/hip/hip_runtime.h (read its header comment first), plus
// stand-ins for the kernel launchers the shim calls (bt709_kernels.h): a "launch" validates nothing and computes nothing, it
// logs itself on its stream and advances a fake device clock.  Real threads, real mutexes: a stream is a FIFO with a worker
// thread, so ASan / UBSan / TSan see the shim's own locking and lifetimes at work.
#include "fake_hip.h"

#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../metalbt709decoder_amd/csrc/bt709_kernels.h"

namespace {

constexpr size_t kRealBytes = 16u << 20;  // allocations up to this are real memory; larger ones are address reservations
constexpr int kMaxDevices = 16;

struct Device {
  std::mutex m;
  uint64_t total = 288ull * 1000 * 1000 * 1000, used = 0;
  std::map<uintptr_t, size_t> allocs;               // every device allocation
  std::map<uintptr_t, size_t> reserved;             // the ones that are reservations only
  std::map<uintptr_t, std::pair<size_t, double>> rates;  // output ranges with a streaming rate of their own
  std::atomic<uint64_t> clock_ns{1000};
};

struct Runtime {
  std::mutex m;
  int device_count = 2;
  Device dev[kMaxDevices];
  std::vector<fake_hip_op> log;
  std::atomic<uint64_t> ops{0};
  std::atomic<int64_t> fail_malloc{0}, fail_launch{0};
  std::atomic<uint64_t> host_allocs{0}, streams{0}, events{0};
  std::vector<double> order_rates;
  size_t order_next = 0;
  std::map<const char *, size_t> pinned;  // hipHostMalloc'ed ranges (under m): hipPointerGetAttributes tells them from pageable memory
};
Runtime &rt() {
  static Runtime *r = new Runtime();  // never destroyed: worker threads may outlive main's statics
  return *r;
}

thread_local int tl_device = 0;
thread_local hipError_t tl_last = hipSuccess;

hipError_t fail(hipError_t e) {
  tl_last = e;
  return e;
}

}  // namespace

struct fakeGraph {
  std::vector<std::function<void()>> tasks;
  std::vector<fake_hip_op> ops;
};

struct fakeEvent {
  std::mutex m;
  std::condition_variable cv;
  uint64_t generation = 0, completed = 0;  // records issued / records whose turn in the stream has come
  uint64_t stamp_ns = 0;
  bool ever = false;
};

struct fakeStream {
  int device = 0;
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool busy = false, stop = false;
  fakeGraph *capturing = nullptr;
  std::thread worker;

  explicit fakeStream(int d) : device(d) {
    worker = std::thread([this] { run(); });
  }
  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [this] { return stop || !q.empty(); });
      if (q.empty()) {
        if (stop) return;
        continue;
      }
      std::function<void()> f = std::move(q.front());
      q.pop_front();
      busy = true;
      lk.unlock();
      f();
      lk.lock();
      busy = false;
      if (q.empty()) idle.notify_all();
    }
  }
  // `op` is logged at ISSUE time (stream order = issue order); the task runs on the worker (or joins the graph being recorded)
  void enqueue(fake_hip_op op, std::function<void()> f) {
    op.stream = this;
    op.device = device;
    {
      std::lock_guard<std::mutex> lk(m);
      if (capturing != nullptr) {
        capturing->tasks.push_back(std::move(f));
        capturing->ops.push_back(op);
        return;
      }
    }
    {
      Runtime &r = rt();
      std::lock_guard<std::mutex> lk(r.m);
      op.seq = r.ops.fetch_add(1);
      if (r.log.size() < (1u << 20)) r.log.push_back(op);
    }
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [this] { return q.empty() && !busy; });
  }
  ~fakeStream() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (worker.joinable()) worker.join();
  }
};

namespace {

fake_hip_op make_op(const char *name, int frames = 0, const void *in = nullptr, const void *out = nullptr) {
  fake_hip_op op;
  std::memset(&op, 0, sizeof op);
  std::snprintf(op.op, sizeof op.op, "%s", name);
  op.frames = frames;
  op.first_in = in;
  op.first_out = out;
  return op;
}

bool valid_device(int d) { return d >= 0 && d < rt().device_count; }

// is [p, p + n) inside a reservation (memory that must never be touched)?
bool reserved(const void *p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  Runtime &r = rt();
  for (int d = 0; d < r.device_count; ++d) {
    Device &dv = r.dev[d];
    std::lock_guard<std::mutex> lk(dv.m);
    auto it = dv.reserved.upper_bound(a);
    if (it != dv.reserved.begin()) {
      --it;
      if (a < it->first + it->second) return true;
    }
  }
  return false;
}

double rate_of(int device, const void *out) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(out);
  Device &dv = rt().dev[device];
  std::lock_guard<std::mutex> lk(dv.m);
  auto it = dv.rates.upper_bound(a);
  if (it != dv.rates.begin()) {
    --it;
    if (a < it->first + it->second.first) return it->second.second;
  }
  return 6000.0;
}

// a kernel launch: the log entry, the injected failure, the tick of the device clock when its turn comes
const char *launch(hipStream_t s, const char *name, int frames, const void *in, const void *out, double bytes) {
  char op[48];
  std::snprintf(op, sizeof op, "kernel:%s", name);
  Runtime &r = rt();
  int64_t f = r.fail_launch.load();
  if (f > 0 && r.fail_launch.fetch_sub(1) == 1) {
    tl_last = hipErrorInvalidValue;  // what hipGetLastError() reports after a refused launch
    return name;
  }
  if (s == nullptr) {
    tl_last = hipErrorInvalidResourceHandle;
    return name;
  }
  const int device = s->device;
  const double ns = bytes / rate_of(device, out);  // bytes / (GB/s) = ns
  s->enqueue(make_op(op, frames, in, out), [device, ns] {
    rt().dev[device].clock_ns.fetch_add(static_cast<uint64_t>(ns) + 1);
    std::this_thread::yield();
  });
  return name;
}

}  // namespace

// ------------------------------------------------------------------ the HIP API subset

hipError_t hipGetDeviceCount(int *count) {
  if (count == nullptr) return fail(hipErrorInvalidValue);
  *count = rt().device_count;
  return rt().device_count > 0 ? hipSuccess : fail(hipErrorNoDevice);
}
hipError_t hipSetDevice(int device) {
  if (!valid_device(device)) return fail(hipErrorInvalidDevice);
  tl_device = device;
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int device) {
  if (p == nullptr || !valid_device(device)) return fail(hipErrorInvalidDevice);
  std::memset(p, 0, sizeof *p);
  std::snprintf(p->name, sizeof p->name, "Fake MI355X #%d", device);
  std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:fake");
  p->totalGlobalMem = rt().dev[device].total;
  p->sharedMemPerBlock = 160 << 10;
  p->warpSize = 64;
  p->clockRate = 2400000;
  p->memoryClockRate = 2000000;
  p->memoryBusWidth = 8192;
  p->l2CacheSize = 4 << 20;
  p->multiProcessorCount = 256;
  p->pciDomainID = 0;
  p->pciBusID = 0x10 + device;
  p->pciDeviceID = 0;
  return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char *s, int len, int device) {
  if (s == nullptr || len < 13 || !valid_device(device)) return fail(hipErrorInvalidValue);
  std::snprintf(s, static_cast<size_t>(len), "0000:%02x:00.0", 0x10 + device);
  return hipSuccess;
}
hipError_t hipDeviceGetUuid(hipUUID *u, int device) {
  if (u == nullptr || !valid_device(device)) return fail(hipErrorInvalidValue);
  for (int i = 0; i < 16; ++i) u->bytes[i] = static_cast<char>(0xA0 + device + i);
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) {
  if (least) *least = 1;
  if (greatest) *greatest = -1;
  return hipSuccess;
}
hipError_t hipGetLastError(void) {
  const hipError_t e = tl_last;
  tl_last = hipSuccess;
  return e;
}
const char *hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNoDevice: return "no device";
    default: return "fake HIP error";
  }
}

hipError_t hipMalloc(void **ptr, size_t bytes) {
  if (ptr == nullptr) return fail(hipErrorInvalidValue);
  *ptr = nullptr;
  Runtime &r = rt();
  int64_t f = r.fail_malloc.load();
  if (f > 0 && r.fail_malloc.fetch_sub(1) == 1) return fail(hipErrorOutOfMemory);
  Device &dv = r.dev[tl_device];
  std::lock_guard<std::mutex> lk(dv.m);
  if (dv.used + bytes > dv.total) return fail(hipErrorOutOfMemory);
  void *p = nullptr;
  if (bytes <= kRealBytes) {
    p = std::malloc(bytes ? bytes : 1);
    if (p == nullptr) return fail(hipErrorOutOfMemory);
  } else {
    p = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return fail(hipErrorOutOfMemory);
    dv.reserved[reinterpret_cast<uintptr_t>(p)] = bytes;
    if (bytes >= (64u << 20)) {
      std::lock_guard<std::mutex> lk2(r.m);
      if (!r.order_rates.empty()) dv.rates[reinterpret_cast<uintptr_t>(p)] = {bytes, r.order_rates[r.order_next++ % r.order_rates.size()]};
    }
  }
  dv.allocs[reinterpret_cast<uintptr_t>(p)] = bytes;
  dv.used += bytes;
  *ptr = p;
  return hipSuccess;
}
hipError_t hipFree(void *ptr) {
  if (ptr == nullptr) return hipSuccess;
  Runtime &r = rt();
  for (int d = 0; d < r.device_count; ++d) {
    Device &dv = r.dev[d];
    std::lock_guard<std::mutex> lk(dv.m);
    auto it = dv.allocs.find(reinterpret_cast<uintptr_t>(ptr));
    if (it == dv.allocs.end()) continue;
    const size_t bytes = it->second;
    dv.allocs.erase(it);
    dv.used -= bytes;
    dv.rates.erase(reinterpret_cast<uintptr_t>(ptr));
    auto rv = dv.reserved.find(reinterpret_cast<uintptr_t>(ptr));
    if (rv != dv.reserved.end()) {
      munmap(ptr, bytes);
      dv.reserved.erase(rv);
    } else {
      std::free(ptr);
    }
    return hipSuccess;
  }
  std::fprintf(stderr, "fake_hip: hipFree of a pointer that is not a live device allocation: %p\n", ptr);
  std::abort();  // a double free or a wild pointer in the shim: fail the test loudly
}
hipError_t hipHostMalloc(void **ptr, size_t bytes, unsigned) {
  if (ptr == nullptr) return fail(hipErrorInvalidValue);
  *ptr = std::malloc(bytes ? bytes : 1);
  if (*ptr == nullptr) return fail(hipErrorOutOfMemory);
  rt().host_allocs.fetch_add(1);
  {
    std::lock_guard<std::mutex> lk(rt().m);
    rt().pinned[static_cast<const char *>(*ptr)] = bytes ? bytes : 1;
  }
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *attributes, const void *ptr) {
  if (attributes == nullptr) return fail(hipErrorInvalidValue);
  attributes->type = hipMemoryTypeUnregistered;
  std::lock_guard<std::mutex> lk(rt().m);
  auto it = rt().pinned.upper_bound(static_cast<const char *>(ptr));
  if (it != rt().pinned.begin()) {
    --it;
    if (static_cast<const char *>(ptr) < it->first + it->second) attributes->type = hipMemoryTypeHost;
  }
  return hipSuccess;
}
hipError_t hipHostFree(void *ptr) {
  if (ptr == nullptr) return hipSuccess;
  {
    std::lock_guard<std::mutex> lk(rt().m);
    rt().pinned.erase(static_cast<const char *>(ptr));
  }
  std::free(ptr);
  rt().host_allocs.fetch_sub(1);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes) {
  Device &dv = rt().dev[tl_device];
  std::lock_guard<std::mutex> lk(dv.m);
  if (free_bytes) *free_bytes = dv.total - dv.used;
  if (total_bytes) *total_bytes = dv.total;
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) {
  if (dst == nullptr || src == nullptr) return fail(hipErrorInvalidValue);
  if (!reserved(dst) && !reserved(src)) std::memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  if (dst == nullptr || src == nullptr || s == nullptr) return fail(hipErrorInvalidValue);
  const bool real = !reserved(dst) && !reserved(src);
  s->enqueue(make_op("memcpy", 0, src, dst), [=] {
    if (real) std::memcpy(dst, src, bytes);
  });
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                            hipStream_t s) {
  if (dst == nullptr || src == nullptr || s == nullptr || dpitch < width || spitch < width) return fail(hipErrorInvalidValue);
  const bool real = !reserved(dst) && !reserved(src);
  s->enqueue(make_op("memcpy2d", 0, src, dst), [=] {
    if (real)
      for (size_t r = 0; r < height; ++r) std::memcpy(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width);
  });
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s) {
  if (dst == nullptr || s == nullptr) return fail(hipErrorInvalidValue);
  const bool real = !reserved(dst);
  s->enqueue(make_op("memset", 0, nullptr, dst), [=] {
    if (real) std::memset(dst, value, bytes);
  });
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *stream, unsigned) {
  if (stream == nullptr) return fail(hipErrorInvalidValue);
  *stream = new fakeStream(tl_device);
  rt().streams.fetch_add(1);
  return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t *stream, unsigned flags, int) { return hipStreamCreateWithFlags(stream, flags); }
hipError_t hipStreamDestroy(hipStream_t s) {
  if (s == nullptr) return fail(hipErrorInvalidResourceHandle);
  s->drain();
  delete s;
  rt().streams.fetch_sub(1);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  if (s == nullptr) return fail(hipErrorInvalidResourceHandle);
  s->drain();
  return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *status) {
  if (s == nullptr || status == nullptr) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> lk(s->m);
  *status = s->capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  if (s == nullptr) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> lk(s->m);
  if (s->capturing) return fail(hipErrorStreamCaptureUnsupported);
  s->capturing = new fakeGraph();
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *graph) {
  if (s == nullptr || graph == nullptr) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> lk(s->m);
  if (!s->capturing) return fail(hipErrorStreamCaptureUnsupported);
  *graph = s->capturing;
  s->capturing = nullptr;
  return hipSuccess;
}

hipError_t hipEventCreate(hipEvent_t *event) {
  if (event == nullptr) return fail(hipErrorInvalidValue);
  *event = new fakeEvent();
  rt().events.fetch_add(1);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned) { return hipEventCreate(event); }

hipError_t hipEventDestroy(hipEvent_t e) {
  if (e == nullptr) return fail(hipErrorInvalidResourceHandle);
  delete e;
  rt().events.fetch_sub(1);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (e == nullptr || s == nullptr) return fail(hipErrorInvalidResourceHandle);
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lk(e->m);
    gen = ++e->generation;
    e->ever = true;
  }
  const int device = s->device;
  s->enqueue(make_op("event_record"), [e, gen, device] {
    std::lock_guard<std::mutex> lk(e->m);
    e->stamp_ns = rt().dev[device].clock_ns.load();
    e->completed = gen;
    e->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  if (e == nullptr) return fail(hipErrorInvalidResourceHandle);
  std::unique_lock<std::mutex> lk(e->m);
  const uint64_t gen = e->generation;
  e->cv.wait(lk, [&] { return e->completed >= gen; });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  if (e == nullptr || s == nullptr) return fail(hipErrorInvalidResourceHandle);
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lk(e->m);
    gen = e->generation;  // the record issued last, as hipStreamWaitEvent captures it
  }
  s->enqueue(make_op("wait_event"), [e, gen] {
    std::unique_lock<std::mutex> lk(e->m);
    e->cv.wait(lk, [&] { return e->completed >= gen; });
  });
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  if (ms == nullptr || a == nullptr || b == nullptr) return fail(hipErrorInvalidValue);
  std::scoped_lock lk(a->m, b->m);
  if (!a->ever || !b->ever || a->completed < a->generation || b->completed < b->generation) return fail(hipErrorNotReady);
  *ms = static_cast<float>((static_cast<double>(b->stamp_ns) - static_cast<double>(a->stamp_ns)) * 1e-6);
  return hipSuccess;
}

hipError_t hipGraphInstantiate(hipGraphExec_t *exec, hipGraph_t graph, hipGraphNode_t *, char *, size_t) {
  if (exec == nullptr || graph == nullptr) return fail(hipErrorInvalidValue);
  *exec = new fakeGraph(*graph);
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) {
  delete g;
  return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t g) {
  delete g;
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t s) {
  if (g == nullptr || s == nullptr) return fail(hipErrorInvalidValue);
  s->enqueue(make_op("graph_launch", static_cast<int>(g->tasks.size())), [] {});
  for (size_t i = 0; i < g->tasks.size(); ++i) s->enqueue(g->ops[i], g->tasks[i]);
  return hipSuccess;
}

// ------------------------------------------------------------------ the kernel launchers the shim links against

namespace bt709 {

LaunchShape &last_launch_shape() {
  static thread_local LaunchShape shape = {};
  return shape;
}

static double frame_bytes(const DecodeParams &p, double out_px_bytes) {
  return static_cast<double>(p.width) * p.height * 1.5 + static_cast<double>(p.out_width ? p.out_width : p.width) * (p.out_height ? p.out_height : p.height) * out_px_bytes;
}

const char *launch_decode(const DecodeParams &p, int frames, int variant, bool has_alpha, bool quantiser, bool nontemporal, int xcd_bands,
                          uint32_t grid_x, uint32_t block_threads, hipStream_t stream) {
  LaunchShape &shape = last_launch_shape();
  if (shape.launches++ == 0) {
    shape.grid[0] = grid_x, shape.grid[1] = p.height / 2, shape.grid[2] = static_cast<uint32_t>(frames);
    shape.block[0] = block_threads, shape.block[1] = shape.block[2] = 1;
    shape.xcd_bands = xcd_bands && frames >= kXcdBandMinFrames && frames % 8 == 0 ? xcd_bands : 0;
  }
  (void)quantiser, (void)nontemporal;
  const char *name = variant == kVariantQuads ? (has_alpha ? "decode_nv12_quads<alpha>" : "decode_nv12_quads<nt>") : "decode_nv12_blocks";
  return launch(stream, name, frames, p.frames[0].y, p.frames[0].out, frames * frame_bytes(p, 4.0));
}
const char *launch_decode_rgba16f(const DecodeParams &p, const HalfParams &, int frames, bool, uint32_t, uint32_t, uint32_t, bool, hipStream_t stream) {
  return launch(stream, "decode_nv12_rgba16f", frames, p.frames[0].y, p.frames[0].out, frames * frame_bytes(p, 8.0));
}
const char *launch_unconvert(const DecodeParams &, const UnconvertBatch &b, size_t, size_t, uint32_t width, uint32_t height, bool, bool,
                             hipStream_t stream) {
  return launch(stream, "unconvert_packed444", b.count, b.in[0], b.out[0], 8.0 * width * height * b.count);
}
const char *launch_decode_half(const DecodeParams &p, int frames, bool, bool, bool, uint32_t, uint32_t, hipStream_t stream) {
  return launch(stream, "decode_nv12_half", frames, p.frames[0].y, p.frames[0].out, frames * frame_bytes(p, 4.0));
}
const char *launch_decode_half_rep(const DecodeParams &p, int frames, bool, bool, uint32_t, uint32_t, hipStream_t stream) {
  return launch(stream, "decode_nv12_half_rep", frames, p.frames[0].y, p.frames[0].out, frames * frame_bytes(p, 4.0));
}
const char *launch_decode_scaled(const DecodeParams &p, int frames, bool, uint32_t, uint32_t, hipStream_t stream) {
  return launch(stream, "decode_nv12_scaled", frames, p.frames[0].y, p.frames[0].out, frames * frame_bytes(p, 4.0));
}
const char *launch_render_scaled(const RenderParams &p, int frames, bool, uint32_t, hipStream_t stream) {
  return launch(stream, "render_scaled", frames, p.in, p.out, frames * 4.0 * (static_cast<double>(p.width) * p.height + static_cast<double>(p.out_width) * p.out_height));
}
const char *launch_encode(const EncodeParams &p, int frames, bool, bool, hipStream_t stream) {
  return launch(stream, "encode_bgra_nv12", frames, p.frames[0].bgra, p.frames[0].y, frames * 5.5 * p.width * p.height);
}
const char *launch_planes(const PlaneParams &p, bool interleave, hipStream_t stream) {
  return launch(stream, interleave ? "interleave_cbcr" : "deinterleave_cbcr", 1, p.u, p.cbcr, 4.0 * p.chroma_width * p.chroma_height);
}
const char *launch_copy_probe(void *dst, const void *src, size_t bytes, hipStream_t stream) {
  return launch(stream, "copy_probe", 0, src, dst, 2.0 * static_cast<double>(bytes));
}
hipError_t prepare_kernels() { return hipSuccess; }
hipError_t prepare_rescale_kernels() { return hipSuccess; }
hipError_t prepare_encode_kernels() { return hipSuccess; }
hipError_t prepare_rgba16f_kernels() { return hipSuccess; }

}  // namespace bt709

// ------------------------------------------------------------------ what the tests ask

extern "C" {

void fake_hip_reset(void) {
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.m);
  r.log.clear();
  r.ops.store(0);
  r.fail_malloc.store(0);
  r.fail_launch.store(0);
  r.order_rates.clear();
  r.order_next = 0;
}
void fake_hip_set_device_count(int n) { rt().device_count = n < 0 ? 0 : (n > kMaxDevices ? kMaxDevices : n); }
void fake_hip_set_device_memory(uint64_t total) {
  for (Device &d : rt().dev) {
    std::lock_guard<std::mutex> lk(d.m);
    d.total = total;
  }
}
uint64_t fake_hip_log_size(void) { return rt().ops.load(); }
int fake_hip_log_get(uint64_t index, fake_hip_op *out) {
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.m);
  if (out == nullptr || index >= r.log.size()) return -1;
  *out = r.log[index];
  return 0;
}
uint64_t fake_hip_allocated(int device) {
  if (!valid_device(device)) return 0;
  Device &d = rt().dev[device];
  std::lock_guard<std::mutex> lk(d.m);
  return d.used;
}
uint64_t fake_hip_allocations(int device) {
  if (!valid_device(device)) return 0;
  Device &d = rt().dev[device];
  std::lock_guard<std::mutex> lk(d.m);
  return d.allocs.size();
}
uint64_t fake_hip_host_allocations(void) { return rt().host_allocs.load(); }
uint64_t fake_hip_live_streams(void) { return rt().streams.load(); }
uint64_t fake_hip_live_events(void) { return rt().events.load(); }
void fake_hip_fail_malloc_at(int64_t nth) { rt().fail_malloc.store(nth); }
void fake_hip_fail_launch_at(int64_t nth) { rt().fail_launch.store(nth); }
void fake_hip_set_output_rate(const void *ptr, uint64_t bytes, double GBps) {
  Device &d = rt().dev[tl_device];
  std::lock_guard<std::mutex> lk(d.m);
  d.rates[reinterpret_cast<uintptr_t>(ptr)] = {static_cast<size_t>(bytes), GBps};
}
void fake_hip_set_rate_by_allocation_order(const double *GBps, int n) {
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.m);
  r.order_rates.assign(GBps, GBps + (n > 0 ? n : 0));
  r.order_next = 0;
}

}  // extern "C"
