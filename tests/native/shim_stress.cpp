// TESTS ONLY -- the host shim (csrc/shim_*.cpp, bt709_ring.cpp) driven hard on the fake HIP runtime (tests/native/fake_hip/),
// built with -fsanitize=address,undefined or -fsanitize=thread (tools/sanitize.sh, tests/test_fake_hip.py).  No kernel runs and no
// pixel is checked here -- parity is the GPU tests' job -- this is about the shim's own state: mutexes, per-stream queues,
// thread-locals, lifetimes, error paths.
//
//   coalescing     two threads, each with its own stream, submit one-frame calls to ONE decoder while a third flips options
//                  and flushes; every frame submitted must reach its stream exactly once, in submission order per stream
//   two decoders   a coalescing decoder's queued frames go out before a later call of ANOTHER decoder on the same stream
//   age limit      a queue older than BT709HIP_OPT_COALESCE_MAX_AGE_US is issued by a call on another stream
//   pool           acquire / submit / wait / release churn, a failing submit, depth-3 pipeline
//   sharder        ticket window over 4 lanes on 2 devices, cancel, refused frames
//   ring           hunt over slabs with scripted rates (the fast one must win), allocation refused at candidate k, byte and
//                  time budgets, a launch that fails inside a probe; nothing leaks on any path
//   ring set       lanes on 2 devices, one launch per lane per step
//   destroy        a decoder destroyed with frames queued; contexts, streams and events all accounted for at exit
//   fuzz           30 000 calls of the decode / rescale / encode / plane entry points with random geometry, pitches, alignments,
//                  tags, formats, NULLs and counts: no crash, no out-of-bounds read of a descriptor array, always a status of the enum
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bt709hip_ext.h"
#include "fake_hip/fake_hip.h"

static int failures = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                           \
    }                                                                       \
  } while (0)
#define OK(expr) CHECK((expr) == BT709HIP_OK)

static bt709hip_frame frame_at(uint8_t *base, int w, int h, int transfer = BT709HIP_TRANSFER_ITU_R_709_2) {
  bt709hip_frame f;
  std::memset(&f, 0, sizeof f);
  f.y = base, f.y_stride = static_cast<size_t>(w);
  f.cbcr = base + static_cast<size_t>(w) * h, f.cbcr_stride = static_cast<size_t>(w);
  f.width = w, f.height = h, f.matrix = BT709HIP_MATRIX_ITU_R_709_2, f.transfer = transfer;
  return f;
}
static bt709hip_surface surface_at(uint8_t *base, int w, int h) {
  bt709hip_surface s;
  std::memset(&s, 0, sizeof s);
  s.bgra = base, s.stride = static_cast<size_t>(w) * 4, s.width = w, s.height = h;
  return s;
}

// the decode kernels the log holds for `stream`, in issue order: (frames, first output pointer)
struct Launch {
  int frames;
  const void *out;
  uint64_t seq;
  std::string op;
};
static std::vector<Launch> launches_on(void *stream, const char *prefix = "kernel:decode") {
  std::vector<Launch> v;
  fake_hip_op op;
  for (uint64_t i = 0; fake_hip_log_get(i, &op) == 0; ++i)
    if (op.stream == stream && std::strncmp(op.op, prefix, std::strlen(prefix)) == 0) v.push_back({op.frames, op.first_out, op.seq, op.op});
  return v;
}

static void test_coalescing_threads() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(0, &ctx));
  bt709hip_decoder *dec = nullptr;
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_APPLE, 0, &dec));
  OK(bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, 8));
  const int w = 64, h = 16, per_thread = 400, ring = 16;
  void *d_in = nullptr, *d_out = nullptr;
  OK(bt709hip_malloc(ctx, static_cast<size_t>(ring) * w * h * 3 / 2 * 2, &d_in));
  OK(bt709hip_malloc(ctx, static_cast<size_t>(ring) * w * h * 4 * 2, &d_out));
  void *streams[2] = {nullptr, nullptr};
  OK(bt709hip_stream_create(ctx, &streams[0]));
  OK(bt709hip_stream_create(ctx, &streams[1]));
  std::atomic<bool> go{false}, done{false};
  std::atomic<int> submitted[2] = {{0}, {0}};
  auto feeder = [&](int t) {
    while (!go.load()) std::this_thread::yield();
    for (int i = 0; i < per_thread; ++i) {
      const int slot = t * ring + i % ring;
      bt709hip_frame f = frame_at(static_cast<uint8_t *>(d_in) + static_cast<size_t>(slot) * w * h * 3 / 2, w, h);
      bt709hip_surface s = surface_at(static_cast<uint8_t *>(d_out) + static_cast<size_t>(slot) * w * h * 4, w, h);
      if (bt709hip_decode(dec, &f, nullptr, &s, w, h, streams[t], 0) == BT709HIP_OK) submitted[t].fetch_add(1);
      else ++failures;
      if (i % 37 == 0) OK(bt709hip_decoder_flush(dec, streams[t]));
      if (i % 53 == 0) OK(bt709hip_stream_synchronize(ctx, streams[t]));
    }
  };
  std::thread a(feeder, 0), b(feeder, 1);
  std::thread meddler([&] {  // option toggling and whole-decoder flushes from a third thread
    while (!go.load()) std::this_thread::yield();
    int k = 0;
    while (!done.load()) {
      OK(bt709hip_decoder_flush_all(dec));
      OK(bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, (k++ % 3 == 0) ? 0 : 4 + k % 8));
      OK(bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE_MAX_AGE_US, k % 2 ? 50 : 0));
      int v = 0;
      OK(bt709hip_decoder_get_option(dec, BT709HIP_OPT_XCD_BANDS, &v));
      std::this_thread::yield();
    }
  });
  go.store(true);
  a.join();
  b.join();
  done.store(true);
  meddler.join();
  OK(bt709hip_decoder_flush_all(dec));
  for (int t = 0; t < 2; ++t) {
    OK(bt709hip_stream_synchronize(ctx, streams[t]));
    int frames = 0;
    for (const Launch &l : launches_on(streams[t])) frames += l.frames;
    CHECK(frames == submitted[t].load() && frames == per_thread);  // every frame exactly once
  }
  OK(bt709hip_decoder_destroy(dec));
  OK(bt709hip_stream_destroy(ctx, streams[0]));
  OK(bt709hip_stream_destroy(ctx, streams[1]));
  OK(bt709hip_free(ctx, d_in));
  OK(bt709hip_free(ctx, d_out));
  OK(bt709hip_context_destroy(ctx));
}

static void test_two_decoders_one_stream_and_age() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(0, &ctx));
  bt709hip_decoder *a = nullptr, *b = nullptr;
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_APPLE, 0, &a));
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_APPLE, 0, &b));
  OK(bt709hip_decoder_set_option(a, BT709HIP_OPT_COALESCE, 8));
  const int w = 64, h = 16;
  void *d_in = nullptr, *d_out = nullptr, *s = nullptr, *other = nullptr;
  OK(bt709hip_malloc(ctx, 8 * w * h * 3 / 2, &d_in));
  OK(bt709hip_malloc(ctx, 8 * w * h * 4, &d_out));
  OK(bt709hip_stream_create(ctx, &s));
  OK(bt709hip_stream_create(ctx, &other));
  uint8_t *in = static_cast<uint8_t *>(d_in), *out = static_cast<uint8_t *>(d_out);
  for (int second_coalesces = 0; second_coalesces < 2; ++second_coalesces) {
    fake_hip_reset();
    OK(bt709hip_decoder_set_option(b, BT709HIP_OPT_COALESCE, second_coalesces ? 8 : 0));
    for (int i = 0; i < 3; ++i) {  // A queues three frames writing surfaces 0..2
      bt709hip_frame f = frame_at(in + i * w * h * 3 / 2, w, h);
      bt709hip_surface o = surface_at(out + i * w * h * 4, w, h);
      OK(bt709hip_decode(a, &f, nullptr, &o, w, h, s, 0));
    }
    CHECK(launches_on(s).empty());
    bt709hip_frame f = frame_at(in + 5 * w * h * 3 / 2, w, h);  // B writes surface 0 from another input, same stream
    bt709hip_surface o = surface_at(out, w, h);
    OK(bt709hip_decode(b, &f, nullptr, &o, w, h, s, 0));
    OK(bt709hip_stream_synchronize(ctx, s));
    const std::vector<Launch> l = launches_on(s);
    CHECK(l.size() == 2 && l[0].frames == 3 && l[1].frames == 1);  // A's queue first, then B's frame
  }
  // age limit: frames queued on s; a call on ANOTHER stream issues them once they are old enough, not before
  fake_hip_reset();
  OK(bt709hip_decoder_set_option(a, BT709HIP_OPT_COALESCE_MAX_AGE_US, 2000));
  bt709hip_frame f = frame_at(in, w, h);
  bt709hip_surface o = surface_at(out, w, h);
  OK(bt709hip_decode(a, &f, nullptr, &o, w, h, s, 0));
  OK(bt709hip_stream_synchronize(ctx, other));
  CHECK(launches_on(s).empty());  // younger than 2 ms
  std::this_thread::sleep_for(std::chrono::milliseconds(5));
  OK(bt709hip_stream_synchronize(ctx, other));
  CHECK(launches_on(s).size() == 1);
  // ... and a decoder destroyed with frames queued issues them on its way out
  OK(bt709hip_decode(a, &f, nullptr, &o, w, h, s, 0));
  OK(bt709hip_decoder_destroy(a));
  CHECK(launches_on(s).size() == 2);
  OK(bt709hip_decoder_destroy(b));
  OK(bt709hip_stream_destroy(ctx, s));
  OK(bt709hip_stream_destroy(ctx, other));
  OK(bt709hip_free(ctx, d_in));
  OK(bt709hip_free(ctx, d_out));
  OK(bt709hip_context_destroy(ctx));
}

static void test_pool_and_sharder() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(1, &ctx));
  bt709hip_decoder *dec = nullptr;
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_SRGB, 1, &dec));  // alpha decoder: three planes per slot
  bt709hip_pool *pool = nullptr;
  OK(bt709hip_pool_create(dec, 64, 16, 3, &pool));
  for (int i = 0; i < 200; ++i) {
    int slot = -1;
    void *y = nullptr, *c = nullptr, *al = nullptr;
    size_t ys = 0, cs = 0, as = 0;
    OK(bt709hip_pool_acquire(pool, &slot, &y, &ys, &c, &cs));
    OK(bt709hip_pool_alpha_plane(pool, slot, &al, &as));
    std::memset(y, i, ys * 16), std::memset(c, i, cs * 8), std::memset(al, i, as * 16);  // pinned staging is writable, whole planes
    if (i % 11 == 0) {
      OK(bt709hip_pool_release(pool, slot));
      continue;
    }
    if (i == 100) fake_hip_fail_launch_at(1);  // the decode inside this submit is refused: the slot must come back
    const int rc = bt709hip_pool_submit(pool, slot);
    CHECK(i == 100 ? rc == BT709HIP_ERR_HIP : rc == BT709HIP_OK);
    if (i % 3 == 0) {
      const void *bgra = nullptr;
      size_t stride = 0;
      OK(bt709hip_pool_wait(pool, slot, &bgra, &stride));
      CHECK(bgra != nullptr && stride == 64 * 4);
    }
  }
  OK(bt709hip_pool_destroy(pool));
  OK(bt709hip_decoder_destroy(dec));
  OK(bt709hip_context_destroy(ctx));

  const int devices[4] = {0, 1, 0, 1};
  bt709hip_shard *sh = nullptr;
  OK(bt709hip_shard_create(devices, 4, BT709HIP_GAMMA_APPLE, 0, 64, 16, 2, &sh));
  CHECK(bt709hip_shard_lane_device(sh, 1) == 1 && bt709hip_shard_lanes(sh) == 4);
  std::vector<uint8_t> host(64 * 16 * 3 / 2, 77);
  bt709hip_frame hf = frame_at(host.data(), 64, 16);
  for (uint64_t i = 0; i < 64; ++i) {
    uint64_t t = ~0ull;
    OK(bt709hip_shard_submit(sh, &hf, nullptr, &t));
    CHECK(t == i);
    if (i >= 7) {
      const void *bgra = nullptr;
      size_t stride = 0;
      OK(bt709hip_shard_wait(sh, i - 7, &bgra, &stride));  // the oldest frame still held (4 lanes x depth 2)
      CHECK(bt709hip_shard_wait(sh, i + 1, &bgra, &stride) == BT709HIP_ERR_INVALID_ARG);
    }
  }
  bt709hip_frame bad = hf;
  bad.transfer = BT709HIP_TRANSFER_SRGB;
  uint64_t t = 0;
  CHECK(bt709hip_shard_submit(sh, &bad, nullptr, &t) == BT709HIP_ERR_TRANSFER);
  void *y = nullptr, *c = nullptr;
  size_t ys = 0, cs = 0;
  OK(bt709hip_shard_acquire(sh, &t, &y, &ys, &c, &cs, nullptr, nullptr));
  CHECK(bt709hip_shard_acquire(sh, &t, &y, &ys, &c, &cs, nullptr, nullptr) == BT709HIP_ERR_INVALID_ARG);
  OK(bt709hip_shard_cancel(sh));
  OK(bt709hip_shard_destroy(sh));
}

static void test_ring_hunts() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(0, &ctx));
  bt709hip_decoder *dec = nullptr;
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_APPLE, 0, &dec));
  OK(bt709hip_decoder_setup(dec));
  const uint64_t base_bytes = fake_hip_allocated(0), base_allocs = fake_hip_allocations(0);
  const int w = 3840, h = 2160, n = 64;  // 0.8 GB in + 2.1 GB out: address reservations on the fake device
  // scripted placement: large allocations get these rates in allocation order (3 inputs first, then outputs)
  const double rates[] = {6000, 6000, 6000, /* outputs */ 5900, 5950, 6600, 5900, 5950, 5920, 5900, 5950, 5900, 5900, 5900, 5900};
  fake_hip_set_rate_by_allocation_order(rates, static_cast<int>(sizeof rates / sizeof rates[0]));
  bt709hip_ring *ring = nullptr;
  bt709hip_ring_options roomy = {200ull << 30, 0, 0};
  OK(bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &roomy, &ring));
  bt709hip_ring_placement p;
  OK(bt709hip_ring_placement_info(ring, &p));
  CHECK(p.tries == 3 && p.in_candidates == 3 && p.out_candidates >= 6 && p.chosen_out == 2);  // the 6 600 GB/s slab wins
  CHECK(p.evicted == 0 && p.stopped_by == 0 && p.hunt_ms > 0.0f && p.peak_bytes <= p.budget_bytes);
  CHECK(fake_hip_allocations(0) == base_allocs + 2);  // the ring's two slabs, nothing else
  OK(bt709hip_ring_decode(ring, 0, n, nullptr, 1));
  OK(bt709hip_ring_destroy(ring));
  CHECK(fake_hip_allocated(0) == base_bytes);

  // frugal: the fast slab is found although only two outputs ever live together
  fake_hip_set_rate_by_allocation_order(rates + 2, static_cast<int>(sizeof rates / sizeof rates[0]) - 2);  // 1 input, then outputs
  bt709hip_ring_options frugal = {0, 0, 1};
  OK(bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &frugal, &ring));
  OK(bt709hip_ring_placement_info(ring, &p));
  CHECK(p.in_candidates == 1 && p.evicted == p.out_candidates - 2 && p.stopped_by == 1);
  CHECK(p.out_prescan_GBps[p.chosen_out] > 6500.0f && p.peak_bytes <= p.budget_bytes);
  const uint64_t frugal_budget = p.budget_bytes;
  OK(bt709hip_ring_destroy(ring));
  CHECK(fake_hip_allocated(0) == base_bytes);
  // round 6: that IS the default -- options NULL (bt709hip_ring_create) or a zeroed struct hunt within twice the ring
  fake_hip_set_rate_by_allocation_order(rates + 2, static_cast<int>(sizeof rates / sizeof rates[0]) - 2);
  OK(bt709hip_ring_create(dec, w, h, n, 0, 3, &ring));
  OK(bt709hip_ring_placement_info(ring, &p));
  CHECK(p.budget_bytes == frugal_budget && p.in_candidates == 1 && p.stopped_by == 1 && p.out_prescan_GBps[p.chosen_out] > 6500.0f);
  OK(bt709hip_ring_destroy(ring));
  CHECK(fake_hip_allocated(0) == base_bytes);
  // a probe that fails AFTER the budget has evicted (freed) the first output: the error comes back and nothing is freed twice
  // (round 5's advisor: the error path used to free outs[0] again; AddressSanitizer is the judge here)
  int failed_hunts = 0;
  for (int k : {20, 45, 90, 150}) {
    fake_hip_set_rate_by_allocation_order(rates + 2, static_cast<int>(sizeof rates / sizeof rates[0]) - 2);
    fake_hip_fail_launch_at(k);
    ring = nullptr;
    const int rc = bt709hip_ring_create(dec, w, h, n, 0, 3, &ring);
    fake_hip_fail_launch_at(0);
    CHECK((rc == BT709HIP_ERR_HIP && ring == nullptr) || (rc == BT709HIP_OK && ring != nullptr));
    failed_hunts += rc == BT709HIP_ERR_HIP;
    if (ring) OK(bt709hip_ring_destroy(ring));
    CHECK(fake_hip_allocated(0) == base_bytes);
  }

  // allocation refused at candidate k (k = 1: the ring's own input; 2: its first output; later: a candidate): clean error or a
  // smaller hunt, never a leak
  for (int k = 1; k <= 9; ++k) {
    fake_hip_set_rate_by_allocation_order(nullptr, 0);
    fake_hip_fail_malloc_at(k);
    ring = nullptr;
    const int rc = bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &roomy, &ring);
    if (k == 1 || k == 4) CHECK(rc == BT709HIP_ERR_HIP && ring == nullptr);  // no input at all / no output at all (3 inputs came first)
    else CHECK(rc == BT709HIP_OK && ring != nullptr);
    fake_hip_fail_malloc_at(0);
    if (ring) {
      OK(bt709hip_ring_placement_info(ring, &p));
      CHECK(p.in_candidates >= 1 && p.out_candidates >= 1);
      OK(bt709hip_ring_destroy(ring));
    }
    CHECK(fake_hip_allocated(0) == base_bytes && fake_hip_allocations(0) == base_allocs);
  }
  // a launch that fails inside a probe: the error comes back, nothing stays allocated, the next decode works
  for (int k : {1, 5, 40}) {
    fake_hip_fail_launch_at(k);
    ring = nullptr;
    const int rc = bt709hip_ring_create_ex(dec, w, h, n, 0, 2, &roomy, &ring);
    fake_hip_fail_launch_at(0);
    CHECK((rc == BT709HIP_ERR_HIP && ring == nullptr) || (rc == BT709HIP_OK && ring != nullptr));
    if (ring) OK(bt709hip_ring_destroy(ring));
    CHECK(fake_hip_allocated(0) == base_bytes);
  }
  // time budget; a byte budget too small for a second output; a device too small for the ring
  bt709hip_ring_options hurried = {0, 1, 0};
  OK(bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &hurried, &ring));
  OK(bt709hip_ring_placement_info(ring, &p));
  CHECK(p.stopped_by == 2 && p.out_candidates == 1);
  OK(bt709hip_ring_destroy(ring));
  bt709hip_ring_options tight = {3ull << 30, 0, 0};
  OK(bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &tight, &ring));
  OK(bt709hip_ring_placement_info(ring, &p));
  CHECK(p.tries == 1 && p.stopped_by == 1 && p.probes == 0);
  OK(bt709hip_ring_destroy(ring));
  CHECK(bt709hip_ring_create(dec, w, h, 60000, 0, 3, &ring) == BT709HIP_ERR_HIP && ring == nullptr);
  CHECK(fake_hip_allocated(0) == base_bytes);
  // a coalescing decoder: the hunt's probes are launches, not queue entries, and the option is back afterwards
  OK(bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, 32));
  OK(bt709hip_ring_create_ex(dec, w, h, 8, 0, 2, &roomy, &ring));
  OK(bt709hip_ring_placement_info(ring, &p));
  int v = 0;
  OK(bt709hip_decoder_get_option(dec, BT709HIP_OPT_COALESCE, &v));
  CHECK(p.probes >= 1 && p.first_GBps > 1000.0f && v == 32);
  OK(bt709hip_ring_destroy(ring));
  OK(bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, 0));
  // RGBA16Float render targets (options.format): 8-byte texels in the descriptors, the hunt on that launch; refused with
  // half_scale and for an unknown format, nothing allocated by the refusals
  bt709hip_ring_options f16 = {200ull << 30, 0, 0, BT709HIP_FORMAT_RGBA16F, 0};
  fake_hip_set_rate_by_allocation_order(rates, static_cast<int>(sizeof rates / sizeof rates[0]));
  OK(bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &f16, &ring));
  bt709hip_surface o;
  OK(bt709hip_ring_frame(ring, 5, nullptr, nullptr, &o));
  OK(bt709hip_ring_placement_info(ring, &p));
  CHECK(o.format == BT709HIP_FORMAT_RGBA16F && o.stride == static_cast<size_t>(w) * 8 && p.chosen_out == 2 && p.probes >= 3);
  OK(bt709hip_ring_decode(ring, 8, 16, nullptr, 1));
  OK(bt709hip_ring_destroy(ring));
  ring = nullptr;
  CHECK(bt709hip_ring_create_ex(dec, w, h, n, 1, 3, &f16, &ring) == BT709HIP_ERR_UNSUPPORTED && ring == nullptr);
  bt709hip_ring_options bad = {0, 0, 0, 7, 0};
  CHECK(bt709hip_ring_create_ex(dec, w, h, n, 0, 3, &bad, &ring) == BT709HIP_ERR_INVALID_ARG && ring == nullptr);
  OK(bt709hip_decoder_destroy(dec));
  OK(bt709hip_context_destroy(ctx));
}

static void test_ring_set() {
  fake_hip_reset();
  const int devices[3] = {0, 1, 1};
  bt709hip_ringset *set = nullptr;
  OK(bt709hip_ringset_create(devices, 3, BT709HIP_GAMMA_APPLE, 0, 1920, 1080, 16, 0, 2, nullptr, &set));
  CHECK(bt709hip_ringset_lanes(set) == 3);
  bt709hip_device_info info;
  OK(bt709hip_context_info(bt709hip_ringset_lane_context(set, 2), &info));
  CHECK(info.device_ordinal == 1 && std::strcmp(info.pci_bus_id, "0000:11:00.0") == 0 && std::strlen(info.uuid) == 32);
  for (int step = 0; step < 50; ++step) OK(bt709hip_ringset_decode(set, 0, 16, 0));
  OK(bt709hip_ringset_synchronize(set));
  OK(bt709hip_ringset_decode(set, 4, 8, 1));
  CHECK(bt709hip_ringset_decode(set, 10, 8, 0) == BT709HIP_ERR_INVALID_ARG);
  OK(bt709hip_ringset_destroy(set));
  const int bad[2] = {0, 7};
  CHECK(bt709hip_ringset_create(bad, 2, BT709HIP_GAMMA_APPLE, 0, 64, 16, 4, 0, 1, nullptr, &set) == BT709HIP_ERR_NO_DEVICE && set == nullptr);
}

static void test_graphs_and_misc() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(0, &ctx));
  bt709hip_decoder *dec = nullptr;
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_LINEAR, 0, &dec));
  OK(bt709hip_decoder_setup(dec));
  OK(bt709hip_decoder_prepare_format(dec, BT709HIP_FORMAT_RGBA16F));
  OK(bt709hip_render_scaled_prepare(ctx));
  OK(bt709hip_encoder_prepare(ctx, BT709HIP_GAMMA_SRGB, BT709HIP_GAMMA_APPLE));
  const int w = 64, h = 16;
  void *d_in = nullptr, *d_out = nullptr, *s = nullptr, *graph = nullptr;
  OK(bt709hip_malloc(ctx, 4 * w * h * 3 / 2, &d_in));
  OK(bt709hip_malloc(ctx, 4 * w * h * 8, &d_out));
  OK(bt709hip_stream_create_with_priority(ctx, -1, &s));
  bt709hip_frame f[4];
  bt709hip_surface o[4];
  for (int i = 0; i < 4; ++i) {
    f[i] = frame_at(static_cast<uint8_t *>(d_in) + i * w * h * 3 / 2, w, h, BT709HIP_TRANSFER_LINEAR);
    o[i] = surface_at(static_cast<uint8_t *>(d_out) + i * w * h * 4, w, h);
  }
  OK(bt709hip_graph_begin_capture(ctx, s));
  OK(bt709hip_decode_batch(dec, 4, f, nullptr, o, s, 0));
  OK(bt709hip_memset(ctx, d_out, 0, 64, s));
  OK(bt709hip_graph_end_capture(ctx, s, &graph));
  for (int i = 0; i < 10; ++i) OK(bt709hip_graph_launch(ctx, graph, s));
  OK(bt709hip_stream_synchronize(ctx, s));
  CHECK(launches_on(s).size() == 10);
  OK(bt709hip_graph_destroy(ctx, graph));
  void *e0 = nullptr, *e1 = nullptr;
  OK(bt709hip_event_create(ctx, &e0));
  OK(bt709hip_event_create(ctx, &e1));
  OK(bt709hip_event_record(ctx, e0, s));
  OK(bt709hip_decode_batch(dec, 4, f, nullptr, o, s, 1));
  OK(bt709hip_event_record(ctx, e1, s));
  OK(bt709hip_stream_wait_event(ctx, nullptr, e1));
  OK(bt709hip_event_synchronize(ctx, e1));
  float ms = -1.0f;
  OK(bt709hip_event_elapsed_ms(ctx, e0, e1, &ms));
  CHECK(ms > 0.0f);
  float rates[3];
  int chosen = -1;
  void *slab = nullptr;
  OK(bt709hip_malloc_streaming(ctx, 512u << 20, 3, &slab, rates, &chosen));
  CHECK(chosen >= 0 && chosen < 3 && rates[chosen] > 0.0f);
  OK(bt709hip_free(ctx, slab));
  CHECK(bt709hip_malloc(ctx, 1ull << 50, &slab) == BT709HIP_ERR_HIP && bt709hip_last_hip_error() != 0);
  OK(bt709hip_decode_batch(dec, 4, f, nullptr, o, s, 1));  // the failed allocation does not linger as the thread's last error
  OK(bt709hip_event_destroy(ctx, e0));
  OK(bt709hip_event_destroy(ctx, e1));
  OK(bt709hip_decoder_destroy(dec));
  OK(bt709hip_stream_destroy(ctx, s));
  OK(bt709hip_free(ctx, d_in));
  OK(bt709hip_free(ctx, d_out));
  OK(bt709hip_context_destroy(ctx));
}

// Host-buffer lifetime of bt709hip_upload / _download: from PAGEABLE memory the copy is complete on return -- the buffer is
// overwritten (upload) or read (download) straight away here, which under TSan is a reported race with the stream's worker
// if the call had not waited, and a wrong byte otherwise; from PINNED memory (bt709hip_host_alloc) the call does not wait.
static void test_host_buffer_lifetime() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(0, &ctx));
  void *s = nullptr, *dev = nullptr, *pinned = nullptr;
  OK(bt709hip_stream_create_with_priority(ctx, 0, &s));
  const size_t n = 1u << 20;
  OK(bt709hip_malloc(ctx, n, &dev));
  OK(bt709hip_host_alloc(ctx, n, &pinned));
  for (void *stream : {static_cast<void *>(nullptr), s}) {
    for (int round = 0; round < 20; ++round) {
      std::vector<uint8_t> src(n, static_cast<uint8_t>(round + 1));
      OK(bt709hip_upload(ctx, dev, 1024, src.data(), 1024, 1024, n / 1024, stream));
      std::memset(src.data(), 0xee, n);  // the caller's buffer is the caller's again
      std::vector<uint8_t> back(n, 0);
      OK(bt709hip_download(ctx, back.data(), 1024, dev, 1024, 1024, n / 1024, stream));
      CHECK(back[0] == round + 1 && back[n - 1] == round + 1 && back[n / 2] == round + 1);  // filled on return, with the bytes uploaded
    }
    // pinned memory: asynchronous as before -- the bytes are there after the stream has been waited for
    std::memset(pinned, 0x5a, n);
    OK(bt709hip_upload(ctx, dev, 1024, pinned, 1024, 1024, n / 1024, stream));
    OK(bt709hip_stream_synchronize(ctx, stream));
    std::memset(pinned, 0, n);
    OK(bt709hip_download(ctx, pinned, 1024, dev, 1024, 1024, n / 1024, stream));
    OK(bt709hip_stream_synchronize(ctx, stream));
    CHECK(static_cast<uint8_t *>(pinned)[n - 1] == 0x5a);
  }
  OK(bt709hip_host_free(ctx, pinned));
  OK(bt709hip_free(ctx, dev));
  OK(bt709hip_stream_destroy(ctx, s));
  OK(bt709hip_context_destroy(ctx));
}

// Argument fuzz: every decode / rescale / encode / plane entry point with random geometry, pitches, alignments, tags, formats,
// NULLs and counts.  On the fake runtime a launch touches nothing, so whatever the shim lets through is harmless here -- the point
// is the shim's OWN reads (descriptor arrays, alpha arrays, per-frame loops) under ASan / UBSan, and that every call returns a
// status of the enum.
static void test_argument_fuzz() {
  fake_hip_reset();
  bt709hip_context *ctx = nullptr;
  OK(bt709hip_context_create(0, &ctx));
  bt709hip_decoder *decs[3] = {nullptr, nullptr, nullptr};
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_APPLE, 0, &decs[0]));
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_LINEAR, 0, &decs[1]));
  OK(bt709hip_decoder_create(ctx, BT709HIP_GAMMA_SRGB, 1, &decs[2]));
  OK(bt709hip_decoder_set_option(decs[1], BT709HIP_OPT_COALESCE, 5));
  void *mem = nullptr, *s = nullptr;
  OK(bt709hip_malloc(ctx, 8u << 20, &mem));
  OK(bt709hip_stream_create(ctx, &s));
  uint8_t *base = static_cast<uint8_t *>(mem);
  uint32_t x = 0x709u;
  auto rnd = [&](uint32_t n) { x = x * 1664525u + 1013904223u; return (x >> 8) % n; };
  const int dims[] = {0, 1, 2, 3, 4, 6, 8, 16, 30, 64, 66, 250, 640, 1920};
  int statuses[16] = {0};
  for (int it = 0; it < 30000; ++it) {
    const int count = static_cast<int>(rnd(40)) - 2;  // -2 .. 37: negative, zero, beyond the pointer table
    const int w = dims[rnd(14)], h = dims[rnd(12)];
    bt709hip_frame f[40], a[40];
    bt709hip_surface o[40], in[40];
    const int n = count < 0 ? 0 : count;
    const bool uniform = rnd(2) == 0;
    for (int i = 0; i < n && i < 40; ++i) {
      const size_t off = uniform ? static_cast<size_t>(i) * 4096 : rnd(1 << 20);
      f[i] = frame_at(base + (rnd(8) ? off & ~size_t(15) : off | 1), w + static_cast<int>(rnd(3)) * 4, h, static_cast<int>(rnd(4)));
      f[i].width = w, f[i].height = h;
      if (rnd(16) == 0) f[i].y = nullptr;
      if (rnd(16) == 0) f[i].matrix = static_cast<int>(rnd(4));
      if (rnd(8) == 0) f[i].width = dims[rnd(14)];
      a[i] = f[i];
      a[i].transfer = rnd(4) ? BT709HIP_TRANSFER_LINEAR : BT709HIP_TRANSFER_SRGB;
      const int ow = rnd(4) ? w : dims[rnd(14)], oh = rnd(4) ? h : dims[rnd(12)];
      o[i] = surface_at(base + (4u << 20) + (off & ~size_t(rnd(4) ? 15 : 3)), ow, oh);
      o[i].format = rnd(6) == 0 ? static_cast<int>(rnd(3)) : 0;
      if (o[i].format == BT709HIP_FORMAT_RGBA16F) o[i].stride *= 2;
      if (rnd(10) == 0) o[i].stride = rnd(2) ? 0 : o[i].stride + 2;
      if (rnd(20) == 0) o[i].reserved = 1;
      in[i] = o[i];
      in[i].bgra = base + (off & ~size_t(15));
    }
    bt709hip_decoder *dec = rnd(12) ? decs[rnd(3)] : nullptr;
    const bt709hip_frame *fp = rnd(20) ? f : nullptr, *ap = rnd(2) ? a : nullptr;
    const bt709hip_surface *op = rnd(20) ? o : nullptr;
    void *st = rnd(3) ? s : nullptr;
    int rc = 0;
    switch (rnd(12)) {
      case 0: rc = bt709hip_decode_batch(dec, count, fp, ap, op, st, static_cast<int>(rnd(2))); break;
      case 1: rc = bt709hip_decode(dec, n ? fp : nullptr, n ? ap : nullptr, n ? op : nullptr, w, h, st, 0); break;
      case 2: rc = bt709hip_decode_half_batch(dec, count, fp, ap, op, st, 0); break;
      case 3: rc = bt709hip_decode_scaled_batch(dec, count, fp, ap, op, st, 0); break;
      case 4: rc = bt709hip_render_scaled_batch(ctx, count, rnd(20) ? in : nullptr, op, st, 0); break;
      case 5: rc = bt709hip_render_scaled(ctx, n ? in : nullptr, n ? op : nullptr, st, 0); break;
      case 6: rc = bt709hip_encode_batch(ctx, count, rnd(20) ? in : nullptr, fp, static_cast<int>(rnd(4)) - 1 + 1, static_cast<int>(rnd(4)), st, 0); break;
      case 7: rc = bt709hip_unconvert(dec, n ? f[0].y : nullptr, static_cast<size_t>(w) * 4 + rnd(3) * 2, w, h, n ? op : nullptr, st, 0); break;
      case 8: {
        const void *ptrs[40];
        for (int i = 0; i < n && i < 40; ++i) ptrs[i] = f[i].y;
        rc = bt709hip_unconvert_batch(dec, count, rnd(20) ? ptrs : nullptr, static_cast<size_t>(w) * 4, w, h, op, st, 0);
        break;
      }
      case 9: rc = bt709hip_interleave_cbcr(ctx, base, rnd(4) ? w : 1, base + 65536, w, base + 131072, 2 * static_cast<size_t>(w) + rnd(2), w, h, st, 0); break;
      case 10: rc = bt709hip_deinterleave_cbcr(ctx, base + 131072, 2 * static_cast<size_t>(w), rnd(8) ? base : nullptr, w, base + 65536, w, w, h, st, 0); break;
      default: rc = bt709hip_copy_probe(ctx, base + rnd(64), base + (1 << 20), rnd(4096) * (rnd(2) ? 16 : 1), st); break;
    }
    CHECK(rc <= 0 && rc >= BT709HIP_ERR_UNSUPPORTED);
    if (rc <= 0 && rc >= -15) ++statuses[-rc];
    if (it % 997 == 0) OK(bt709hip_stream_synchronize(ctx, st));
  }
  CHECK(statuses[0] > 1000 && statuses[-BT709HIP_ERR_INVALID_ARG] > 100 && statuses[-BT709HIP_ERR_SIZE_MISMATCH] > 100 &&
        statuses[-BT709HIP_ERR_STRIDE] > 100 && statuses[-BT709HIP_ERR_ODD_DIMENSIONS] > 100 && statuses[-BT709HIP_ERR_UNSUPPORTED] > 50);
  for (bt709hip_decoder *d : decs) OK(bt709hip_decoder_destroy(d));
  OK(bt709hip_stream_destroy(ctx, s));
  OK(bt709hip_free(ctx, mem));
  OK(bt709hip_context_destroy(ctx));
}

int main(int argc, char **argv) {
  fake_hip_set_device_count(2);
  const std::string only = argc > 1 ? argv[1] : "";
  struct {
    const char *name;
    void (*fn)();
  } tests[] = {{"coalescing", test_coalescing_threads}, {"two_decoders", test_two_decoders_one_stream_and_age}, {"pool_sharder", test_pool_and_sharder},
               {"ring", test_ring_hunts},               {"ring_set", test_ring_set},                             {"graphs", test_graphs_and_misc},
               {"host_buffers", test_host_buffer_lifetime},     {"fuzz", test_argument_fuzz}};
  for (auto &t : tests) {
    if (!only.empty() && only != t.name) continue;
    const int before = failures;
    t.fn();
    std::printf("%-14s %s\n", t.name, failures == before ? "ok" : "FAILED");
  }
  // everything accounted for: no device or pinned allocation, no stream, no event outlives its owner
  CHECK(fake_hip_allocations(0) == 0 && fake_hip_allocations(1) == 0 && fake_hip_host_allocations() == 0);
  CHECK(fake_hip_live_streams() == 0 && fake_hip_live_events() == 0);
  std::printf("%s: shim stress on the fake HIP runtime, %d failures\n", failures ? "FAIL" : "ok", failures);
  return failures ? 1 : 0;
}
