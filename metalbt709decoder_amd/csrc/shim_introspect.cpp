// C-ABI shim, diagnostics and introspection (include/bt709hip_ext.h): the streaming-copy probe, the placement-aware single
// allocation, and the host-side replays of the kernels' table lookups that the parity tests use.  No reference twin.
#include "shim_internal.h"

extern "C" {

// -------------------------------------------------------------- diagnostics

int bt709hip_copy_probe(bt709hip_context *ctx, void *dst, const void *src, size_t bytes, void *stream) {
  if (int rc = bind(ctx)) return rc;
  if (bytes == 0) return BT709HIP_OK;
  if (dst == nullptr || src == nullptr || (bytes & 15) || !aligned(dst, 16) || !aligned(src, 16))
    return BT709HIP_ERR_INVALID_ARG;
  FLUSH_STREAM(ctx, stream);
  set_kernel_name(launch_copy_probe(dst, src, bytes, pick(ctx, stream)));
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
  // Candidates are taken ONE AT A TIME against the incumbent (round 5): two slabs are alive while one is probed (three for the
  // moment of an allocation, see the loop) -- round 4 held all `tries` of them, 34 GB for a 4K ring's output slab -- and the
  // diversity does not suffer: hipFree + hipMalloc of a slab this size hands out other physical pages (profiles/r05_hunt_budget.txt).
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
  void *best_p = nullptr, *loser = nullptr;
  int best = -1;
  float best_rate = -1.0f;
  for (int i = 0; i < (probing ? tries : 1); ++i) {
    // The slab that lost the previous comparison stays allocated until this candidate HAS its memory: freed first, the allocator
    // could hand the very same block back and the "candidates" would be one placement probed `tries` times (round 5's advisor).
    // Three slabs are alive for the moment of the allocation, two while probing.
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess && loser != nullptr) {  // no room for three: give the loser back and try once more
      (void)hipGetLastError();
      (void)hipFree(loser);
      loser = nullptr;
      e = hipMalloc(&p, bytes);
    }
    if (loser != nullptr) (void)hipFree(loser), loser = nullptr;
    if (e != hipSuccess) {
      (void)hipGetLastError();
      break;  // out of memory: the incumbent (if any) stays
    }
    const float rate = probing ? probe(p) : 0.0f;
    if (rates_GBps) rates_GBps[i] = rate;
    if (best_p == nullptr || rate > best_rate) {
      loser = best_p;
      best_p = p, best = i, best_rate = rate;
    } else {
      loser = p;
    }
  }
  if (loser != nullptr) (void)hipFree(loser);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipGetLastError();
  if (best_p == nullptr) return hip_fail(hipErrorOutOfMemory);
  *dptr = best_p;
  if (chosen) *chosen = best;
  return BT709HIP_OK;
}

const char *bt709hip_last_kernel_name(void) { return kernel_name(); }

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
