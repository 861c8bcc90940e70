// C-ABI shim, frames that live in HOST memory (include/bt709hip_ext.h): the in-flight frame pool (one HIP stream, pinned
// staging and device buffers per slot; the reference's MaxBuffersInFlight = 3 semaphore, Renderer/AAPLRenderer.m:34, 891-977)
// and the frame sharder that deals frames to one pool per GPU (no collective, nothing crosses GPUs).
#include "shim_internal.h"

extern "C" {

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
    return rc == BT709HIP_OK ? hipSuccess : (last_hip_error() != hipSuccess ? last_hip_error() : hipErrorOutOfMemory);
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

}  // extern "C"
