// C-ABI shim, the decoder (include/bt709hip.h): MetalBT709Decoder's properties, -setupMetal and -decodeBT709:... validated
// the way -processBT709ToSRGB: does (Renderer/MetalBT709Decoder.m:252-492) before the fused kernel is launched, plus the
// batched, 2:1 and any-ratio forms of the same call.
#include "shim_internal.h"

namespace bt709shim __attribute__((visibility("hidden"))) {

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

uint32_t grid_x_for(const bt709hip_context *ctx, uint32_t rows, int frames) {
  uint32_t per_frame = static_cast<uint32_t>(ctx->grid_blocks / (frames > 0 ? frames : 1));
  if (per_frame < 1) per_frame = 1;
  return rows < per_frame ? rows : per_frame;
}

}  // namespace bt709shim

extern "C" {

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
namespace bt709shim __attribute__((visibility("hidden"))) {

static void fold_align(uint32_t *a, uintptr_t v) {
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

}  // namespace
}  // extern "C++"

}  // extern "C"

namespace bt709shim __attribute__((visibility("hidden"))) {

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
    set_kernel_name(launch_decode_rgba16f(p, dec->half, count, dec->has_alpha != 0, info.in_align, info.out_align,
                                           static_cast<uint32_t>(dec->ctx->props.multiProcessorCount), dec->xcd_bands != 0, s));
    return finish_launch(s, wait_until_completed);
  }
  // Fast path: one short-lived workgroup per tile of a row pair, dispatched in address order
  // (see the kernel file's header).  General path keeps the grid-strided shape.
  const bool fast = (p.width % 4) == 0 && info.in_align >= 4 && info.out_align >= 16;
  const uint32_t gx = fast ? quads_tiles(p.width) : grid_x_for(dec->ctx, p.height / 2, count);
  const uint32_t threads = quads_block_threads(p.width);
  last_launch_shape() = LaunchShape{};
  set_kernel_name(launch_decode(p, count, fast ? kVariantQuads : kVariantBlocks, dec->has_alpha != 0,
                                 dec->gamma == kGammaSRGB, dec->nontemporal, dec->xcd_bands, gx, threads, s));
  return finish_launch(s, wait_until_completed);
}

}  // namespace bt709shim

extern "C" {

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
  set_kernel_name(name ? name : launch_decode_half(p, count, wide, dec->has_alpha != 0, dec->nontemporal, gx, threads, s));
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
  set_kernel_name(name);
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

}  // extern "C"
