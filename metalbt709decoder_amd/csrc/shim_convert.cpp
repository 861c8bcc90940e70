// C-ABI shim, the converters either side of the decode (include/bt709hip.h): +unconvert: on packed 4:4:4 words
// (Renderer/BGRAToBT709Converter.h:34-46), pass 2 alone (-renderScaled:, Renderer/MetalScaleRenderContext.h:34-40), the
// BGRA -> NV12 encoder (+convertIntoCoreVideoBuffer:, BGRAToBT709Converter.h:73-76) and the planar <-> NV12 chroma layouts
// of the reference's Y4M files (Renderer/y4m_writer.h:194-241).
#include "shim_internal.h"

namespace bt709shim __attribute__((visibility("hidden"))) {

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

}  // namespace bt709shim

extern "C" {

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
  set_kernel_name(launch_unconvert(t, batch, in_stride, outs[0].stride, static_cast<uint32_t>(width), static_cast<uint32_t>(height), vec,
                                    dec->gamma == kGammaSRGB, s));
  return finish_launch(s, wait_until_completed);
}

int bt709hip_unconvert(bt709hip_decoder *dec, const void *ycbcr_words, size_t in_stride, int width, int height,
                       const bt709hip_surface *out, void *stream, int wait_until_completed) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  return bt709hip_unconvert_batch(dec, 1, &ycbcr_words, in_stride, width, height, out, stream, wait_until_completed);
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
  set_kernel_name(name);
  return finish_launch(s, wait_until_completed);
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
  set_kernel_name(launch_encode(p, count, fast, ctx->xcd_bands != 0, s));
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
  set_kernel_name(launch_planes(p, interleave, s));
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

}  // extern "C"
