// C++ host-side mirror of the reference's decode operator over the C ABI
// (include/bt709hip.h, include/bt709hip_ext.h).  Header-only; link with -lbt709hip.
//
// The reference's host side is Objective-C:
//     Renderer/MetalRenderContext.h:17-105    @interface MetalRenderContext
//     Renderer/MetalBT709Decoder.h:21-72      @interface MetalBT709Decoder
// There is no Objective-C runtime on this platform, so the same interface is kept in
// C++: same class and property names, same argument meaning, BOOL-style results
// (true/false, diagnostics through lastStatus()/bt709hip_strerror instead of NSLog).
// INTEGRATION.md shows the Objective-C binding a maintainer of the reference would add.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/bt709hip_ext.h"

namespace bt709 {

enum MetalBT709Gamma {  // Renderer/MetalBT709Decoder.h:15-19
  MetalBT709GammaApple = BT709HIP_GAMMA_APPLE,
  MetalBT709GammaSRGB = BT709HIP_GAMMA_SRGB,
  MetalBT709GammaLinear = BT709HIP_GAMMA_LINEAR,
  MetalBT709GammaITU709 = BT709HIP_GAMMA_ITU709  // extension
};

// Device + queue holder (MetalRenderContext.h:17-43).
class MetalRenderContext {
 public:
  int device = 0;  // HIP ordinal; MTLCreateSystemDefaultDevice() ~ 0

  ~MetalRenderContext() { bt709hip_context_destroy(ctx_); }
  MetalRenderContext() = default;
  MetalRenderContext(const MetalRenderContext &) = delete;
  MetalRenderContext &operator=(const MetalRenderContext &) = delete;

  // -setupMetal: idempotent (MetalRenderContext.m:36-74)
  bool setupMetal() {
    if (ctx_) return true;
    lastStatus_ = bt709hip_context_create(device, &ctx_);
    return lastStatus_ == BT709HIP_OK;
  }
  bt709hip_context *handle() const { return ctx_; }
  int lastStatus() const { return lastStatus_; }

  // commandQueue / -commandBuffer: a HIP stream per in-flight frame
  void *newCommandBuffer() {
    void *s = nullptr;
    lastStatus_ = bt709hip_stream_create(ctx_, &s);
    return s;
  }
  void releaseCommandBuffer(void *s) { bt709hip_stream_destroy(ctx_, s); }
  // a command buffer that is encoded once and replayed: everything issued on `s` between the two
  // calls is recorded into a HIP graph instead of executed
  bool beginRecording(void *s) { return (lastStatus_ = bt709hip_graph_begin_capture(ctx_, s)) == BT709HIP_OK; }
  void *endRecording(void *s) {
    void *g = nullptr;
    lastStatus_ = bt709hip_graph_end_capture(ctx_, s, &g);
    return g;
  }
  bool replay(void *recorded, void *s) { return (lastStatus_ = bt709hip_graph_launch(ctx_, recorded, s)) == BT709HIP_OK; }
  void releaseRecording(void *recorded) { bt709hip_graph_destroy(ctx_, recorded); }
  bool waitUntilCompleted(void *s) { return bt709hip_stream_synchronize(ctx_, s) == BT709HIP_OK; }

 private:
  bt709hip_context *ctx_ = nullptr;
  int lastStatus_ = BT709HIP_OK;
};

// 420v buffer in device memory + the two attachments the decoder validates
// (createCoreVideoYCbCrBuffer / setBT709Attributes, BGRAToBT709Converter.m:412-494).
class CVPixelBuffer {
 public:
  CVPixelBuffer(MetalRenderContext &ctx, int width, int height) : ctx_(ctx) {
    f_.width = width;
    f_.height = height;
    f_.y_stride = f_.cbcr_stride = (static_cast<size_t>(width) + 15) / 16 * 16;
    const size_t ysz = (f_.y_stride * height + 255) / 256 * 256;
    void *p = nullptr;
    if (bt709hip_malloc(ctx.handle(), ysz + f_.cbcr_stride * (height / 2) + 16, &p) == BT709HIP_OK) {
      base_ = p;
      f_.y = p;
      f_.cbcr = static_cast<uint8_t *>(p) + ysz;
    }
  }
  ~CVPixelBuffer() { bt709hip_free(ctx_.handle(), base_); }
  CVPixelBuffer(const CVPixelBuffer &) = delete;
  CVPixelBuffer &operator=(const CVPixelBuffer &) = delete;

  void setBT709Attributes() {
    f_.matrix = BT709HIP_MATRIX_ITU_R_709_2;
    f_.transfer = BT709HIP_TRANSFER_ITU_R_709_2;
  }
  void setAttachments(int matrix, int transfer) {
    f_.matrix = matrix;
    f_.transfer = transfer;
  }
  // copyBT709ToCoreVideo (BGRAToBT709Converter.m:1042-1099): packed (Cr<<16)|(Cb<<8)|Y words in;
  // CbCr of even columns, every row writing into row/2 (the odd row's pair remains).
  bool copyBT709ToCoreVideo(const uint32_t *packed) {
    const int w = f_.width, h = f_.height;
    std::vector<uint8_t> y(static_cast<size_t>(w) * h), c(static_cast<size_t>(w) * (h / 2));
    for (int row = 0; row < h; ++row)
      for (int col = 0; col < w; ++col) {
        const uint32_t p = packed[static_cast<size_t>(row) * w + col];
        y[static_cast<size_t>(row) * w + col] = static_cast<uint8_t>(p & 0xFF);
        if ((col & 1) == 0) {
          c[static_cast<size_t>(row / 2) * w + col] = static_cast<uint8_t>((p >> 8) & 0xFF);
          c[static_cast<size_t>(row / 2) * w + col + 1] = static_cast<uint8_t>((p >> 16) & 0xFF);
        }
      }
    return uploadPlanes(y.data(), c.data());
  }
  bool uploadPlanes(const uint8_t *y, const uint8_t *cbcr) {
    const size_t w = static_cast<size_t>(f_.width);
    int rc = bt709hip_upload(ctx_.handle(), const_cast<void *>(f_.y), f_.y_stride, y, w, w, f_.height, nullptr);
    if (rc == BT709HIP_OK)
      rc = bt709hip_upload(ctx_.handle(), const_cast<void *>(f_.cbcr), f_.cbcr_stride, cbcr, w, w, f_.height / 2,
                           nullptr);
    if (rc == BT709HIP_OK) rc = bt709hip_stream_synchronize(ctx_.handle(), nullptr);
    return rc == BT709HIP_OK;
  }
  const bt709hip_frame *frame() const { return &f_; }

 private:
  MetalRenderContext &ctx_;
  bt709hip_frame f_{};
  void *base_ = nullptr;
};

// Render target (-makeBGRATexture / -getBGRATexturePixels, MetalRenderContext.h:62-105): BGRA8Unorm_sRGB by
// default, or the RGBA16Float intermediate of AAPLRenderer.m:143-170 (pixelFormat = BT709HIP_FORMAT_RGBA16F)
class BGRATexture {
 public:
  BGRATexture(MetalRenderContext &ctx, int width, int height, int pixelFormat = BT709HIP_FORMAT_BGRA8_SRGB) : ctx_(ctx) {
    s_.width = width;
    s_.height = height;
    s_.format = pixelFormat;
    s_.stride = (static_cast<size_t>(width) * (pixelFormat == BT709HIP_FORMAT_RGBA16F ? 8 : 4) + 15) / 16 * 16;
    bt709hip_malloc(ctx.handle(), s_.stride * height + 16, &s_.bgra);
  }
  ~BGRATexture() { bt709hip_free(ctx_.handle(), s_.bgra); }
  BGRATexture(const BGRATexture &) = delete;
  BGRATexture &operator=(const BGRATexture &) = delete;

  // (A<<24)|(R<<16)|(G<<8)|B words; an RGBA16Float texture reads back as two words per pixel (four halves R,G,B,A)
  std::vector<uint32_t> getBGRATexturePixels() const {
    const size_t words = s_.format == BT709HIP_FORMAT_RGBA16F ? 2 : 1;
    std::vector<uint32_t> px(static_cast<size_t>(s_.width) * s_.height * words);
    if (px.empty()) return px;
    const size_t row = static_cast<size_t>(s_.width) * 4 * words;
    bt709hip_download(ctx_.handle(), px.data(), row, s_.bgra, s_.stride, row, s_.height, nullptr);
    bt709hip_stream_synchronize(ctx_.handle(), nullptr);
    return px;
  }
  const bt709hip_surface *surface() const { return &s_; }
  int width() const { return s_.width; }
  int height() const { return s_.height; }

 private:
  MetalRenderContext &ctx_;
  bt709hip_surface s_{};
};

// What the reference's caller actually holds: a CVPixelBuffer whose planes live in HOST memory (the
// GPU reads them in place on Apple's unified memory) and a BGRA texture it reads back with -getBytes.
// On a discrete GPU the decoder moves them through its in-flight frame pool.
struct HostPixelBuffer {  // CVPixelBufferLockBaseAddress + GetBaseAddressOfPlane / GetBytesPerRowOfPlane
  const uint8_t *y = nullptr;
  size_t yStride = 0;
  const uint8_t *cbcr = nullptr;
  size_t cbcrStride = 0;
  int width = 0, height = 0;
  int matrix = BT709HIP_MATRIX_ITU_R_709_2;        // kCVImageBufferYCbCrMatrixKey
  int transfer = BT709HIP_TRANSFER_ITU_R_709_2;    // kCVImageBufferTransferFunctionKey
};
struct HostTexture {  // BGRA8Unorm_sRGB pixels in host memory
  uint8_t *bgra = nullptr;
  size_t stride = 0;
  int width = 0, height = 0;
};
// MTLRenderPassDescriptor as far as the decoder reads it: colorAttachments[0].texture, the view's drawable.  The
// reference renders straight into it when bgraSRGBTexture is nil (MetalBT709Decoder.m:272-281 guards the size check
// with `if (bgraSRGBTexture != nil)`, :462-466 picks the render-pass route; caller: AAPLRenderer.m:927-934).
struct HostRenderPassDescriptor {
  HostTexture colorAttachment0Texture;
};

// Pass 2 on its own (Renderer/MetalScaleRenderContext.h:17-40); the view's drawable is a BGRATexture.
class MetalScaleRenderContext {
 public:
  // -setupRenderPipelines:mtkView: (MetalScaleRenderContext.m:31-51)
  bool setupRenderPipelines(MetalRenderContext &mrc) {
    return (lastStatus_ = bt709hip_render_scaled_prepare(mrc.handle())) == BT709HIP_OK;
  }
  // -renderScaled:mtkView:renderWidth:renderHeight:commandBuffer:renderPassDescriptor:bgraTexture: (.h:34-40)
  bool renderScaled(MetalRenderContext &mrc, const BGRATexture &mtkView, int renderWidth, int renderHeight,
                    void *commandBuffer, const void * /*renderPassDescriptor*/, const BGRATexture &bgraTexture,
                    bool waitUntilCompleted = false) {
    if (renderWidth != mtkView.width() || renderHeight != mtkView.height()) {
      lastStatus_ = BT709HIP_ERR_SIZE_MISMATCH;
      return false;
    }
    lastStatus_ = bt709hip_render_scaled(mrc.handle(), bgraTexture.surface(), mtkView.surface(), commandBuffer,
                                         waitUntilCompleted ? 1 : 0);
    return lastStatus_ == BT709HIP_OK;
  }
  // The same pass over `count` intermediates of one geometry, evenly spaced in memory, in ONE launch
  // (bt709hip_render_scaled_batch; no reference twin)
  bool renderScaledBatch(MetalRenderContext &mrc, int count, const bt709hip_surface *mtkViews, void *commandBuffer,
                         const bt709hip_surface *bgraTextures, bool waitUntilCompleted = false) {
    lastStatus_ = bt709hip_render_scaled_batch(mrc.handle(), count, bgraTextures, mtkViews, commandBuffer, waitUntilCompleted ? 1 : 0);
    return lastStatus_ == BT709HIP_OK;
  }
  int lastStatus() const { return lastStatus_; }

 private:
  int lastStatus_ = BT709HIP_OK;
};

class MetalBT709Decoder {
 public:
  MetalRenderContext *metalRenderContext = nullptr;
  MetalBT709Gamma gamma = MetalBT709GammaApple;  // "Defaults to apple gamma"
  bool useComputeRenderer = true;                // only a compute path exists here
  bool hasAlphaChannel = false;
  int alphaFill = 0xFF;

  ~MetalBT709Decoder() {
    releaseHostPool();
    if (metalRenderContext && metalRenderContext->handle()) bt709hip_decoder_destroy(dec_);
  }
  MetalBT709Decoder() = default;
  MetalBT709Decoder(const MetalBT709Decoder &) = delete;
  MetalBT709Decoder &operator=(const MetalBT709Decoder &) = delete;

  int lastStatus() const { return lastStatus_; }

  // - (BOOL) setupMetal  (MetalBT709Decoder.m:46-104)
  bool setupMetal() {
    if (metalRenderContext == nullptr) return fail(BT709HIP_ERR_NOT_SETUP);
    if (!metalRenderContext->setupMetal()) return fail(BT709HIP_ERR_NO_DEVICE);
    if (dec_) return true;  // second call is a nop
    int rc = bt709hip_decoder_create(metalRenderContext->handle(), gamma, hasAlphaChannel ? 1 : 0, &dec_);
    if (rc == BT709HIP_OK) rc = bt709hip_decoder_set_alpha_fill(dec_, alphaFill);
    if (rc == BT709HIP_OK) rc = bt709hip_decoder_setup(dec_);
    if (rc != BT709HIP_OK) return fail(rc);
    gamma = static_cast<MetalBT709Gamma>(bt709hip_decoder_get_gamma(dec_));  // alpha forces sRGB (.m:165-169)
    return true;
  }

  // - (BOOL) decodeBT709:alphaPixelBuffer:bgraSRGBTexture:commandBuffer:renderPassDescriptor:
  //          renderWidth:renderHeight:waitUntilCompleted:   (MetalBT709Decoder.h:65-72)
  // renderPassDescriptor has no HIP meaning (a drawable is just another BGRATexture).
  bool decodeBT709(const CVPixelBuffer *yCbCrInputTexture, const CVPixelBuffer *alphaPixelBuffer,
                   const BGRATexture *bgraSRGBTexture, void *commandBuffer, const void * /*renderPassDescriptor*/,
                   int renderWidth, int renderHeight, bool waitUntilCompleted) {
    if (!setupMetal()) return false;
    if (yCbCrInputTexture == nullptr || bgraSRGBTexture == nullptr) return fail(BT709HIP_ERR_INVALID_ARG);
    const int rc = bt709hip_decode(dec_, yCbCrInputTexture->frame(),
                                   alphaPixelBuffer ? alphaPixelBuffer->frame() : nullptr,
                                   bgraSRGBTexture->surface(), renderWidth, renderHeight, commandBuffer,
                                   waitUntilCompleted ? 1 : 0);
    return rc == BT709HIP_OK ? ok() : fail(rc);
  }

  // The SAME selector for a caller whose buffers are in host memory, as the reference's are
  // (AAPLRenderer.m:927-957, MetalBT709DecoderTests.m:248-255): planes are copied into the pinned
  // staging of an in-flight slot, upload + decode + download are enqueued on the slot's stream, and
  // the pixels are copied out of the slot when it completes.
  //   bgraSRGBTexture == nullptr: the one-pass route -- the target is renderPassDescriptor's colour attachment 0
  //     (the view's drawable; the frame lands in its top-left renderWidth x renderHeight viewport, .m:575-599).
  //   waitUntilCompleted == true: the texture holds the frame when the call returns (.m:486-489).
  //   waitUntilCompleted == false: the reference only ENCODES here, and whatever the caller encodes next into the
  //     same command buffer (-renderScaled: sampling the intermediate, AAPLRenderer.m:950-976; presentDrawable, :936)
  //     sees the frame.  A host texture has no command buffer to order against, so by default the frame is complete
  //     on return as well -- every reference call site works unchanged.  deferredCompletion = true keeps the frame
  //     in flight instead (up to maxBuffersInFlight of them): the texture is filled when its slot is recycled or by
  //     finishHostFrames(), which such a caller invokes where it would commit the command buffer.
  // alphaPixelBuffer: required iff hasAlphaChannel (only its Y plane is read).
  bool decodeBT709(const HostPixelBuffer &yCbCrInputTexture, const HostPixelBuffer *alphaPixelBuffer,
                   const HostTexture *bgraSRGBTexture, const void * /*commandBuffer*/,
                   const HostRenderPassDescriptor *renderPassDescriptor, int renderWidth, int renderHeight,
                   bool waitUntilCompleted) {
    if (!setupMetal()) return false;
    const HostPixelBuffer &in = yCbCrInputTexture;
    // the checks -processBT709ToSRGB: makes before it touches a plane (.m:272-368), in its order
    if (bgraSRGBTexture != nullptr && (bgraSRGBTexture->width != in.width || bgraSRGBTexture->height != in.height))
      return fail(BT709HIP_ERR_SIZE_MISMATCH);
    if (renderWidth != in.width || renderHeight != in.height) return fail(BT709HIP_ERR_SIZE_MISMATCH);
    if (alphaPixelBuffer && (alphaPixelBuffer->width != in.width || alphaPixelBuffer->height != in.height))
      return fail(BT709HIP_ERR_SIZE_MISMATCH);
    if (in.matrix != BT709HIP_MATRIX_ITU_R_709_2) return fail(BT709HIP_ERR_MATRIX);
    if (in.transfer != requiredTransfer()) return fail(BT709HIP_ERR_TRANSFER);
    if (alphaPixelBuffer && alphaPixelBuffer->transfer != BT709HIP_TRANSFER_LINEAR) return fail(BT709HIP_ERR_ALPHA_TRANSFER);
    if ((in.width & 1) || (in.height & 1)) return fail(BT709HIP_ERR_ODD_DIMENSIONS);
    if (hasAlphaChannel && alphaPixelBuffer == nullptr) return fail(BT709HIP_ERR_INVALID_ARG);
    // the output: the texture, or -- texture nil -- the render pass's colour attachment (.m:462-470)
    if (bgraSRGBTexture == nullptr && renderPassDescriptor == nullptr) return fail(BT709HIP_ERR_INVALID_ARG);
    HostTexture target = bgraSRGBTexture ? *bgraSRGBTexture : renderPassDescriptor->colorAttachment0Texture;
    if (bgraSRGBTexture == nullptr) {  // viewport = renderWidth x renderHeight at the attachment's origin
      if (target.width < in.width || target.height < in.height) return fail(BT709HIP_ERR_SIZE_MISMATCH);
      target.width = in.width;
      target.height = in.height;
    }
    if (in.width == 0 || in.height == 0) return ok();
    if (in.y == nullptr || in.cbcr == nullptr || target.bgra == nullptr) return fail(BT709HIP_ERR_INVALID_ARG);
    if (in.yStride < static_cast<size_t>(in.width) || in.cbcrStride < static_cast<size_t>(in.width) ||
        target.stride < static_cast<size_t>(in.width) * 4)
      return fail(BT709HIP_ERR_STRIDE);
    if (!hostPool(in.width, in.height)) return false;

    int slot = -1;
    void *py = nullptr, *pc = nullptr;
    size_t ys = 0, cs = 0;
    // the pool hands its slots out round-robin and next_ follows it at every acquire (not at submit: a frame that
    // fails after acquire still consumed its turn); the slot about to be recycled may still owe its pixels
    if (pending_[next_].valid && !finishSlot(next_)) return false;
    int rc = bt709hip_pool_acquire(pool_, &slot, &py, &ys, &pc, &cs);
    if (rc != BT709HIP_OK) return fail(rc);
    next_ = (static_cast<size_t>(slot) + 1) % pending_.size();
    copyRows(py, ys, in.y, in.yStride, static_cast<size_t>(in.width), in.height);
    copyRows(pc, cs, in.cbcr, in.cbcrStride, static_cast<size_t>(in.width), in.height / 2);
    if (hasAlphaChannel) {
      void *pa = nullptr;
      size_t as = 0;
      rc = bt709hip_pool_alpha_plane(pool_, slot, &pa, &as);
      if (rc != BT709HIP_OK) {
        bt709hip_pool_release(pool_, slot);  // nothing was enqueued: hand the slot back
        return fail(rc);
      }
      copyRows(pa, as, alphaPixelBuffer->y, alphaPixelBuffer->yStride, static_cast<size_t>(in.width), in.height);
    }
    rc = bt709hip_pool_submit(pool_, slot);
    if (rc != BT709HIP_OK) return fail(rc);  // a failed submit has handed the slot back itself
    pending_[static_cast<size_t>(slot)] = {true, target};
    if (waitUntilCompleted || !deferredCompletion) return finishSlot(static_cast<size_t>(slot));  // .m:486-489
    return ok();
  }
  // the same with the texture by reference (it cannot be nil)
  bool decodeBT709(const HostPixelBuffer &yCbCrInputTexture, const HostPixelBuffer *alphaPixelBuffer,
                   const HostTexture &bgraSRGBTexture, const void *commandBuffer,
                   const HostRenderPassDescriptor *renderPassDescriptor, int renderWidth, int renderHeight,
                   bool waitUntilCompleted) {
    return decodeBT709(yCbCrInputTexture, alphaPixelBuffer, &bgraSRGBTexture, commandBuffer, renderPassDescriptor, renderWidth,
                       renderHeight, waitUntilCompleted);
  }
  // false (default): a host-memory frame is complete when -decodeBT709: returns, whatever waitUntilCompleted says;
  // true: frames submitted with waitUntilCompleted == false stay in flight until their slot is recycled or finishHostFrames()
  bool deferredCompletion = false;

  // Completes every frame submitted with waitUntilCompleted == false (fills their textures).
  bool finishHostFrames() {
    bool all = true;
    for (size_t i = 0; i < pending_.size(); ++i)
      if (pending_[i].valid) all = finishSlot(i) && all;
    return all;
  }
  int maxBuffersInFlight = 3;  // AAPLRenderer.m:34; read when the host pool is first needed

  // -decodeBT709 into an intermediate + MetalScaleRenderContext -renderScaled: (AAPLRenderer.m:940-977), fused:
  // the tuned kernels for the exact 2:1 ratio, the bilinear kernel for any other view size (bit-identical where
  // both apply).
  bool decodeBT709Scaled(const CVPixelBuffer *in, const BGRATexture *out, void *commandBuffer,
                         bool waitUntilCompleted, const CVPixelBuffer *alphaPixelBuffer = nullptr) {
    if (!setupMetal()) return false;
    if (in == nullptr || out == nullptr) return fail(BT709HIP_ERR_INVALID_ARG);
    const bt709hip_frame *f = in->frame();
    const bt709hip_frame *a = alphaPixelBuffer ? alphaPixelBuffer->frame() : nullptr;
    const bt709hip_surface *s = out->surface();
    const bool exact_half = 2 * s->width == f->width && 2 * s->height == f->height && f->width % 4 == 0 && f->height % 4 == 0;
    const int rc = exact_half ? bt709hip_decode_half(dec_, f, a, s, commandBuffer, waitUntilCompleted ? 1 : 0)
                              : bt709hip_decode_scaled(dec_, f, a, s, commandBuffer, waitUntilCompleted ? 1 : 0);
    return rc == BT709HIP_OK ? ok() : fail(rc);
  }

  // kernel-selection knob (bt709hip_decoder_option): tuning and test hook
  bool setOption(int option, int value) {
    if (!setupMetal()) return false;
    const int rc = bt709hip_decoder_set_option(dec_, option, value);
    return rc == BT709HIP_OK ? ok() : fail(rc);
  }

  // The coalescing submit (include/bt709hip_ext.h BT709HIP_OPT_COALESCE): keep the reference's one-decodeBT709-call-per-frame cadence
  // on device-resident frames and let `frames` (2..32; 0 = off) queued calls go out as one launch; maxAgeMicroseconds > 0: a queue
  // older than that is issued by the context's next call on ANY stream (BT709HIP_OPT_COALESCE_MAX_AGE_US), so an idle caller's
  // frames do not wait for ever.
  bool setCoalescing(int frames, int maxAgeMicroseconds = 0) {
    return setOption(BT709HIP_OPT_COALESCE_MAX_AGE_US, maxAgeMicroseconds) && setOption(BT709HIP_OPT_COALESCE, frames);
  }

  bt709hip_decoder *handle() const { return dec_; }

 private:
  bool ok() {
    lastStatus_ = BT709HIP_OK;
    return true;
  }
  bool fail(int rc) {
    lastStatus_ = rc;
    std::fprintf(stderr, "MetalBT709Decoder: %s\n", bt709hip_strerror(rc));  // NSLog in the reference
    return false;
  }
  int requiredTransfer() const {  // MetalBT709Decoder.m:335-353
    return gamma == MetalBT709GammaSRGB ? BT709HIP_TRANSFER_SRGB
                                        : (gamma == MetalBT709GammaLinear ? BT709HIP_TRANSFER_LINEAR : BT709HIP_TRANSFER_ITU_R_709_2);
  }
  static void copyRows(void *dst, size_t dstStride, const void *src, size_t srcStride, size_t rowBytes, int rows) {
    for (int r = 0; r < rows; ++r)
      std::memcpy(static_cast<uint8_t *>(dst) + static_cast<size_t>(r) * dstStride,
                  static_cast<const uint8_t *>(src) + static_cast<size_t>(r) * srcStride, rowBytes);
  }
  // one pool per frame size; a new size drains and replaces it
  bool hostPool(int width, int height) {
    if (pool_ && poolW_ == width && poolH_ == height) return true;
    if (pool_ && !finishHostFrames()) return false;
    releaseHostPool();
    const int rc = bt709hip_pool_create(dec_, width, height, maxBuffersInFlight, &pool_);
    if (rc != BT709HIP_OK) return fail(rc);
    poolW_ = width;
    poolH_ = height;
    pending_.assign(static_cast<size_t>(maxBuffersInFlight), Pending{});
    next_ = 0;
    return true;
  }
  void releaseHostPool() {
    if (pool_) bt709hip_pool_destroy(pool_);
    pool_ = nullptr;
    pending_.clear();
  }
  bool finishSlot(size_t slot) {
    const void *p = nullptr;
    size_t stride = 0;
    const int rc = bt709hip_pool_wait(pool_, static_cast<int>(slot), &p, &stride);
    if (rc != BT709HIP_OK) return fail(rc);
    const HostTexture &t = pending_[slot].texture;
    copyRows(t.bgra, t.stride, p, stride, static_cast<size_t>(t.width) * 4, t.height);
    pending_[slot].valid = false;
    return ok();
  }
  struct Pending {
    bool valid = false;
    HostTexture texture;
  };
  bt709hip_decoder *dec_ = nullptr;
  int lastStatus_ = BT709HIP_OK;
  bt709hip_pool *pool_ = nullptr;
  int poolW_ = 0, poolH_ = 0;
  std::vector<Pending> pending_;
  size_t next_ = 0;
};

// `depth` frames in flight between host memory and the GPU, one HIP stream each: the role of
// CVPixelBufferPool + the texture cache + the renderer's in-flight semaphore (AAPLRenderer.m:34).
class InFlightFramePool {
 public:
  InFlightFramePool(MetalBT709Decoder &decoder, int width, int height, int depth = 3) {
    if (decoder.setupMetal()) lastStatus_ = bt709hip_pool_create(decoder.handle(), width, height, depth, &pool_);
    else lastStatus_ = decoder.lastStatus();
  }
  ~InFlightFramePool() { bt709hip_pool_destroy(pool_); }
  InFlightFramePool(const InFlightFramePool &) = delete;
  InFlightFramePool &operator=(const InFlightFramePool &) = delete;
  bool valid() const { return pool_ != nullptr; }
  int lastStatus() const { return lastStatus_; }

  // pinned planes of the next slot (waits for that slot's previous frame); returns the slot or -1
  int acquire(uint8_t **y, size_t *yStride, uint8_t **cbcr, size_t *cbcrStride) {
    int slot = -1;
    void *py = nullptr, *pc = nullptr;
    lastStatus_ = bt709hip_pool_acquire(pool_, &slot, &py, yStride, &pc, cbcrStride);
    *y = static_cast<uint8_t *>(py);
    *cbcr = static_cast<uint8_t *>(pc);
    return lastStatus_ == BT709HIP_OK ? slot : -1;
  }
  // pinned alpha plane of an acquired slot (decoders with hasAlphaChannel); fill it before submit
  uint8_t *alphaPlane(int slot, size_t *stride) {
    void *p = nullptr;
    lastStatus_ = bt709hip_pool_alpha_plane(pool_, slot, &p, stride);
    return static_cast<uint8_t *>(p);
  }
  bool submit(int slot) { return (lastStatus_ = bt709hip_pool_submit(pool_, slot)) == BT709HIP_OK; }
  // pinned BGRA rows of a submitted slot, valid until the slot is acquired again
  const uint8_t *wait(int slot, size_t *stride) {
    const void *p = nullptr;
    lastStatus_ = bt709hip_pool_wait(pool_, slot, &p, stride);
    return static_cast<const uint8_t *>(p);
  }

 private:
  bt709hip_pool *pool_ = nullptr;
  int lastStatus_ = BT709HIP_OK;
};

// Frames resident in DEVICE memory: a ring of same-sized NV12 inputs and BGRA outputs carved from two slabs and placed by
// bt709hip_ring_create's hunt (include/bt709hip_ext.h "frame ring"; round 4).  The reference's twin is its per-in-flight-frame
// CVPixelBuffers + render texture (Renderer/AAPLRenderer.m:34, 530-862).
class FrameRing {
 public:
  // options: the hunt's budget and the ring's render-target format (bt709hip_ring_options: max_bytes, max_ms, frugal, format --
  // BT709HIP_FORMAT_RGBA16F makes it a ring of RGBA16Float targets; nullptr = BGRA8, a hunt within twice the ring, no time limit)
  FrameRing(MetalBT709Decoder &decoder, int width, int height, int frames, bool halfScale = false, int tries = 0,
            const bt709hip_ring_options *options = nullptr)
      : decoder_(decoder) {
    if (decoder.setupMetal())
      lastStatus_ = bt709hip_ring_create_ex(decoder.handle(), width, height, frames, halfScale ? 1 : 0, tries, options, &ring_);
    else lastStatus_ = decoder.lastStatus();
  }
  ~FrameRing() { bt709hip_ring_destroy(ring_); }
  FrameRing(const FrameRing &) = delete;
  FrameRing &operator=(const FrameRing &) = delete;
  bool valid() const { return ring_ != nullptr; }
  int lastStatus() const { return lastStatus_; }
  int frames() const { return bt709hip_ring_frames(ring_); }
  bt709hip_ring_placement placement() const {
    bt709hip_ring_placement p{};
    bt709hip_ring_placement_info(ring_, &p);
    return p;
  }
  // tight host planes -> frame i (blocking)
  bool upload(int i, const uint8_t *y, const uint8_t *cbcr) {
    bt709hip_frame f{};
    if ((lastStatus_ = bt709hip_ring_frame(ring_, i, &f, nullptr, nullptr)) != BT709HIP_OK) return false;
    bt709hip_context *ctx = bt709hip_decoder_context(decoder_.handle());
    const size_t w = static_cast<size_t>(f.width), h = static_cast<size_t>(f.height);
    lastStatus_ = bt709hip_upload(ctx, const_cast<void *>(f.y), f.y_stride, y, w, w, h, nullptr);
    if (lastStatus_ == BT709HIP_OK) lastStatus_ = bt709hip_upload(ctx, const_cast<void *>(f.cbcr), f.cbcr_stride, cbcr, w, w, h / 2, nullptr);
    if (lastStatus_ == BT709HIP_OK) lastStatus_ = bt709hip_stream_synchronize(ctx, nullptr);
    return lastStatus_ == BT709HIP_OK;
  }
  // frames [first, first + count) in ONE launch
  bool decode(int first, int count, bool waitUntilCompleted) {
    return (lastStatus_ = bt709hip_ring_decode(ring_, first, count, nullptr, waitUntilCompleted ? 1 : 0)) == BT709HIP_OK;
  }
  // the reference's cadence: ONE -decodeBT709: call for frame i (with BT709HIP_OPT_COALESCE on and waitUntilCompleted
  // false the call is validated and queued; a read-back, a flush or the n-th call issues the batch)
  bool decodeFrame(int i, bool waitUntilCompleted) {
    bt709hip_frame f{};
    bt709hip_surface o{};
    if ((lastStatus_ = bt709hip_ring_frame(ring_, i, &f, nullptr, &o)) != BT709HIP_OK) return false;
    return (lastStatus_ = bt709hip_decode(decoder_.handle(), &f, nullptr, &o, f.width, f.height, nullptr, waitUntilCompleted ? 1 : 0)) ==
           BT709HIP_OK;
  }
  // zero-fills every output frame (tests: a later decode must write every pixel again)
  bool clearOutputs() {
    bt709hip_context *ctx = bt709hip_decoder_context(decoder_.handle());
    for (int i = 0; i < frames(); ++i) {
      bt709hip_surface o{};
      if ((lastStatus_ = bt709hip_ring_frame(ring_, i, nullptr, nullptr, &o)) != BT709HIP_OK) return false;
      if ((lastStatus_ = bt709hip_memset(ctx, o.bgra, 0, o.stride * static_cast<size_t>(o.height), nullptr)) != BT709HIP_OK) return false;
    }
    return (lastStatus_ = bt709hip_stream_synchronize(ctx, nullptr)) == BT709HIP_OK;
  }
  // (A<<24)|(R<<16)|(G<<8)|B words of output frame i (the download is ordered behind queued frames)
  std::vector<uint32_t> pixels(int i) {
    bt709hip_surface o{};
    std::vector<uint32_t> px;
    if ((lastStatus_ = bt709hip_ring_frame(ring_, i, nullptr, nullptr, &o)) != BT709HIP_OK) return px;
    px.resize(static_cast<size_t>(o.width) * o.height);
    bt709hip_context *ctx = bt709hip_decoder_context(decoder_.handle());
    const size_t row = static_cast<size_t>(o.width) * 4;
    lastStatus_ = bt709hip_download(ctx, px.data(), row, o.bgra, o.stride, row, o.height, nullptr);
    if (lastStatus_ == BT709HIP_OK) lastStatus_ = bt709hip_stream_synchronize(ctx, nullptr);
    return px;
  }

 private:
  MetalBT709Decoder &decoder_;
  bt709hip_ring *ring_ = nullptr;
  int lastStatus_ = BT709HIP_OK;
};

// ONE process, several GPUs, frames resident in DEVICE memory (bt709hip_ringset_*, round 5): a ring per lane, each with a
// context and a decoder of its own on devices[lane], all driven by the calling thread -- decode() issues one ring launch per
// lane and returns; the devices run concurrently.  The reference's shape: one process that drives everything
// (Renderer/AAPLRenderer.m:874-985).  Host frames: FrameSharder below.
class FrameRingSet {
 public:
  FrameRingSet(const std::vector<int> &devices, int width, int height, int frames, MetalBT709Gamma gamma = MetalBT709GammaApple,
               bool hasAlphaChannel = false, bool halfScale = false, int tries = 0, const bt709hip_ring_options *options = nullptr) {
    lastStatus_ = bt709hip_ringset_create(devices.data(), static_cast<int>(devices.size()), gamma, hasAlphaChannel ? 1 : 0, width,
                                          height, frames, halfScale ? 1 : 0, tries, options, &set_);
  }
  ~FrameRingSet() { bt709hip_ringset_destroy(set_); }
  FrameRingSet(const FrameRingSet &) = delete;
  FrameRingSet &operator=(const FrameRingSet &) = delete;
  bool valid() const { return set_ != nullptr; }
  int lastStatus() const { return lastStatus_; }
  int lanes() const { return bt709hip_ringset_lanes(set_); }
  bt709hip_context *laneContext(int lane) { return bt709hip_ringset_lane_context(set_, lane); }
  bt709hip_decoder *laneDecoder(int lane) { return bt709hip_ringset_lane_decoder(set_, lane); }
  bt709hip_ring *laneRing(int lane) { return bt709hip_ringset_lane_ring(set_, lane); }
  // which physical device a lane sits on (ordinal + PCI bus id)
  bt709hip_device_info laneDevice(int lane) {
    bt709hip_device_info info{};
    lastStatus_ = bt709hip_context_info(laneContext(lane), &info);
    return info;
  }
  // tight host planes -> frame i of lane's ring (blocking)
  bool upload(int lane, int i, const uint8_t *y, const uint8_t *cbcr) {
    bt709hip_frame f{};
    if ((lastStatus_ = bt709hip_ring_frame(laneRing(lane), i, &f, nullptr, nullptr)) != BT709HIP_OK) return false;
    bt709hip_context *ctx = laneContext(lane);
    const size_t w = static_cast<size_t>(f.width), h = static_cast<size_t>(f.height);
    lastStatus_ = bt709hip_upload(ctx, const_cast<void *>(f.y), f.y_stride, y, w, w, h, nullptr);
    if (lastStatus_ == BT709HIP_OK) lastStatus_ = bt709hip_upload(ctx, const_cast<void *>(f.cbcr), f.cbcr_stride, cbcr, w, w, h / 2, nullptr);
    if (lastStatus_ == BT709HIP_OK) lastStatus_ = bt709hip_stream_synchronize(ctx, nullptr);
    return lastStatus_ == BT709HIP_OK;
  }
  // frames [first, first + count) of EVERY lane: one launch per lane, all enqueued before any is waited for
  bool decode(int first, int count, bool waitUntilCompleted) {
    return (lastStatus_ = bt709hip_ringset_decode(set_, first, count, waitUntilCompleted ? 1 : 0)) == BT709HIP_OK;
  }
  bool synchronize() { return (lastStatus_ = bt709hip_ringset_synchronize(set_)) == BT709HIP_OK; }
  // (A<<24)|(R<<16)|(G<<8)|B words of output frame i of lane's ring
  std::vector<uint32_t> pixels(int lane, int i) {
    bt709hip_surface o{};
    std::vector<uint32_t> px;
    if ((lastStatus_ = bt709hip_ring_frame(laneRing(lane), i, nullptr, nullptr, &o)) != BT709HIP_OK) return px;
    px.resize(static_cast<size_t>(o.width) * o.height);
    bt709hip_context *ctx = laneContext(lane);
    const size_t row = static_cast<size_t>(o.width) * 4;
    lastStatus_ = bt709hip_download(ctx, px.data(), row, o.bgra, o.stride, row, o.height, nullptr);
    if (lastStatus_ == BT709HIP_OK) lastStatus_ = bt709hip_stream_synchronize(ctx, nullptr);
    return px;
  }

 private:
  bt709hip_ringset *set_ = nullptr;
  int lastStatus_ = BT709HIP_OK;
};

// ONE process, several GPUs (bt709hip_shard_*): frame i -> lane i mod n, each lane its own context, decoder and in-flight
// pool on devices[lane]; no collective.  What AAPLRenderer's single queue with MaxBuffersInFlight frames
// (Renderer/AAPLRenderer.m:34, 874-985) becomes on an 8-GPU node.  Driven by one thread at a time.
class FrameSharder {
 public:
  FrameSharder(const std::vector<int> &devices, int width, int height, MetalBT709Gamma gamma = MetalBT709GammaApple,
               bool hasAlphaChannel = false, int depth = 3)
      : width_(width), height_(height) {
    lastStatus_ = bt709hip_shard_create(devices.data(), static_cast<int>(devices.size()), gamma, hasAlphaChannel ? 1 : 0, width,
                                        height, depth, &shard_);
  }
  ~FrameSharder() { bt709hip_shard_destroy(shard_); }
  FrameSharder(const FrameSharder &) = delete;
  FrameSharder &operator=(const FrameSharder &) = delete;
  bool valid() const { return shard_ != nullptr; }
  int lastStatus() const { return lastStatus_; }
  int lanes() const { return bt709hip_shard_lanes(shard_); }

  // copies the planes into the next lane's pinned staging and enqueues upload, decode, download there; the ticket, or -1
  long long submit(const HostPixelBuffer &frame, const HostPixelBuffer *alpha = nullptr) {
    bt709hip_frame f{}, a{};
    f.y = frame.y, f.y_stride = frame.yStride, f.cbcr = frame.cbcr, f.cbcr_stride = frame.cbcrStride;
    f.width = frame.width, f.height = frame.height, f.matrix = frame.matrix, f.transfer = frame.transfer;
    if (alpha) {
      a.y = alpha->y, a.y_stride = alpha->yStride, a.width = alpha->width, a.height = alpha->height;
      a.matrix = alpha->matrix, a.transfer = alpha->transfer;
    }
    uint64_t ticket = 0;
    lastStatus_ = bt709hip_shard_submit(shard_, &f, alpha ? &a : nullptr, &ticket);
    return lastStatus_ == BT709HIP_OK ? static_cast<long long>(ticket) : -1;
  }
  // the frame's pinned BGRA rows (valid for lanes * depth further frames), or nullptr
  const uint8_t *wait(long long ticket, size_t *stride) {
    const void *p = nullptr;
    lastStatus_ = bt709hip_shard_wait(shard_, static_cast<uint64_t>(ticket), &p, stride);
    return lastStatus_ == BT709HIP_OK ? static_cast<const uint8_t *>(p) : nullptr;
  }

 private:
  bt709hip_shard *shard_ = nullptr;
  int width_, height_;
  int lastStatus_ = BT709HIP_OK;
};

}  // namespace bt709
