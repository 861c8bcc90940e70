// Source-only: the Objective-C binding a maintainer of mdejong/MetalBT709Decoder would add to route
// the decode to libbt709hip.so (include/bt709hip.h + bt709hip_ext.h).  There is no Objective-C runtime, Foundation,
// CoreVideo or Metal on the ROCm image, so this file is NOT built or run here; the flow it implements
// -- host planes -> in-flight pool -> host pixels, behind the unchanged 8-argument selector, the nil-texture /
// render-pass-descriptor route included -- is built and tested as host/MetalBT709Decoder.hpp's HostPixelBuffer
// overload (tests/test_gpu_parity.py::test_cpp_host_mirror_host_memory_overload).  Kept in sync with
// INTEGRATION.md section 2 (a CPU test compares the two).
//
// The reference's public header (Renderer/MetalBT709Decoder.h) does not change.  This file REPLACES the method
// bodies of Renderer/MetalBT709Decoder.m (it is compiled instead of that file's -setupMetal / -decodeBT709:
// implementations on a machine whose GPU is an MI355X).  Which reference call sites work unchanged: INTEGRATION.md 2.
#import "MetalBT709Decoder.h"
#import "MetalBT709Decoder+HIP.h"   // hipDeferredCompletion, -finishHIPFrames, hipCoalesceFrames (class extension)
#import "MetalRenderContext.h"
#import <CoreVideo/CoreVideo.h>
#include "bt709hip_ext.h"

enum { BT709HIPMaxInFlight = 3 };  // MaxBuffersInFlight, AAPLRenderer.m:34

@interface MetalBT709Decoder () {
  bt709hip_context *_hipContext;   // HIP twin of metalRenderContext.device / commandQueue
  bt709hip_decoder *_hipDecoder;
  bt709hip_pool *_hipPool;         // CVPixelBufferPool + texture cache + in-flight semaphore (AAPLRenderer.m:34)
  int _poolWidth, _poolHeight;
  // Frames the caller did not wait for: their pixels still have to reach the caller's texture.  STRONG references
  // (ARC object array): a renderer may drop its per-frame texture before the frame is finished.
  id<MTLTexture> _pendingTexture[BT709HIPMaxInFlight];   // nil = the slot owes nothing
  int _nextSlot;                   // the slot bt709hip_pool_acquire hands out next (follows every acquire)
}
- (BOOL) finishHIPSlot:(int)slot;
- (void) applyHIPCoalescing;
@end

static int32_t BT709HIPMatrixTag(CVPixelBufferRef pb) {
  CFTypeRef v = CVBufferGetAttachment(pb, kCVImageBufferYCbCrMatrixKey, NULL);
  if (v && CFEqual(v, kCVImageBufferYCbCrMatrix_ITU_R_709_2)) return BT709HIP_MATRIX_ITU_R_709_2;
  if (v && CFEqual(v, kCVImageBufferYCbCrMatrix_ITU_R_601_4)) return BT709HIP_MATRIX_ITU_R_601_4;
  return BT709HIP_MATRIX_UNSPECIFIED;
}
static int32_t BT709HIPTransferTag(CVPixelBufferRef pb) {
  CFTypeRef v = CVBufferGetAttachment(pb, kCVImageBufferTransferFunctionKey, NULL);
  if (v && CFEqual(v, kCVImageBufferTransferFunction_ITU_R_709_2)) return BT709HIP_TRANSFER_ITU_R_709_2;
  if (v && CFEqual(v, kCVImageBufferTransferFunction_sRGB))        return BT709HIP_TRANSFER_SRGB;
  if (v && CFEqual(v, kCVImageBufferTransferFunction_Linear))      return BT709HIP_TRANSFER_LINEAR;
  return BT709HIP_TRANSFER_UNSPECIFIED;
}
static void BT709HIPCopyPlane(void *dst, size_t dstStride, CVPixelBufferRef pb, size_t plane, size_t rowBytes) {
  const uint8_t *src = CVPixelBufferGetBaseAddressOfPlane(pb, plane);
  const size_t srcStride = CVPixelBufferGetBytesPerRowOfPlane(pb, plane), rows = CVPixelBufferGetHeightOfPlane(pb, plane);
  for (size_t r = 0; r < rows; r++) memcpy((uint8_t *)dst + r * dstStride, src + r * srcStride, rowBytes);
}

@implementation MetalBT709Decoder

- (void) dealloc {
  [self finishHIPFrames];
  bt709hip_pool_destroy(_hipPool);
  bt709hip_decoder_destroy(_hipDecoder);
  bt709hip_context_destroy(_hipContext);
}

// - (BOOL) setupMetal   (MetalBT709Decoder.h:56; reference body .m:46-104)
- (BOOL) setupMetal {
  if (self.metalRenderContext == nil) return FALSE;                     // "metalRenderContext must be set" (.m:48-54)
  if (_hipDecoder != NULL) return TRUE;                                  // second call is a nop (.m:66-70)
  if (_hipContext == NULL && bt709hip_context_create(0, &_hipContext) != BT709HIP_OK) return FALSE;
  if (bt709hip_decoder_create(_hipContext, (int)self.gamma, self.hasAlphaChannel, &_hipDecoder) != BT709HIP_OK) return FALSE;
  if (bt709hip_decoder_setup(_hipDecoder) != BT709HIP_OK) return FALSE;
  self.gamma = (MetalBT709Gamma)bt709hip_decoder_get_gamma(_hipDecoder);   // hasAlphaChannel forces sRGB (.m:165-169)
  [self applyHIPCoalescing];
  return TRUE;
}

// hipCoalesceFrames / hipCoalesceMaxAgeMicroseconds (class extension): applied at setup and whenever they change
- (void) applyHIPCoalescing {
  if (_hipDecoder == NULL) return;
  bt709hip_decoder_set_option(_hipDecoder, BT709HIP_OPT_COALESCE, self.hipCoalesceFrames);
  bt709hip_decoder_set_option(_hipDecoder, BT709HIP_OPT_COALESCE_MAX_AGE_US, self.hipCoalesceMaxAgeMicroseconds);
}
- (void) setHipCoalesceFrames:(int)n { _hipCoalesceFrames = n; [self applyHIPCoalescing]; }
- (void) setHipCoalesceMaxAgeMicroseconds:(int)us { _hipCoalesceMaxAgeMicroseconds = us; [self applyHIPCoalescing]; }
- (void *) hipDecoderHandle { return _hipDecoder; }

// Copies a finished slot's pinned BGRA rows into the texture the caller passed for that frame: its top-left
// frame-sized region (the whole texture for bgraSRGBTexture; the viewport for a larger drawable, .m:575-599).
- (BOOL) finishHIPSlot:(int)slot {
  const void *bgra = NULL; size_t stride = 0;
  if (bt709hip_pool_wait(_hipPool, slot, &bgra, &stride) != BT709HIP_OK) return FALSE;
  id<MTLTexture> tex = _pendingTexture[slot];
  // raw bytes into an sRGB texture: -replaceRegion: does not convert, and the HIP kernel already wrote sRGB-encoded bytes
  [tex replaceRegion:MTLRegionMake2D(0, 0, _poolWidth, _poolHeight) mipmapLevel:0 withBytes:bgra bytesPerRow:stride];
  _pendingTexture[slot] = nil;
  return TRUE;
}

// Completes every frame that is still in flight (hipDeferredCompletion callers: before -commit).
- (BOOL) finishHIPFrames {
  BOOL all = TRUE;
  for (int s = 0; s < BT709HIPMaxInFlight; s++) if (_pendingTexture[s] != nil) all = [self finishHIPSlot:s] && all;
  return all;
}

// The UNCHANGED selector (MetalBT709Decoder.h:65-72).
//   bgraSRGBTexture nil: the one-pass route (AAPLRenderer.m:927-934) -- the target is
//     renderPassDescriptor.colorAttachments[0].texture, the view's drawable (.m:272-281 skips the size check for a nil
//     texture; .m:462-466 renders through the descriptor).
//   waitUntilCompleted TRUE: the texture holds the frame on return (.m:486-489).
//   waitUntilCompleted FALSE: the reference only ENCODES into the caller's command buffer, and whatever the caller
//     encodes next (-renderScaled: sampling the intermediate, AAPLRenderer.m:950-976; presentDrawable, :936) sees the
//     frame.  The HIP decode is not part of that command buffer, so by default the frame is complete on return as
//     well: every reference call site keeps working, unchanged.  A caller that sets hipDeferredCompletion = YES keeps
//     up to three frames in flight instead and calls -finishHIPFrames before it commits the command buffer.
- (BOOL) decodeBT709:(CVPixelBufferRef)yCbCrPixelBuffer
    alphaPixelBuffer:(CVPixelBufferRef)alphaPixelBuffer
     bgraSRGBTexture:(id<MTLTexture>)bgraSRGBTexture
       commandBuffer:(id<MTLCommandBuffer>)commandBuffer
renderPassDescriptor:(MTLRenderPassDescriptor*)renderPassDescriptor
         renderWidth:(int)renderWidth
        renderHeight:(int)renderHeight
  waitUntilCompleted:(BOOL)waitUntilCompleted
{
  if (![self setupMetal]) return FALSE;
  const int width = (int)CVPixelBufferGetWidth(yCbCrPixelBuffer), height = (int)CVPixelBufferGetHeight(yCbCrPixelBuffer);
  // -processBT709ToSRGB:'s checks, in its order (.m:272-368)
  if (bgraSRGBTexture != nil && ((int)bgraSRGBTexture.width != width || (int)bgraSRGBTexture.height != height)) return FALSE;
  if (renderWidth != width || renderHeight != height) return FALSE;
  if (alphaPixelBuffer && ((int)CVPixelBufferGetWidth(alphaPixelBuffer) != width ||
                           (int)CVPixelBufferGetHeight(alphaPixelBuffer) != height)) return FALSE;
  if (BT709HIPMatrixTag(yCbCrPixelBuffer) != BT709HIP_MATRIX_ITU_R_709_2) {
    NSLog(@"unsupported YCbCrMatrix, only BT.709 matrix is supported"); return FALSE; }
  const int32_t wantTransfer = self.gamma == MetalBT709GammaSRGB ? BT709HIP_TRANSFER_SRGB
                             : (self.gamma == MetalBT709GammaLinear ? BT709HIP_TRANSFER_LINEAR : BT709HIP_TRANSFER_ITU_R_709_2);
  if (BT709HIPTransferTag(yCbCrPixelBuffer) != wantTransfer) { NSLog(@"TransferFunction does not match gamma"); return FALSE; }
  if (alphaPixelBuffer && BT709HIPTransferTag(alphaPixelBuffer) != BT709HIP_TRANSFER_LINEAR) return FALSE;
  if (self.hasAlphaChannel && alphaPixelBuffer == NULL) return FALSE;
  // the output: the texture, or -- texture nil -- the render pass's colour attachment (.m:462-470)
  id<MTLTexture> target = bgraSRGBTexture != nil ? bgraSRGBTexture : renderPassDescriptor.colorAttachments[0].texture;
  if (target == nil || (int)target.width < width || (int)target.height < height) return FALSE;

  if (_hipPool == NULL || _poolWidth != width || _poolHeight != height) {   // one pool per frame size
    if (_hipPool && ![self finishHIPFrames]) return FALSE;
    bt709hip_pool_destroy(_hipPool); _hipPool = NULL;
    if (bt709hip_pool_create(_hipDecoder, width, height, BT709HIPMaxInFlight, &_hipPool) != BT709HIP_OK) return FALSE;
    _poolWidth = width; _poolHeight = height; _nextSlot = 0;
  }
  if (_pendingTexture[_nextSlot] != nil && ![self finishHIPSlot:_nextSlot]) return FALSE;  // the slot about to be recycled

  int slot; void *y, *cbcr; size_t ys, cs;
  if (bt709hip_pool_acquire(_hipPool, &slot, &y, &ys, &cbcr, &cs) != BT709HIP_OK) return FALSE;
  _nextSlot = (slot + 1) % BT709HIPMaxInFlight;   // follows the pool at every acquire: a frame that fails below still took its turn
  CVPixelBufferLockBaseAddress(yCbCrPixelBuffer, kCVPixelBufferLock_ReadOnly);
  BT709HIPCopyPlane(y, ys, yCbCrPixelBuffer, 0, (size_t)width);       // Y:    W x H bytes
  BT709HIPCopyPlane(cbcr, cs, yCbCrPixelBuffer, 1, (size_t)width);    // CbCr: (W/2) x (H/2) byte pairs
  CVPixelBufferUnlockBaseAddress(yCbCrPixelBuffer, kCVPixelBufferLock_ReadOnly);
  if (self.hasAlphaChannel) {
    void *a; size_t as;
    if (bt709hip_pool_alpha_plane(_hipPool, slot, &a, &as) != BT709HIP_OK) {
      bt709hip_pool_release(_hipPool, slot);                          // nothing was enqueued: hand the slot back
      return FALSE;
    }
    CVPixelBufferLockBaseAddress(alphaPixelBuffer, kCVPixelBufferLock_ReadOnly);
    BT709HIPCopyPlane(a, as, alphaPixelBuffer, 0, (size_t)width);     // only the Y plane of the alpha buffer is read
    CVPixelBufferUnlockBaseAddress(alphaPixelBuffer, kCVPixelBufferLock_ReadOnly);
  }
  int status = bt709hip_pool_submit(_hipPool, slot);                  // upload + decode + download on the slot's stream
  if (status != BT709HIP_OK) {                                        // a failed submit has handed the slot back itself
    NSLog(@"decodeBT709: %s", bt709hip_strerror(status)); return FALSE; }
  _pendingTexture[slot] = target;
  if (waitUntilCompleted || !self.hipDeferredCompletion) return [self finishHIPSlot:slot];   // .m:486-489
  // hipDeferredCompletion: the frame stays in flight on its own HIP stream; its pixels reach the texture when the slot is
  // recycled (three calls later) or in -finishHIPFrames, which this caller invokes -- on this same thread, the pool is
  // single-threaded -- before [commandBuffer commit].
  return TRUE;
}
@end
