// Source-only: the Objective-C binding a maintainer of mdejong/MetalBT709Decoder would add to route
// the decode to libbt709hip.so (include/bt709hip.h).  There is no Objective-C runtime, Foundation,
// CoreVideo or Metal on the ROCm image, so this file is NOT built or run here; the flow it implements
// -- host planes -> in-flight pool -> host pixels, behind the unchanged 8-argument selector -- is built
// and tested as host/MetalBT709Decoder.hpp's HostPixelBuffer overload
// (tests/test_gpu_parity.py::test_cpp_host_mirror_host_memory_overload).  Kept in sync with
// INTEGRATION.md section 2.
//
// The reference's public header (Renderer/MetalBT709Decoder.h) does not change and neither does any
// caller: AAPLRenderer.m:927-957 and MetalBT709DecoderTests.m:248-255 keep calling
//   -decodeBT709:alphaPixelBuffer:bgraSRGBTexture:commandBuffer:renderPassDescriptor:
//    renderWidth:renderHeight:waitUntilCompleted:
// This file REPLACES the method bodies of Renderer/MetalBT709Decoder.m (it is compiled instead of that
// file's -setupMetal / -decodeBT709: implementations on a machine whose GPU is an MI355X).
#import "MetalBT709Decoder.h"
#import "MetalRenderContext.h"
#import <CoreVideo/CoreVideo.h>
#include "bt709hip.h"

// Frames the caller did not wait for: their pixels still have to reach the caller's texture.
typedef struct {
  BOOL valid;
  __unsafe_unretained id<MTLTexture> texture;
} BT709HIPPending;

@interface MetalBT709Decoder () {
  bt709hip_context *_hipContext;   // HIP twin of metalRenderContext.device / commandQueue
  bt709hip_decoder *_hipDecoder;
  bt709hip_pool *_hipPool;         // CVPixelBufferPool + texture cache + in-flight semaphore (AAPLRenderer.m:34)
  int _poolWidth, _poolHeight;
  BT709HIPPending _pending[3];
  int _nextSlot;
}
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
  return TRUE;
}

// Copies a finished slot's pinned BGRA rows into the texture the caller passed for that frame.
- (BOOL) finishHIPSlot:(int)slot {
  const void *bgra = NULL; size_t stride = 0;
  if (bt709hip_pool_wait(_hipPool, slot, &bgra, &stride) != BT709HIP_OK) return FALSE;
  id<MTLTexture> tex = _pending[slot].texture;
  // raw bytes into an sRGB texture: -replaceRegion: does not convert, and the HIP kernel already wrote sRGB-encoded bytes
  [tex replaceRegion:MTLRegionMake2D(0, 0, tex.width, tex.height) mipmapLevel:0 withBytes:bgra bytesPerRow:stride];
  _pending[slot].valid = FALSE;
  return TRUE;
}

// Completes every frame submitted with waitUntilCompleted:FALSE.
- (BOOL) finishHIPFrames {
  BOOL all = TRUE;
  for (int s = 0; s < 3; s++) if (_pending[s].valid) all = [self finishHIPSlot:s] && all;
  return all;
}

// The UNCHANGED selector (MetalBT709Decoder.h:65-72).  commandBuffer and renderPassDescriptor have no HIP
// meaning: the pool's per-slot HIP stream plays the command buffer's role, and a view drawable is a
// texture like any other (pass it as bgraSRGBTexture).
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
  if ((int)bgraSRGBTexture.width != width || (int)bgraSRGBTexture.height != height) return FALSE;
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

  if (_hipPool == NULL || _poolWidth != width || _poolHeight != height) {   // one pool per frame size
    if (_hipPool && ![self finishHIPFrames]) return FALSE;
    bt709hip_pool_destroy(_hipPool); _hipPool = NULL;
    if (bt709hip_pool_create(_hipDecoder, width, height, 3, &_hipPool) != BT709HIP_OK) return FALSE;
    _poolWidth = width; _poolHeight = height; _nextSlot = 0;
  }
  if (_pending[_nextSlot].valid && ![self finishHIPSlot:_nextSlot]) return FALSE;  // the slot about to be recycled

  int slot; void *y, *cbcr; size_t ys, cs;
  if (bt709hip_pool_acquire(_hipPool, &slot, &y, &ys, &cbcr, &cs) != BT709HIP_OK) return FALSE;
  CVPixelBufferLockBaseAddress(yCbCrPixelBuffer, kCVPixelBufferLock_ReadOnly);
  BT709HIPCopyPlane(y, ys, yCbCrPixelBuffer, 0, (size_t)width);       // Y:    W x H bytes
  BT709HIPCopyPlane(cbcr, cs, yCbCrPixelBuffer, 1, (size_t)width);    // CbCr: (W/2) x (H/2) byte pairs
  CVPixelBufferUnlockBaseAddress(yCbCrPixelBuffer, kCVPixelBufferLock_ReadOnly);
  if (self.hasAlphaChannel) {
    void *a; size_t as;
    if (bt709hip_pool_alpha_plane(_hipPool, slot, &a, &as) != BT709HIP_OK) return FALSE;
    CVPixelBufferLockBaseAddress(alphaPixelBuffer, kCVPixelBufferLock_ReadOnly);
    BT709HIPCopyPlane(a, as, alphaPixelBuffer, 0, (size_t)width);     // only the Y plane of the alpha buffer is read
    CVPixelBufferUnlockBaseAddress(alphaPixelBuffer, kCVPixelBufferLock_ReadOnly);
  }
  int status = bt709hip_pool_submit(_hipPool, slot);                  // upload + decode + download on the slot's stream
  if (status != BT709HIP_OK) { NSLog(@"decodeBT709: %s", bt709hip_strerror(status)); return FALSE; }
  _pending[slot].valid = TRUE;
  _pending[slot].texture = bgraSRGBTexture;
  _nextSlot = (slot + 1) % 3;
  if (waitUntilCompleted) return [self finishHIPSlot:slot];           // .m:486-489
  // Asynchronous, like the reference: the frame is in flight on its own HIP stream.  Its pixels reach the
  // texture when the slot is recycled (three calls later) or when the caller invokes -finishHIPFrames --
  // on this same thread (the pool is single-threaded), at the point where the reference's renderer
  // presents the drawable (AAPLRenderer.m:979-1069).
  return TRUE;
}
@end
