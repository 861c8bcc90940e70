// Source-only: the Objective-C binding a maintainer of mdejong/MetalBT709Decoder would add to
// route -decodeBT709:... to libbt709hip.so (include/bt709hip.h).  There is no Objective-C
// runtime, Foundation, CoreVideo or Metal on the ROCm image, so this file is NOT built or run
// here; every C-ABI call it makes is exercised by host/MetalBT709Decoder.hpp (C++) and
// metalbt709decoder_amd/decoder.py (Python), which mirror the same interface.  Kept in sync with
// INTEGRATION.md section 2.
// Renderer/MetalBT709Decoder+HIP.m
#import "MetalBT709Decoder.h"
#import <CoreVideo/CoreVideo.h>
#include "bt709hip.h"

@interface MetalBT709Decoder (HIP)
@property (nonatomic, assign) bt709hip_context *hipContext;   // owned by the HIP twin of MetalRenderContext
@property (nonatomic, assign) bt709hip_decoder *hipDecoder;
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

@implementation MetalBT709Decoder (HIP)

- (BOOL) setupHIP {                                   // twin of -setupMetal (MetalBT709Decoder.m:46-104)
  if (self.hipContext == NULL) return FALSE;          // "metalRenderContext must be set"
  if (self.hipDecoder == NULL) {
    bt709hip_decoder *dec = NULL;
    if (bt709hip_decoder_create(self.hipContext, (int)self.gamma, self.hasAlphaChannel, &dec) != BT709HIP_OK) return FALSE;
    self.hipDecoder = dec;
  }
  return bt709hip_decoder_setup(self.hipDecoder) == BT709HIP_OK;
}

// Same selector shape as MetalBT709Decoder.h:65-72; planes are device pointers the caller uploaded with
// bt709hip_upload (or that a HIP-side demuxer produced); `stream` plays the commandBuffer's role.
- (BOOL) decodeBT709HIP:(CVPixelBufferRef)yCbCrPixelBuffer
               devicePlanes:(const void * const [2])planes      // {Y, CbCr} device pointers
               planeStrides:(const size_t [2])strides
           alphaDevicePlane:(const void *)alphaPlane alphaStride:(size_t)alphaStride
            alphaPixelBuffer:(CVPixelBufferRef)alphaPixelBuffer
            bgraSRGBSurface:(bt709hip_surface)surface
                     stream:(void *)stream
                renderWidth:(int)renderWidth renderHeight:(int)renderHeight
         waitUntilCompleted:(BOOL)waitUntilCompleted
{
  if (![self setupHIP]) return FALSE;
  bt709hip_frame f = { planes[0], strides[0], planes[1], strides[1],
                       (int32_t)CVPixelBufferGetWidth(yCbCrPixelBuffer), (int32_t)CVPixelBufferGetHeight(yCbCrPixelBuffer),
                       BT709HIPMatrixTag(yCbCrPixelBuffer), BT709HIPTransferTag(yCbCrPixelBuffer) };
  bt709hip_frame a; const bt709hip_frame *ap = NULL;
  if (alphaPixelBuffer != NULL) {
    a = (bt709hip_frame){ alphaPlane, alphaStride, NULL, 0,
                          (int32_t)CVPixelBufferGetWidth(alphaPixelBuffer), (int32_t)CVPixelBufferGetHeight(alphaPixelBuffer),
                          BT709HIPMatrixTag(alphaPixelBuffer), BT709HIPTransferTag(alphaPixelBuffer) };
    ap = &a;
  }
  int status = bt709hip_decode(self.hipDecoder, &f, ap, &surface, renderWidth, renderHeight, stream, waitUntilCompleted);
  if (status != BT709HIP_OK) { NSLog(@"decodeBT709HIP: %s", bt709hip_strerror(status)); return FALSE; }
  return TRUE;
}
@end
