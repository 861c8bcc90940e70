"""MI355X-native BT.709 NV12 -> sRGB BGRA decode path (gfx950 HIP kernel behind the
reference's MetalBT709Decoder operator API).

The product is metalbt709decoder_amd/libbt709hip.so (C ABI: include/bt709hip.h + include/bt709hip_ext.h); this
package is the host-side mirror of the reference's classes over it.  Nothing here
imports, links or runs oracle/: that directory is test infrastructure.
"""
from . import _capi
from ._capi import Bt709Error, load as load_library
from .decoder import (BGRATexture, BGRAToBT709Converter, CommandBuffer, CVPixelBuffer, FrameRing, FrameRingSet, FrameSharder, InFlightFramePool, MetalBT709Decoder,
                      MetalBT709GammaApple, MetalBT709GammaITU709, MetalBT709GammaLinear, MetalBT709GammaSRGB,
                      MetalRenderContext, MetalScaleRenderContext, MTLRenderPassDescriptor, MTLPixelFormatBGRA8Unorm_sRGB, MTLPixelFormatRGBA16Float,
                      kCVImageBufferTransferFunction_ITU_R_709_2, kCVImageBufferTransferFunction_Linear,
                      kCVImageBufferTransferFunction_sRGB, kCVImageBufferYCbCrMatrix_ITU_R_601_4,
                      kCVImageBufferYCbCrMatrix_ITU_R_709_2)

__all__ = [n for n in dir() if not n.startswith("_")]
