// TEST-ONLY stand-in, see Foundation/Foundation.h in this directory.
#pragma once
#import <Foundation/Foundation.h>
typedef NSUInteger MTLPixelFormat;
typedef NSUInteger MTLTextureUsage;
enum { MTLPixelFormatBGRA8Unorm = 80, MTLPixelFormatBGRA8Unorm_sRGB = 81, MTLPixelFormatRGBA16Float = 115 };
typedef struct { NSUInteger x, y, z; } MTLOrigin;
typedef struct { NSUInteger width, height, depth; } MTLSize;
typedef struct { MTLOrigin origin; MTLSize size; } MTLRegion;
static inline MTLRegion MTLRegionMake2D(NSUInteger x, NSUInteger y, NSUInteger w, NSUInteger h) {
  MTLRegion r = {{x, y, 0}, {w, h, 1}};
  return r;
}
@protocol MTLDevice
@end
@protocol MTLLibrary
@end
@protocol MTLCommandQueue
@end
@protocol MTLBuffer
@end
@protocol MTLRenderPipelineState
@end
@protocol MTLComputePipelineState
@end
@protocol MTLCommandBuffer
- (void)commit;
- (void)waitUntilCompleted;
@end
@protocol MTLTexture
@property (readonly) NSUInteger width;
@property (readonly) NSUInteger height;
@property (readonly) MTLPixelFormat pixelFormat;
- (void)replaceRegion:(MTLRegion)region mipmapLevel:(NSUInteger)level withBytes:(const void *)bytes bytesPerRow:(NSUInteger)bytesPerRow;
- (void)getBytes:(void *)bytes bytesPerRow:(NSUInteger)bytesPerRow fromRegion:(MTLRegion)region mipmapLevel:(NSUInteger)level;
@end
@interface MTLRenderPassColorAttachmentDescriptor : NSObject
@property (nonatomic, strong) id<MTLTexture> texture;
@end
@interface MTLRenderPassColorAttachmentDescriptorArray : NSObject
- (MTLRenderPassColorAttachmentDescriptor *)objectAtIndexedSubscript:(NSUInteger)index;
@end
@interface MTLRenderPassDescriptor : NSObject
@property (readonly) MTLRenderPassColorAttachmentDescriptorArray *colorAttachments;
@end
