// The two additions the HIP binding makes to MetalBT709Decoder's interface (Renderer/MetalBT709Decoder.h:27-72 itself
// does not change).  A class extension -- imported by objc/MetalBT709Decoder+HIP.m, which is the class's primary
// implementation on an MI355X machine, so the property is synthesized there -- and by the one kind of caller that
// needs it: a renderer that wants several frames in flight.  Every other reference call site compiles and behaves
// as before without importing this header.
#import "MetalBT709Decoder.h"

@interface MetalBT709Decoder ()
// NO (default): a host-memory frame is complete when -decodeBT709:... returns, whatever waitUntilCompleted says, because
// the HIP decode is not part of the caller's MTLCommandBuffer.  YES: with waitUntilCompleted:NO up to MaxBuffersInFlight
// (3, AAPLRenderer.m:34) frames stay in flight on their own HIP streams; the caller invokes -finishHIPFrames before
// [commandBuffer commit].
@property (nonatomic, assign) BOOL hipDeferredCompletion;
// Completes every frame still in flight: their pixels are copied into the textures passed with them.  Same thread
// as -decodeBT709:... (the in-flight pool is single-threaded).
- (BOOL) finishHIPFrames;
@end
