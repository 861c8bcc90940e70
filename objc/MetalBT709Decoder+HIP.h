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
// The coalescing submit of the HIP decoder (include/bt709hip_ext.h BT709HIP_OPT_COALESCE / _COALESCE_MAX_AGE_US), for a caller that
// hands this decoder DEVICE-resident frames through its bt709hip handle (hipDecoderHandle) at the reference's one-call-per-frame
// cadence: n = 2..32 frames gathered per launch (0 = off, the default), and the age in microseconds after which a queue is issued
// by the context's next call on any stream (0 = no limit).  Host-memory frames -- this class's own selector -- gain nothing: each
// goes through an in-flight pool slot with a stream of its own and is PCIe-bound.  Set before -setupMetal or at any time after.
@property (nonatomic, assign) int hipCoalesceFrames;
@property (nonatomic, assign) int hipCoalesceMaxAgeMicroseconds;
// The bt709hip_decoder behind this object (NULL before -setupMetal), as a void * so that this header needs no bt709hip.h.
- (void *) hipDecoderHandle;
// Completes every frame still in flight: their pixels are copied into the textures passed with them.  Same thread
// as -decodeBT709:... (the in-flight pool is single-threaded).
- (BOOL) finishHIPFrames;
@end
