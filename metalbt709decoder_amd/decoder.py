"""Host-side mirror of the reference's decode operator, over the C ABI.

The reference's host side is Objective-C (no ObjC runtime, Foundation, CoreVideo or
Metal exists on this platform), so the classes below keep the reference's names,
property names, argument meaning and BOOL/nil error behaviour in Python so that the
parity tests read like EmptyiOSTests/MetalBT709DecoderTests.m:189-277:

    MetalRenderContext   Renderer/MetalRenderContext.h:17-105
    MetalBT709Decoder    Renderer/MetalBT709Decoder.h:15-72
    CVPixelBuffer        the 420v buffer + colour attachments the decoder reads
                         (BGRAToBT709Converter.m:412-494)
    BGRAToBT709Converter only its buffer helpers: createCoreVideoYCbCrBuffer,
                         setBT709Attributes, copyBT709ToCoreVideo (plumbing; the
                         colour math of that class is CPU code and is NOT here)

Everything that touches pixels runs on the GPU through libbt709hip.so; there is no
CPU implementation of the decode in this package.
"""
import ctypes as C
import logging

import numpy as np

from . import _capi
from ._capi import Frame, Surface

log = logging.getLogger("metalbt709decoder_amd")

# MetalBT709Gamma (MetalBT709Decoder.h:15-19) + the ITU extension
MetalBT709GammaApple = 0
MetalBT709GammaSRGB = 1
MetalBT709GammaLinear = 2
MetalBT709GammaITU709 = 3

# kCVImageBufferYCbCrMatrixKey / kCVImageBufferTransferFunctionKey values
kCVImageBufferYCbCrMatrix_ITU_R_709_2 = 1
kCVImageBufferYCbCrMatrix_ITU_R_601_4 = 2
kCVImageBufferYCbCrMatrix_SMPTE_240M_1995 = 3
kCVImageBufferTransferFunction_ITU_R_709_2 = 1
kCVImageBufferTransferFunction_sRGB = 2
kCVImageBufferTransferFunction_Linear = 3

MTLPixelFormatBGRA8Unorm_sRGB = 81  # default render target format
MTLPixelFormatRGBA16Float = 115     # linear-light half floats: the reference's pre-10.14 intermediate (AAPLRenderer.m:143-170)
_FORMAT_OF = {MTLPixelFormatBGRA8Unorm_sRGB: _capi.FORMAT_BGRA8_SRGB, MTLPixelFormatRGBA16Float: _capi.FORMAT_RGBA16F}


def _align_up(v, a):
    return (v + a - 1) // a * a


STREAMING_SLAB_BYTES = 256 << 20  # the Infinity Cache: a slab this large streams from HBM, where placement matters


class DeviceBuffer:
    """hipMalloc'd bytes owned through the context."""

    def __init__(self, ctx, nbytes, placement_tries=None):
        """placement_tries > 1: bt709hip_malloc_streaming -- that many candidates, the fastest-streaming one kept
        (where a slab lands in HBM changes its streaming rate by up to 10 % on MI355X).  Default (None): 4 for slabs of
        256 MB or more -- the sizes that stream from HBM; a ring of frames is one such slab -- and 1 (plain bt709hip_malloc)
        below that; pass 1 to turn the probing off.  A whole frame ring is better made with FrameRing (bt709hip_ring_create
        probes with the decoder's own launch and chooses the input x output pairing)."""
        self.ctx, self.nbytes = ctx, int(nbytes)
        if placement_tries is None:
            placement_tries = 4 if self.nbytes >= STREAMING_SLAB_BYTES else 1
        p = C.c_void_p()
        self.placement = None
        if placement_tries > 1:
            rates, chosen = (C.c_float * placement_tries)(), C.c_int(-1)
            _capi.check(ctx.lib.bt709hip_malloc_streaming(ctx.handle, self.nbytes, int(placement_tries), C.byref(p), rates,
                                                          C.byref(chosen)), "bt709hip_malloc_streaming")
            self.placement = {"probe_GBps": [round(r, 1) for r in rates], "chosen": chosen.value}
        else:
            _capi.check(ctx.lib.bt709hip_malloc(ctx.handle, self.nbytes, C.byref(p)), "bt709hip_malloc")
        self.ptr = p.value or 0

    def free(self):
        if self.ptr and self.ctx.handle:
            self.ctx.lib.bt709hip_free(self.ctx.handle, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CommandBuffer:
    """One in-flight unit of work = one HIP stream (MTLCommandBuffer's role)."""

    def __init__(self, ctx, stream=None, owned=False):
        self.ctx, self.stream, self.owned = ctx, stream, owned
        self.label = ""

    def commit(self):  # work is enqueued eagerly; nothing to flush
        return None

    def waitUntilCompleted(self):
        _capi.check(self.ctx.lib.bt709hip_stream_synchronize(self.ctx.handle, self.stream), "stream sync")

    def release(self):
        if self.owned and self.stream:
            self.ctx.lib.bt709hip_stream_destroy(self.ctx.handle, self.stream)
            self.stream = None

    # A command buffer that is recorded once and replayed = a HIP graph (needs a stream of its own).
    def beginRecording(self):
        _capi.check(self.ctx.lib.bt709hip_graph_begin_capture(self.ctx.handle, self.stream), "begin capture")

    def endRecording(self):
        g = C.c_void_p()
        _capi.check(self.ctx.lib.bt709hip_graph_end_capture(self.ctx.handle, self.stream, C.byref(g)), "end capture")
        return RecordedCommands(self.ctx, g.value)


class InFlightFramePool:
    """`depth` frames in flight between host memory and the GPU, one HIP stream each (the reference
    keeps MaxBuffersInFlight = 3 behind a semaphore, AAPLRenderer.m:34; its CVPixelBuffers are read by
    the GPU in place, a discrete GPU needs the copies the pool owns)."""

    def __init__(self, decoder, size, depth=3):
        import numpy as np
        self._np = np
        self.decoder, (self.width, self.height), self.depth = decoder, size, depth
        self.lib = decoder.metalRenderContext.lib
        if not decoder.setupMetal():
            raise RuntimeError("decoder setup failed: %s" % decoder.lastStatus)
        h = C.c_void_p()
        _capi.check(self.lib.bt709hip_pool_create(decoder._handle, self.width, self.height, depth, C.byref(h)), "pool create")
        self.handle = h

    def acquire(self):
        """-> (slot, y, cbcr): writable numpy views of the slot's pinned planes."""
        slot, y, c = C.c_int(), C.c_void_p(), C.c_void_p()
        ys, cs = C.c_size_t(), C.c_size_t()
        _capi.check(self.lib.bt709hip_pool_acquire(self.handle, C.byref(slot), C.byref(y), C.byref(ys), C.byref(c), C.byref(cs)),
                    "pool acquire")
        np, w, h = self._np, self.width, self.height
        yv = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_uint8)), shape=(h, ys.value))[:, :w]
        cv = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint8)), shape=(h // 2, cs.value))[:, :w]
        return slot.value, yv, cv

    def alphaPlane(self, slot):
        """Writable numpy view of the slot's pinned alpha plane (decoders with hasAlphaChannel only);
        call between acquire and submit."""
        a, stride = C.c_void_p(), C.c_size_t()
        _capi.check(self.lib.bt709hip_pool_alpha_plane(self.handle, slot, C.byref(a), C.byref(stride)), "pool alpha plane")
        return self._np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint8)), shape=(self.height, stride.value))[:, :self.width]

    def submit(self, slot):
        _capi.check(self.lib.bt709hip_pool_submit(self.handle, slot), "pool submit")

    def wait(self, slot):
        """-> (H, W*4) uint8 view of the slot's pinned BGRA rows (valid until the slot is acquired again)."""
        p, stride = C.c_void_p(), C.c_size_t()
        _capi.check(self.lib.bt709hip_pool_wait(self.handle, slot, C.byref(p), C.byref(stride)), "pool wait")
        a = self._np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.height, stride.value))
        return a[:, :self.width * 4]

    def release(self):
        if self.handle:
            self.lib.bt709hip_pool_destroy(self.handle)
            self.handle = None


class FrameRing:
    """`frames` same-sized frames resident in device memory, inputs carved from one slab and outputs from another, placed by
    bt709hip_ring_create's hunt (include/bt709hip_ext.h "frame ring"): what a streaming application keeps in HBM.  The
    reference's twin is its per-in-flight-frame CVPixelBuffers + render texture (AAPLRenderer.m:34, 530-862); unified
    memory has no placement to choose."""

    def __init__(self, decoder, size, frames, halfScale=False, tries=0, maxBytes=0, maxMilliseconds=0, frugal=False, _handle=None,
                 pixelFormat=MTLPixelFormatBGRA8Unorm_sRGB):
        """maxBytes / maxMilliseconds / frugal: the hunt's budget (bt709hip_ring_options; maxBytes 0 = the default: twice the ring --
        the incumbent pair + one candidate pair, `frugal` -- which also caps the input candidates at two; no time limit).  pixelFormat: the ring's render targets, BGRA8 sRGB or MTLPixelFormatRGBA16Float (the hunt then probes with
        that launch).  _handle: wrap a ring that somebody else owns (FrameRingSet's lanes)."""
        self.pixelFormat = pixelFormat
        self.decoder, (self.width, self.height), self.frames = decoder, size, int(frames)
        self.ctx = decoder.metalRenderContext
        self.lib = self.ctx.lib
        self.halfScale = bool(halfScale)
        self._owned = _handle is None
        if _handle is not None:
            self.handle = C.c_void_p(_handle)
            return
        if not decoder.setupMetal():
            raise RuntimeError("decoder setup failed: %s" % decoder.lastStatus)
        h = C.c_void_p()
        opt = _capi.RingOptions(int(maxBytes), int(maxMilliseconds), int(bool(frugal)), _FORMAT_OF[pixelFormat], 0)
        _capi.check(self.lib.bt709hip_ring_create_ex(decoder._handle, self.width, self.height, self.frames, int(bool(halfScale)),
                                                     int(tries), C.byref(opt), C.byref(h)), "ring create")
        self.handle = h

    def pixelBuffer(self, i):
        """CVPixelBuffer view of input frame i (tagged for the decoder's gamma)."""
        f = Frame()
        _capi.check(self.lib.bt709hip_ring_frame(self.handle, i, C.byref(f), None, None), "ring frame")
        b = CVPixelBuffer(self.ctx, f.width, f.height, f.y_stride, f.cbcr_stride, planes=(f.y, f.cbcr))
        b.setAttachment("YCbCrMatrix", f.matrix)
        b.setAttachment("TransferFunction", f.transfer)
        return b

    def alphaPixelBuffer(self, i):
        a = Frame()
        _capi.check(self.lib.bt709hip_ring_frame(self.handle, i, None, C.byref(a), None), "ring frame")
        if not a.y:
            return None
        b = CVPixelBuffer(self.ctx, a.width, a.height, a.y_stride, a.y_stride, planes=(a.y, a.y))
        b.setAttachment("YCbCrMatrix", a.matrix)
        b.setAttachment("TransferFunction", a.transfer)
        return b

    def texture(self, i):
        """BGRATexture view of output frame i."""
        o = Surface()
        _capi.check(self.lib.bt709hip_ring_frame(self.handle, i, None, None, C.byref(o)), "ring frame")
        return BGRATexture(self.ctx, o.width, o.height, o.stride, ptr=o.bgra, pixelFormat=self.pixelFormat)

    def placement(self):
        p = _capi.RingPlacement()
        _capi.check(self.lib.bt709hip_ring_placement_info(self.handle, C.byref(p)), "ring placement")
        return p

    def decode(self, first=0, count=None, commandBuffer=None, waitUntilCompleted=False):
        """Frames [first, first + count) in one launch."""
        count = self.frames - first if count is None else count
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = self.lib.bt709hip_ring_decode(self.handle, int(first), int(count), stream, int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            self.decoder.lastStatus = rc
            return False
        return True

    def release(self):
        if self.handle:
            if self._owned:
                self.lib.bt709hip_ring_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class FrameRingSet:
    """ONE process, several GPUs, frames resident in DEVICE memory (bt709hip_ringset_*): a FrameRing per lane, each with a
    render context and a decoder of its own on devices[lane], all driven by the calling thread -- decode() issues one ring launch
    per lane and returns; the devices run concurrently.  The reference's shape: one process that drives everything
    (Renderer/AAPLRenderer.m:874-985).  Host frames: FrameSharder."""

    def __init__(self, devices, size, frames, gamma=MetalBT709GammaApple, hasAlphaChannel=False, halfScale=False, tries=0, maxBytes=0,
                 maxMilliseconds=0, frugal=False, pixelFormat=MTLPixelFormatBGRA8Unorm_sRGB):
        self.lib = _capi.load()
        self.width, self.height = size
        self.frames = int(frames)
        arr = (C.c_int * len(devices))(*devices)
        opt = _capi.RingOptions(int(maxBytes), int(maxMilliseconds), int(bool(frugal)), _FORMAT_OF[pixelFormat], 0)
        h = C.c_void_p()
        self.lastStatus = self.lib.bt709hip_ringset_create(arr, len(devices), int(gamma), int(bool(hasAlphaChannel)), self.width,
                                                           self.height, self.frames, int(bool(halfScale)), int(tries), C.byref(opt),
                                                           C.byref(h))
        self.handle = h.value if self.lastStatus == _capi.OK else None
        self.lanes = []
        if not self.handle:
            return
        for lane in range(len(devices)):
            ctx = MetalRenderContext(devices[lane])  # views of what the set owns: never released from here
            ctx.lib, ctx.handle = self.lib, self.lib.bt709hip_ringset_lane_context(self.handle, lane)
            ctx.commandQueue = CommandQueue(ctx)
            dec = MetalBT709Decoder()
            dec.metalRenderContext, dec._handle = ctx, self.lib.bt709hip_ringset_lane_decoder(self.handle, lane)
            dec.hasAlphaChannel = bool(hasAlphaChannel)
            dec.gamma = self.lib.bt709hip_decoder_get_gamma(dec._handle)
            dec._borrowed = ctx._borrowed = True
            ring = FrameRing(dec, size, frames, halfScale=halfScale, _handle=self.lib.bt709hip_ringset_lane_ring(self.handle, lane),
                             pixelFormat=pixelFormat)
            self.lanes.append(ring)

    def decode(self, first=0, count=None, waitUntilCompleted=False):
        """Frames [first, first + count) of EVERY lane's ring: one launch per lane, all enqueued before any is waited for."""
        count = self.frames - first if count is None else count
        self.lastStatus = self.lib.bt709hip_ringset_decode(self.handle, int(first), int(count), int(bool(waitUntilCompleted)))
        return self.lastStatus == _capi.OK

    def synchronize(self):
        self.lastStatus = self.lib.bt709hip_ringset_synchronize(self.handle)
        return self.lastStatus == _capi.OK

    def release(self):
        if self.handle:
            for ring in self.lanes:
                ring.handle = None
                ring.decoder._handle = None
                ring.ctx.handle = None
            self.lanes = []
            self.lib.bt709hip_ringset_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class FrameSharder:
    """ONE process, several GPUs (bt709hip_shard_*): frame i -> lane i mod n, each lane its own context, decoder and
    in-flight pool on devices[lane]; no collective.  What AAPLRenderer's single queue with MaxBuffersInFlight frames
    (Renderer/AAPLRenderer.m:34, 874-985) becomes on an 8-GPU node.  Frames are host numpy planes."""

    def __init__(self, devices, size, gamma=MetalBT709GammaApple, hasAlphaChannel=False, depth=3):
        self.lib = _capi.load()
        self.width, self.height = size
        self.hasAlphaChannel = bool(hasAlphaChannel)
        self.gamma = MetalBT709GammaSRGB if hasAlphaChannel else gamma
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        self.lastStatus = self.lib.bt709hip_shard_create(arr, len(devices), int(gamma), int(self.hasAlphaChannel), self.width,
                                                         self.height, int(depth), C.byref(h))
        self.handle = h.value if self.lastStatus == _capi.OK else None
        self.lanes = len(devices) if self.handle else 0

    def submit(self, y, cbcr, alpha=None, transfer=None):
        """Copies the planes into the next lane's pinned staging and enqueues upload, decode, download there.
        Returns the frame's ticket, or None (lastStatus says why)."""
        if not self.handle:
            return None
        y = np.ascontiguousarray(y, dtype=np.uint8)
        cbcr = np.ascontiguousarray(cbcr, dtype=np.uint8)
        tag = transfer if transfer is not None else {MetalBT709GammaSRGB: kCVImageBufferTransferFunction_sRGB,
                                                      MetalBT709GammaLinear: kCVImageBufferTransferFunction_Linear}.get(
                                                          self.gamma, kCVImageBufferTransferFunction_ITU_R_709_2)
        f = Frame(y.ctypes.data, y.shape[1], cbcr.ctypes.data, cbcr.shape[1], y.shape[1], y.shape[0],
                  kCVImageBufferYCbCrMatrix_ITU_R_709_2, tag)
        a = None
        if alpha is not None:
            alpha = np.ascontiguousarray(alpha, dtype=np.uint8)
            a = Frame(alpha.ctypes.data, alpha.shape[1], None, alpha.shape[1], alpha.shape[1], alpha.shape[0],
                      kCVImageBufferYCbCrMatrix_ITU_R_709_2, kCVImageBufferTransferFunction_Linear)
        t = C.c_uint64()
        self.lastStatus = self.lib.bt709hip_shard_submit(self.handle, C.byref(f), C.byref(a) if a is not None else None,
                                                         C.byref(t))
        return t.value if self.lastStatus == _capi.OK else None

    def wait(self, ticket):
        """(H, W*4) uint8 BGRA rows of that frame (a copy), or None."""
        p, stride = C.c_void_p(), C.c_size_t()
        self.lastStatus = self.lib.bt709hip_shard_wait(self.handle, int(ticket), C.byref(p), C.byref(stride))
        if self.lastStatus != _capi.OK:
            return None
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.height, stride.value))
        return raw[:, :self.width * 4].copy()

    def laneDevice(self, lane):
        return self.lib.bt709hip_shard_lane_device(self.handle, int(lane))

    def release(self):
        if self.handle:
            self.lib.bt709hip_shard_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class RecordedCommands:
    """Everything issued on a command buffer between beginRecording and endRecording; replay()
    re-issues it with one launch."""

    def __init__(self, ctx, graph):
        self.ctx, self.graph = ctx, graph

    def replay(self, commandBuffer=None):
        stream = commandBuffer.stream if commandBuffer is not None else None
        _capi.check(self.ctx.lib.bt709hip_graph_launch(self.ctx.handle, self.graph, stream), "graph launch")

    def release(self):
        if self.graph:
            self.ctx.lib.bt709hip_graph_destroy(self.ctx.handle, self.graph)
            self.graph = None


class CommandQueue:
    def __init__(self, ctx):
        self.ctx = ctx

    def commandBuffer(self, new_stream=False):
        """Default: the context's stream.  new_stream=True gives the frame a stream of
        its own (north-star: one HIP stream per in-flight frame)."""
        if not new_stream:
            return CommandBuffer(self.ctx, None)
        s = C.c_void_p()
        _capi.check(self.ctx.lib.bt709hip_stream_create(self.ctx.handle, C.byref(s)), "stream create")
        return CommandBuffer(self.ctx, s.value, owned=True)


class BGRATexture:
    """Render target in device memory (id<MTLTexture>): 8-bit BGRA sRGB by default, or
    RGBA16Float (pixelFormat=MTLPixelFormatRGBA16Float, 8 bytes per pixel, linear light)."""

    def __init__(self, ctx, width, height, stride=None, ptr=None, pixelFormat=MTLPixelFormatBGRA8Unorm_sRGB):
        self.ctx, self.width, self.height = ctx, int(width), int(height)
        self.pixelFormat = pixelFormat
        self.bytesPerPixel = 8 if pixelFormat == MTLPixelFormatRGBA16Float else 4
        self.stride = int(stride) if stride else _align_up(self.width * self.bytesPerPixel, 16)
        self._buf = None
        if ptr is None:
            self._buf = DeviceBuffer(ctx, max(self.stride * self.height, 16))
            ptr = self._buf.ptr
        self.ptr = ptr

    def surface(self):
        return Surface(self.ptr, self.stride, self.width, self.height, _FORMAT_OF[self.pixelFormat], 0)


class MTLRenderPassDescriptor:
    """As far as the decoder reads it: colorAttachments[0].texture, the view's drawable (a BGRATexture).  The
    reference renders straight into it when bgraSRGBTexture is nil (MetalBT709Decoder.m:272-281, 462-466;
    caller AAPLRenderer.m:927-934)."""

    class _Attachment:
        texture = None

    def __init__(self, texture=None):
        self.colorAttachments = [MTLRenderPassDescriptor._Attachment()]
        self.colorAttachments[0].texture = texture


class CVPixelBuffer:
    """kCVPixelFormatType_420YpCbCr8BiPlanarVideoRange buffer in device memory plus the
    attachments -processBT709ToSRGB: validates (MetalBT709Decoder.m:311-368)."""

    def __init__(self, ctx, width, height, y_stride=None, cbcr_stride=None, planes=None):
        self.ctx, self.width, self.height = ctx, int(width), int(height)
        self.y_stride = int(y_stride) if y_stride else _align_up(self.width, 16)
        self.cbcr_stride = int(cbcr_stride) if cbcr_stride else _align_up(self.width, 16)
        self.attachments = {}
        self._buf = None
        if planes is None:
            ysz = _align_up(self.y_stride * self.height, 256)
            csz = self.cbcr_stride * (self.height // 2)
            self._buf = DeviceBuffer(ctx, max(ysz + csz, 16))
            self.y_ptr, self.cbcr_ptr = self._buf.ptr, self._buf.ptr + ysz
        else:
            self.y_ptr, self.cbcr_ptr = planes

    # CVBufferGetAttachment / CVBufferSetAttachment
    def setAttachment(self, key, value):
        self.attachments[key] = value

    def getAttachment(self, key):
        return self.attachments.get(key, 0)

    def frame(self):
        return Frame(self.y_ptr, self.y_stride, self.cbcr_ptr, self.cbcr_stride, self.width, self.height,
                     self.getAttachment("YCbCrMatrix"), self.getAttachment("TransferFunction"))

    # plane upload/download (CVPixelBufferLockBaseAddress + memcpy in the reference)
    def upload_planes(self, y, cbcr, commandBuffer=None):
        y = np.ascontiguousarray(y, dtype=np.uint8)
        cbcr = np.ascontiguousarray(cbcr, dtype=np.uint8)
        assert y.shape == (self.height, self.width) and cbcr.shape == (self.height // 2, self.width)
        self.ctx._upload(self.y_ptr, self.y_stride, y, commandBuffer)
        self.ctx._upload(self.cbcr_ptr, self.cbcr_stride, cbcr, commandBuffer)
        self.ctx._sync(commandBuffer)

    def download_planes(self, commandBuffer=None):
        """Tight (H, W) luma and (H/2, W) interleaved CbCr arrays read back from the device."""
        y = np.empty((self.height, self.width), dtype=np.uint8)
        c = np.empty((self.height // 2, self.width), dtype=np.uint8)
        if self.width and self.height:
            stream = commandBuffer.stream if commandBuffer else None
            lib, h = self.ctx.lib, self.ctx.handle
            _capi.check(lib.bt709hip_download(h, y.ctypes.data, self.width, self.y_ptr, self.y_stride, self.width,
                                              self.height, stream), "download Y")
            _capi.check(lib.bt709hip_download(h, c.ctypes.data, self.width, self.cbcr_ptr, self.cbcr_stride,
                                              self.width, self.height // 2, stream), "download CbCr")
            self.ctx._sync(commandBuffer)
        return y, c


class MetalRenderContext:
    """Device + queue holder (Renderer/MetalRenderContext.h:17-43)."""

    def __init__(self, device=0):
        self.device = device  # HIP device ordinal; MTLCreateSystemDefaultDevice() ~ 0
        self.handle = None
        self.lib = None
        self.commandQueue = None

    def setupMetal(self):
        """Idempotent (MetalRenderContext.m:36-74).  Returns False when there is no GPU."""
        if self.handle:
            return True
        self.lib = _capi.load()
        h = C.c_void_p()
        rc = self.lib.bt709hip_context_create(int(self.device), C.byref(h))
        if rc != _capi.OK:
            log.error("bt709hip_context_create(%d): %s", self.device, _capi.strerror(rc))
            return False
        self.handle = h.value
        self.commandQueue = CommandQueue(self)
        return True

    def info(self):
        info = _capi.DeviceInfo()
        _capi.check(self.lib.bt709hip_context_info(self.handle, C.byref(info)))
        return info

    def release(self):
        if self.handle:
            if not getattr(self, "_borrowed", False):  # a FrameRingSet lane's context belongs to the set
                self.lib.bt709hip_context_destroy(self.handle)
            self.handle = None

    # -- texture helpers (MetalRenderContext.h:62-105)
    def makeBGRATexture(self, size, pixels=None, stride=None, pixelFormat=MTLPixelFormatBGRA8Unorm_sRGB):
        w, h = size
        tex = BGRATexture(self, w, h, stride, pixelFormat=pixelFormat)
        if pixels is not None:
            self.fillBGRATexture(tex, pixels)
        else:
            _capi.check(self.lib.bt709hip_memset(self.handle, tex.ptr, 0, tex.stride * tex.height, None))
            self._sync(None)
        return tex

    def fillBGRATexture(self, tex, pixels):
        a = np.ascontiguousarray(pixels).view(np.uint8).reshape(tex.height, tex.width * tex.bytesPerPixel)
        self._upload(tex.ptr, tex.stride, a, None)
        self._sync(None)

    def getBGRATexturePixels(self, tex, commandBuffer=None):
        """Read-back as an (H, W) uint32 array of (A<<24)|(R<<16)|(G<<8)|B words; an RGBA16Float
        texture reads back as (H, W, 4) float16 in R,G,B,A order."""
        if tex.bytesPerPixel == 8:
            raw = np.empty((tex.height, tex.width * 8), dtype=np.uint8)
            if tex.height and tex.width:
                stream = commandBuffer.stream if commandBuffer else None
                _capi.check(self.lib.bt709hip_download(self.handle, raw.ctypes.data, raw.shape[1], tex.ptr, tex.stride,
                                                       raw.shape[1], tex.height, stream), "download")
                self._sync(commandBuffer)
            return raw.view(np.float16).reshape(tex.height, tex.width, 4)
        out = np.empty((tex.height, tex.width * 4), dtype=np.uint8)
        if tex.height and tex.width:
            stream = commandBuffer.stream if commandBuffer else None
            _capi.check(self.lib.bt709hip_download(self.handle, out.ctypes.data, out.shape[1], tex.ptr, tex.stride,
                                                   out.shape[1], tex.height, stream), "download")
            self._sync(commandBuffer)
        return out.view(np.uint32).reshape(tex.height, tex.width)

    # -- internals
    def _upload(self, dptr, dpitch, arr, commandBuffer, wait=True):
        """bt709hip_upload is asynchronous and `arr` is pageable numpy memory, often a temporary of the caller: the copy is
        waited for here (wait=False: the caller keeps `arr` alive and unchanged until it synchronises the stream itself)."""
        if arr.size == 0:
            return
        stream = commandBuffer.stream if commandBuffer else None
        _capi.check(self.lib.bt709hip_upload(self.handle, dptr, dpitch, arr.ctypes.data, arr.shape[1],
                                             arr.shape[1], arr.shape[0], stream), "upload")
        if wait:
            self._sync(commandBuffer)

    def _sync(self, commandBuffer):
        stream = commandBuffer.stream if commandBuffer else None
        _capi.check(self.lib.bt709hip_stream_synchronize(self.handle, stream), "stream sync")


class MetalScaleRenderContext:
    """Pass 2 on its own (Renderer/MetalScaleRenderContext.h:17-40): rescale an intermediate texture
    into a view.  The view's drawable is a BGRATexture here (there is no MTKView)."""

    def __init__(self):
        self.fragmentFunction = "samplingShader"  # AAPLShaders.metal:73-85, the only one the reference binds
        self.pipelineState = None
        self.lastStatus = _capi.OK

    def setupRenderPipelines(self, mrc, mtkView=None):
        """-setupRenderPipelines:mtkView: (MetalScaleRenderContext.m:31-51): builds the pass's tables."""
        self.lastStatus = mrc.lib.bt709hip_render_scaled_prepare(mrc.handle)
        self.pipelineState = self.lastStatus == _capi.OK
        return self.pipelineState

    def renderScaled(self, mrc, mtkView, renderWidth, renderHeight, commandBuffer=None, renderPassDescriptor=None,
                     bgraTexture=None, waitUntilCompleted=False):
        """-renderScaled:mtkView:renderWidth:renderHeight:commandBuffer:renderPassDescriptor:bgraTexture:
        (MetalScaleRenderContext.h:34-40).  mtkView: the BGRATexture standing for the view's drawable;
        renderWidth x renderHeight must be its size (the reference sets the viewport to it, .m:80);
        bgraTexture: the intermediate pass 1 rendered (BGRA8 sRGB or RGBA16Float)."""
        if mtkView is None or bgraTexture is None or (renderWidth, renderHeight) != (mtkView.width, mtkView.height):
            self.lastStatus = _capi.ERR_SIZE_MISMATCH if mtkView is not None and bgraTexture is not None else _capi.ERR_INVALID_ARG
            return False
        src, dst = bgraTexture.surface(), mtkView.surface()
        stream = commandBuffer.stream if commandBuffer is not None else None
        self.lastStatus = mrc.lib.bt709hip_render_scaled(mrc.handle, C.byref(src), C.byref(dst), stream,
                                                         int(bool(waitUntilCompleted)))
        if self.lastStatus != _capi.OK:
            log.error("renderScaled: %s", _capi.strerror(self.lastStatus))
        return self.lastStatus == _capi.OK

    def renderScaledBatch(self, mrc, mtkViews, commandBuffer=None, bgraTextures=None, waitUntilCompleted=False):
        """The same pass over several intermediates of one geometry in ONE launch (bt709hip_render_scaled_batch; no
        reference twin): textures and views evenly spaced in memory, as carved from one allocation."""
        if not mtkViews or not bgraTextures or len(mtkViews) != len(bgraTextures):
            self.lastStatus = _capi.ERR_INVALID_ARG
            return False
        n = len(mtkViews)
        srcs = (_capi.Surface * n)(*[t.surface() for t in bgraTextures])
        dsts = (_capi.Surface * n)(*[v.surface() for v in mtkViews])
        stream = commandBuffer.stream if commandBuffer is not None else None
        self.lastStatus = mrc.lib.bt709hip_render_scaled_batch(mrc.handle, n, srcs, dsts, stream, int(bool(waitUntilCompleted)))
        if self.lastStatus != _capi.OK:
            log.error("renderScaledBatch: %s", _capi.strerror(self.lastStatus))
        return self.lastStatus == _capi.OK


class BGRAToBT709Converter:
    """Buffer helpers of Renderer/BGRAToBT709Converter.{h,m} that the decode tests use."""

    @staticmethod
    def createCoreVideoYCbCrBuffer(ctx, size, y_stride=None, cbcr_stride=None):
        w, h = size  # BGRAToBT709Converter.m:471-494
        return CVPixelBuffer(ctx, w, h, y_stride, cbcr_stride)

    @staticmethod
    def setBT709Attributes(buf):
        buf.setAttachment("YCbCrMatrix", kCVImageBufferYCbCrMatrix_ITU_R_709_2)  # .m:412-451
        buf.setAttachment("TransferFunction", kCVImageBufferTransferFunction_ITU_R_709_2)
        return True

    @staticmethod
    def convertIntoCoreVideoBuffer(bgraTexture, cvPixelBuffer, inputGamma, outputGamma, commandBuffer=None,
                                   waitUntilCompleted=True):
        """+convertIntoCoreVideoBuffer:cvPixelBuffer:inputGamma:outputGamma: (BGRAToBT709Converter.h:73-76)
        -> cvpbu_ycbcr_subsample: BGRA pixels -> NV12 with linear-light 2x2 chroma averaging, on the GPU.
        The source is a BGRATexture (device memory) instead of a CGImage.  Gammas are MetalBT709Gamma*
        values Apple / SRGB / Linear.  Returns True/False like the reference."""
        ctx = bgraTexture.ctx
        surf, frame = bgraTexture.surface(), cvPixelBuffer.frame()
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = ctx.lib.bt709hip_encode(ctx.handle, C.byref(surf), C.byref(frame), int(inputGamma), int(outputGamma),
                                     stream, int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            log.error("convertIntoCoreVideoBuffer: %s", _capi.strerror(rc))
            return False
        return True

    @staticmethod
    def convertIntoCoreVideoBuffers(bgraTextures, cvPixelBuffers, inputGamma, outputGamma, commandBuffer=None,
                                    waitUntilCompleted=True):
        """`count` same-sized pictures in one launch (no reference twin: the app converts one
        CGImage per call, BGRAToBT709Converter.m:532-569)."""
        n = len(bgraTextures)
        if n != len(cvPixelBuffers):
            raise ValueError("one output buffer per input texture")
        if n == 0:
            return True
        ctx = bgraTextures[0].ctx
        surfs = (Surface * n)(*[t.surface() for t in bgraTextures])
        frames = (Frame * n)(*[b.frame() for b in cvPixelBuffers])
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = ctx.lib.bt709hip_encode_batch(ctx.handle, n, surfs, frames, int(inputGamma), int(outputGamma), stream,
                                           int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            log.error("convertIntoCoreVideoBuffers: %s", _capi.strerror(rc))
            return False
        return True

    @staticmethod
    def unconvert(decoder, inBT709Pixels, outBGRATexture, width, height, commandBuffer=None):
        """+unconvert:outBGRAPixels:width:height:type: (BGRAToBT709Converter.h:34-46, Software type) on the GPU:
        packed 4:4:4 words Y | Cb << 8 | Cr << 16 (a host array here, uploaded into a scratch buffer) -> BGRA words in
        outBGRATexture, with `decoder`'s gamma and alpha fill (alphaFill = 0 gives unconvertSoftware's words).
        Returns True/False like the reference (odd sizes are refused, .m:69-74)."""
        if not decoder.setupMetal():
            return False
        ctx = decoder.metalRenderContext
        words = np.ascontiguousarray(inBT709Pixels, dtype=np.uint32).reshape(int(height), int(width)) if width and height \
            else np.zeros((0, 0), np.uint32)
        stride = _align_up(int(width) * 4, 16)
        scratch = DeviceBuffer(ctx, max(stride * int(height), 16))
        if words.size:
            ctx._upload(scratch.ptr, stride, words.view(np.uint8).reshape(int(height), int(width) * 4), commandBuffer)
        surf = outBGRATexture.surface()
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = ctx.lib.bt709hip_unconvert(decoder._handle, scratch.ptr, stride, int(width), int(height), C.byref(surf), stream, 1)
        scratch.free()
        if rc != _capi.OK:
            return decoder._fail(rc, "unconvert")
        decoder.lastStatus = _capi.OK
        return True

    @staticmethod
    def unconvertBatch(decoder, inBT709PixelsList, outBGRATextures, width, height, commandBuffer=None):
        """+unconvert: over several frames of one size in ONE launch (bt709hip_unconvert_batch; no reference twin -- the reference
        converts a frame per call): the packed 4:4:4 word arrays are uploaded into one slab (evenly spaced), the outputs are the
        given textures (up to 32 arbitrary ones, or any number carved evenly from one allocation)."""
        if not decoder.setupMetal():
            return False
        ctx = decoder.metalRenderContext
        n = len(inBT709PixelsList)
        stride = int(width) * 4
        scratch = DeviceBuffer(ctx, max(n * stride * int(height), 4), placement_tries=1)
        ptrs = (C.c_void_p * n)()
        for i, words in enumerate(inBT709PixelsList):
            w32 = np.ascontiguousarray(words, dtype=np.uint32).reshape(int(height), int(width))
            ptrs[i] = scratch.ptr + i * stride * int(height)
            ctx._upload(ptrs[i], stride, w32.view(np.uint8).reshape(int(height), stride), commandBuffer)
        surfs = (Surface * n)(*[t.surface() for t in outBGRATextures])
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = ctx.lib.bt709hip_unconvert_batch(decoder._handle, n, ptrs, stride, int(width), int(height), surfs, stream, 1)
        scratch.free()
        if rc != _capi.OK:
            return decoder._fail(rc, "unconvertBatch")
        decoder.lastStatus = _capi.OK
        return True

    @staticmethod
    def copyBT709ToCoreVideo(inBT709Pixels, cvPixelBuffer):
        """Packed (Cr<<16)|(Cb<<8)|Y words -> NV12 planes (BGRAToBT709Converter.m:1042-1099):
        Y of every pixel; CbCr of every even column, every row writing into row/2, so the
        odd row's pair is what remains."""
        w, h = cvPixelBuffer.width, cvPixelBuffer.height
        p = np.ascontiguousarray(inBT709Pixels, dtype=np.uint32).reshape(h, w)
        y = (p & 0xFF).astype(np.uint8)
        cbcr = np.empty((h // 2, w), dtype=np.uint8)
        src = p[1::2, 0::2]
        cbcr[:, 0::2] = (src >> 8) & 0xFF
        cbcr[:, 1::2] = (src >> 16) & 0xFF
        cvPixelBuffer.upload_planes(y, cbcr)
        return True


class MetalBT709Decoder:
    """Renderer/MetalBT709Decoder.h:21-72 over the HIP kernel."""

    def __init__(self):
        self.metalRenderContext = None
        self.colorPixelFormat = MTLPixelFormatBGRA8Unorm_sRGB
        self.gamma = MetalBT709GammaApple  # "Defaults to apple gamma"
        self.useComputeRenderer = True     # there is only a compute path here
        self.hasAlphaChannel = False
        self.alphaFill = 0xFF
        self.lastStatus = _capi.OK
        self._handle = None
        self._options = {}  # bt709hip_decoder_option -> value, applied at setup and on change

    def setOption(self, option, value):
        """Kernel-selection knob (bt709hip_decoder_set_option; _capi.OPT_*): tuning and test hook."""
        self._options[int(option)] = int(value)
        if self._handle:
            _capi.check(self.metalRenderContext.lib.bt709hip_decoder_set_option(self._handle, int(option), int(value)),
                        "decoder set option")

    def flush(self, commandBuffer=None, allStreams=False):
        """Coalescing submit (setOption(_capi.OPT_COALESCE, n)): issue the frames queued for the command buffer's stream (or
        for every stream).  A no-op without the option."""
        lib = self.metalRenderContext.lib
        if allStreams:
            rc = lib.bt709hip_decoder_flush_all(self._handle)
        else:
            rc = lib.bt709hip_decoder_flush(self._handle, commandBuffer.stream if commandBuffer is not None else None)
        if rc != _capi.OK:
            return self._fail(rc, "flush")
        return True

    def _fail(self, rc, what):
        self.lastStatus = rc
        log.error("%s: %s", what, _capi.strerror(rc))  # NSLog in the reference
        return False

    def setupMetal(self):
        if self.metalRenderContext is None:  # MetalBT709Decoder.m:48-54
            self.lastStatus = _capi.ERR_NOT_SETUP
            return False
        ctx = self.metalRenderContext
        if not ctx.setupMetal():
            self.lastStatus = _capi.ERR_NO_DEVICE
            return False
        if self._handle:  # second call is a nop (.m:66-70)
            return True
        h = C.c_void_p()
        rc = ctx.lib.bt709hip_decoder_create(ctx.handle, int(self.gamma), int(bool(self.hasAlphaChannel)),
                                             C.byref(h))
        if rc != _capi.OK:
            return self._fail(rc, "decoder create")
        self._handle = h.value
        ctx.lib.bt709hip_decoder_set_alpha_fill(self._handle, int(self.alphaFill))
        for opt, val in self._options.items():
            _capi.check(ctx.lib.bt709hip_decoder_set_option(self._handle, opt, val), "decoder set option")
        rc = ctx.lib.bt709hip_decoder_setup(self._handle)
        if rc != _capi.OK:
            return self._fail(rc, "decoder setup")
        # hasAlphaChannel forces the sRGB function (.m:165-169)
        self.gamma = ctx.lib.bt709hip_decoder_get_gamma(self._handle)
        self.lastStatus = _capi.OK
        return True

    def decodeBT709(self, yCbCrInputTexture, alphaPixelBuffer=None, bgraSRGBTexture=None, commandBuffer=None,
                    renderPassDescriptor=None, renderWidth=0, renderHeight=0, waitUntilCompleted=False):
        """Returns True on success, False on any validation or launch failure
        (MetalBT709Decoder.h:65-72).  bgraSRGBTexture None: the one-pass route -- the target is
        renderPassDescriptor.colorAttachments[0].texture (the view's drawable, a BGRATexture at least as large
        as the frame; the frame lands in its top-left renderWidth x renderHeight viewport), as
        MetalBT709Decoder.m:272-281, 462-466 and the renderer's call at AAPLRenderer.m:927-934."""
        if not self.setupMetal():
            return False
        target = bgraSRGBTexture
        if target is None and renderPassDescriptor is not None:
            target = renderPassDescriptor.colorAttachments[0].texture
        if yCbCrInputTexture is None or target is None:
            return self._fail(_capi.ERR_INVALID_ARG, "decodeBT709")
        lib = self.metalRenderContext.lib
        frame = yCbCrInputTexture.frame()
        alpha = alphaPixelBuffer.frame() if alphaPixelBuffer is not None else None
        surf = target.surface()
        if bgraSRGBTexture is None:  # the reference checks a texture's size only when one is passed (.m:272-281)
            if surf.width < frame.width or surf.height < frame.height:
                return self._fail(_capi.ERR_SIZE_MISMATCH, "decodeBT709")
            surf.width, surf.height = frame.width, frame.height
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = lib.bt709hip_decode(self._handle, C.byref(frame), C.byref(alpha) if alpha is not None else None,
                                 C.byref(surf), int(renderWidth), int(renderHeight), stream,
                                 int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            return self._fail(rc, "decodeBT709")
        self.lastStatus = _capi.OK
        return True

    def decodeBT709Batch(self, pixelBuffers, textures, alphaPixelBuffers=None, commandBuffer=None,
                         waitUntilCompleted=False):
        """`count` independent same-geometry frames in one launch (no reference twin: the
        reference decodes one frame per command buffer)."""
        if not self.setupMetal():
            return False
        n = len(pixelBuffers)
        frames = (Frame * n)(*[b.frame() for b in pixelBuffers])
        surfs = (Surface * n)(*[t.surface() for t in textures])
        alphas = (Frame * n)(*[b.frame() for b in alphaPixelBuffers]) if alphaPixelBuffers else None
        stream = commandBuffer.stream if commandBuffer is not None else None
        rc = self.metalRenderContext.lib.bt709hip_decode_batch(self._handle, n, frames, alphas, surfs, stream,
                                                               int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            return self._fail(rc, "decodeBT709Batch")
        self.lastStatus = _capi.OK
        return True

    def decodeBT709Scaled(self, yCbCrInputTexture, bgraSRGBTexture, commandBuffer=None, waitUntilCompleted=False,
                          alphaPixelBuffer=None):
        """-decodeBT709 into an intermediate + MetalScaleRenderContext -renderScaled:
        (AAPLRenderer.m:940-977), fused: the tuned kernel for the exact 2:1 ratio, the general
        bilinear kernel for any other view size (bit-identical where both apply)."""
        if not self.setupMetal():
            return False
        frame, surf = yCbCrInputTexture.frame(), bgraSRGBTexture.surface()
        alpha = alphaPixelBuffer.frame() if alphaPixelBuffer is not None else None
        stream = commandBuffer.stream if commandBuffer is not None else None
        lib = self.metalRenderContext.lib
        exact_half = (2 * surf.width == frame.width and 2 * surf.height == frame.height
                      and frame.width % 4 == 0 and frame.height % 4 == 0)
        fn = lib.bt709hip_decode_half if exact_half else lib.bt709hip_decode_scaled  # any view size
        rc = fn(self._handle, C.byref(frame), C.byref(alpha) if alpha is not None else None, C.byref(surf), stream,
                int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            return self._fail(rc, "decodeBT709Scaled")
        self.lastStatus = _capi.OK
        return True

    def decodeBT709ScaledBatch(self, pixelBuffers, textures, commandBuffer=None, waitUntilCompleted=False,
                               alphaPixelBuffers=None):
        """Fused decode + rescale of `count` same-geometry frames into same-sized outputs in one launch
        (no reference twin): the 2:1 kernels when every output is exactly half the frame (large launches
        run the persistent conflict-free kernel), the general bilinear kernel otherwise."""
        if not self.setupMetal():
            return False
        n = len(pixelBuffers)
        frames = (Frame * n)(*[b.frame() for b in pixelBuffers])
        surfs = (Surface * n)(*[t.surface() for t in textures])
        alphas = (Frame * n)(*[b.frame() for b in alphaPixelBuffers]) if alphaPixelBuffers else None
        stream = commandBuffer.stream if commandBuffer is not None else None
        lib = self.metalRenderContext.lib
        exact_half = n > 0 and all(2 * s.width == f.width and 2 * s.height == f.height and f.width % 4 == 0
                                   and f.height % 4 == 0 for f, s in zip(frames, surfs))
        fn = lib.bt709hip_decode_half_batch if exact_half else lib.bt709hip_decode_scaled_batch
        rc = fn(self._handle, n, frames, alphas, surfs, stream, int(bool(waitUntilCompleted)))
        if rc != _capi.OK:
            return self._fail(rc, "decodeBT709ScaledBatch")
        self.lastStatus = _capi.OK
        return True

    def release(self):
        ctx = self.metalRenderContext
        if self._handle and ctx is not None and ctx.handle and not getattr(self, "_borrowed", False):  # a destroyed context took the device with it; a FrameRingSet lane's decoder belongs to the set
            ctx.lib.bt709hip_decoder_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
