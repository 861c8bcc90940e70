"""YUV4MPEG2 ("Y4M") reader/writer for the reference's on-disk 4:2:0 format, plus the GPU
hops between its planar chroma and the decoder's NV12.

Format written by the reference (Renderer/y4m_writer.h):
    header  "YUV4MPEG2 W<w> H<h> F<num>:<den> Ip A1:1 C420jpeg\\n"   (:61-173)
            "XYSCSS=420JPEG\\n"                                       (:175-183, a second line the
                                                                      reference emits after the header)
    frame   "FRAME\\n" + Y (w*h bytes) + U (w/2*h/2) + V (w/2*h/2)    (:194-241)
fps choices are those of Y4MHeaderFPS (:22-30).

The writer reproduces those bytes exactly; the reader also accepts files without the XYSCSS
line and with other header tags (as long as the chroma tag is a 4:2:0 one).
"""
import ctypes as C
import re

import numpy as np

from . import _capi
from .decoder import BGRAToBT709Converter, DeviceBuffer

# Y4MHeaderFPS (y4m_writer.h:22-30) -> the ratio string written at :98-139
FPS = {1: "1:1", 15: "15:1", 24: "24:1", 25: "25:1", 29.97: "30000:1001", 30: "30:1", 60: "60:1"}


class Y4MWriter:
    def __init__(self, path, width, height, fps=30):
        if width % 2 or height % 2:
            raise ValueError("4:2:0 needs even dimensions")
        if fps not in FPS:
            raise ValueError("fps must be one of %s (Y4MHeaderFPS)" % sorted(FPS))
        self.width, self.height = int(width), int(height)
        self.f = open(path, "wb")  # y4m_open_file, :52-59
        self.f.write(("YUV4MPEG2 W%d H%d F%s Ip A1:1 C420jpeg\n" % (self.width, self.height, FPS[fps])).encode())
        self.f.write(b"XYSCSS=420JPEG\n")

    def write_frame(self, y, u, v):
        """y: (H, W) u8; u, v: (H/2, W/2) u8 (y4m_write_frame, :194-241)."""
        y, u, v = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v))
        assert y.shape == (self.height, self.width)
        assert u.shape == v.shape == (self.height // 2, self.width // 2)
        self.f.write(b"FRAME\n")
        self.f.write(y.tobytes())
        self.f.write(u.tobytes())
        self.f.write(v.tobytes())

    def write_nv12(self, y, cbcr):
        """Host NV12 planes (cbcr: (H/2, W) interleaved)."""
        cbcr = np.asarray(cbcr, dtype=np.uint8)
        self.write_frame(y, cbcr[:, 0::2], cbcr[:, 1::2])

    def write_pixel_buffer(self, buf):
        """A device CVPixelBuffer: chroma is de-interleaved on the GPU, then read back."""
        y, u, v = pixel_buffer_to_i420(buf)
        self.write_frame(y, u, v)

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Y4MReader:
    """Iterates (y, u, v) numpy planes of a C420* YUV4MPEG2 file."""

    def __init__(self, path):
        self.f = open(path, "rb")
        line = self.f.readline()
        if not line.startswith(b"YUV4MPEG2 "):
            raise ValueError("not a YUV4MPEG2 file")
        tags = line.decode("ascii", "replace").split()[1:]
        self.width = self.height = None
        self.fps = None
        chroma = "420jpeg"
        for t in tags:
            if t[0] == "W":
                self.width = int(t[1:])
            elif t[0] == "H":
                self.height = int(t[1:])
            elif t[0] == "F":
                m = re.match(r"F(\d+):(\d+)", t)
                self.fps = (int(m.group(1)), int(m.group(2))) if m else None
            elif t[0] == "C":
                chroma = t[1:]
        if not self.width or not self.height:
            raise ValueError("Y4M header lacks W/H")
        if not chroma.startswith("420"):
            raise ValueError("only 4:2:0 chroma is supported, got C%s" % chroma)
        if self.width % 2 or self.height % 2:
            raise ValueError("4:2:0 needs even dimensions")
        pos = self.f.tell()
        nxt = self.f.readline()
        if not nxt.startswith(b"XYSCSS"):  # the reference's extra line (:175-183); optional
            self.f.seek(pos)

    def __iter__(self):
        return self

    def __next__(self):
        line = self.f.readline()
        if not line:
            raise StopIteration
        if not line.startswith(b"FRAME"):
            raise ValueError("expected FRAME, got %r" % line[:16])
        w, h = self.width, self.height
        n = w * h
        data = self.f.read(n + n // 2)
        if len(data) != n + n // 2:
            raise ValueError("truncated frame")
        a = np.frombuffer(data, dtype=np.uint8)
        y = a[:n].reshape(h, w)
        u = a[n:n + n // 4].reshape(h // 2, w // 2)
        v = a[n + n // 4:].reshape(h // 2, w // 2)
        return y, u, v

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ------------------------------------------------------------------ GPU hops

def i420_to_pixel_buffer(ctx, y, u, v, tag_bt709=True):
    """Upload planar Y,U,V and interleave the chroma on the device into a new 420v CVPixelBuffer."""
    y, u, v = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v))
    h, w = y.shape
    buf = BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (w, h))
    if tag_bt709:
        BGRAToBT709Converter.setBT709Attributes(buf)
    cw, ch = w // 2, h // 2
    pitch = (cw + 15) // 16 * 16
    stage = DeviceBuffer(ctx, max(2 * pitch * ch, 16))
    ctx._upload(buf.y_ptr, buf.y_stride, y, None)
    ctx._upload(stage.ptr, pitch, u, None)
    ctx._upload(stage.ptr + pitch * ch, pitch, v, None)
    _capi.check(ctx.lib.bt709hip_interleave_cbcr(ctx.handle, stage.ptr, pitch, stage.ptr + pitch * ch, pitch,
                                                 buf.cbcr_ptr, buf.cbcr_stride, cw, ch, None, 1), "interleave")
    stage.free()
    return buf


def pixel_buffer_to_i420(buf):
    """De-interleave a device CVPixelBuffer's chroma on the GPU and read Y,U,V back."""
    ctx = buf.ctx
    w, h = buf.width, buf.height
    cw, ch = w // 2, h // 2
    pitch = (cw + 15) // 16 * 16
    stage = DeviceBuffer(ctx, max(2 * pitch * ch, 16))
    _capi.check(ctx.lib.bt709hip_deinterleave_cbcr(ctx.handle, buf.cbcr_ptr, buf.cbcr_stride, stage.ptr, pitch,
                                                   stage.ptr + pitch * ch, pitch, cw, ch, None, 1), "deinterleave")
    y = np.empty((h, w), np.uint8)
    u = np.empty((ch, cw), np.uint8)
    v = np.empty((ch, cw), np.uint8)
    lib, hd = ctx.lib, ctx.handle
    if w and h:
        _capi.check(lib.bt709hip_download(hd, y.ctypes.data, w, buf.y_ptr, buf.y_stride, w, h, None))
        _capi.check(lib.bt709hip_download(hd, u.ctypes.data, cw, stage.ptr, pitch, cw, ch, None))
        _capi.check(lib.bt709hip_download(hd, v.ctypes.data, cw, stage.ptr + pitch * ch, pitch, cw, ch, None))
        ctx._sync(None)
    stage.free()
    return y, u, v
