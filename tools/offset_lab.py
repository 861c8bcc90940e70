#!/usr/bin/env python3
"""Lab: ONE allocation holding both slabs of a 256-frame 4K ring; the output slab slides inside it.  If the allocation is
physically contiguous in large pieces, the in -> out distance is then a PHYSICAL distance and its effect on the banded launch
is a property of the memory system, not of the allocator's lottery.  usage: python tools/offset_lab.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402

RING, W, H = 256, 3840, 2160
ctx = mb.MetalRenderContext(0)
assert ctx.setupMetal()
lib, h = ctx.lib, ctx.handle
dec = mb.MetalBT709Decoder()
dec.metalRenderContext = ctx
assert dec.setupMetal()
yb, cb, ob = W * H, W * H // 2, W * H * 4
in_stride = (yb + cb + 255) // 256 * 256
in_bytes, out_bytes = in_stride * RING, ob * RING
MB = 1 << 20
gap0 = (in_bytes + 2 * MB - 1) // (2 * MB) * (2 * MB)
slack = 3 << 30
p = C.c_void_p()
_capi.check(lib.bt709hip_malloc(h, gap0 + out_bytes + slack, C.byref(p)))
d_in = p.value
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))
frames, surfs = (Frame * RING)(), (Surface * RING)()
for i in range(RING):
    b = d_in + i * in_stride
    frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)


def measure(d_out):
    for i in range(RING):
        surfs[i] = Surface(d_out + i * ob, W * 4, W, H)
    for _ in range(6):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
    lib.bt709hip_stream_synchronize(h, None)
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(12):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    return 12 * RING * W * H / (ms.value / 1e3) / 1e9 * 5.5 / 8000


t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    measure(d_in + gap0)
extras = [0] + [k * 256 * 1024 for k in range(1, 9)] + [4 * MB, 8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB, 1024 * MB, 1536 * MB, 2048 * MB, 2560 * MB, 3071 * MB]
extras += [4096, 65536, 3 * MB + 4096, 100 * MB + 12288]
for x in extras + [0]:
    print("out = in + %5d MiB + %10d B : frac %.4f" % (gap0 // MB, x, measure(d_in + gap0 + x)), flush=True)
