#!/usr/bin/env python3
"""Lab: which cheap probe ranks OUTPUT slabs the way the decode launch does?  N output slabs, one input slab; per slab the rate of
(a) bt709hip_copy_probe (first half -> second half), (b) bt709hip_memset over the slab (write only), (c) the 256-frame launch.
usage: python tools/placement_probes.py [n=6]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
RING, W, H = 256, 3840, 2160
ctx = mb.MetalRenderContext(0)
assert ctx.setupMetal()
lib, h = ctx.lib, ctx.handle
dec = mb.MetalBT709Decoder()
dec.metalRenderContext = ctx
assert dec.setupMetal()
yb, cb, ob = W * H, W * H // 2, W * H * 4
in_stride = (yb + cb + 255) // 256 * 256
d_in = C.c_void_p()
_capi.check(lib.bt709hip_malloc(h, in_stride * RING, C.byref(d_in)))
outs = []
for _ in range(N):
    b = C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, ob * RING, C.byref(b)))
    outs.append(b)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0))
lib.bt709hip_event_create(h, C.byref(e1))


def timed(fn, reps):
    fn()
    lib.bt709hip_stream_synchronize(h, None)
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(reps):
        fn()
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    return ms.value / 1e3 / reps


print("slab   copy GB/s   memset GB/s   decode frac")
for d_out in outs:
    frames, surfs = (Frame * RING)(), (Surface * RING)()
    for i in range(RING):
        b = d_in.value + i * in_stride
        frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
        surfs[i] = Surface(d_out.value + i * ob, W * 4, W, H)
    half = (ob * RING // 2) // 4096 * 4096
    tc = timed(lambda: _capi.check(lib.bt709hip_copy_probe(h, d_out.value + half, d_out.value, half, None)), 6)
    tm = timed(lambda: _capi.check(lib.bt709hip_memset(h, d_out, 0, ob * RING, None)), 6)
    td = timed(lambda: _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0)), 10)
    print("0x%x  %7.0f  %7.0f   %.4f" % (d_out.value, 2 * half / tc / 1e9, ob * RING / tm / 1e9, RING * W * H * 5.5 / td / 8e12), flush=True)
