#!/usr/bin/env python3
"""Lab: is the placement effect separable?  N input slabs x N output slabs of one process, the 256-frame launch timed on
every pairing.  usage: python tools/placement_cross.py [n=4] [ring=256]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
RING = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W, H = 3840, 2160
ctx = mb.MetalRenderContext(0)
assert ctx.setupMetal()
lib, h = ctx.lib, ctx.handle
dec = mb.MetalBT709Decoder()
dec.metalRenderContext = ctx
assert dec.setupMetal()
yb, cb, ob = W * H, W * H // 2, W * H * 4
in_stride = (yb + cb + 255) // 256 * 256
ins, outs = [], []
for _ in range(N):
    a, b = C.c_void_p(), C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, in_stride * RING, C.byref(a)))
    _capi.check(lib.bt709hip_malloc(h, ob * RING, C.byref(b)))
    ins.append(a)
    outs.append(b)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0))
lib.bt709hip_event_create(h, C.byref(e1))


def rate(d_in, d_out):
    frames, surfs = (Frame * RING)(), (Surface * RING)()
    for i in range(RING):
        b = d_in.value + i * in_stride
        frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
        surfs[i] = Surface(d_out.value + i * ob, W * 4, W, H)
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
        lib.bt709hip_stream_synchronize(h, None)
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(10):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    return 10 * RING * W * H * 5.5 / (ms.value / 1e3) / 8e12


print("rows = input slab, columns = output slab; fraction of 8 TB/s")
for a in ins:
    print("  ".join("%.4f" % rate(a, b) for b in outs), flush=True)
print("again:")
for a in ins:
    print("  ".join("%.4f" % rate(a, b) for b in outs), flush=True)
