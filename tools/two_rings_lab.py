#!/usr/bin/env python3
"""Lab: why do several PROCESSES on one GPU beat one process (profiles/r03_multi_rank_one_gpu.txt) when several streams of one
process do not?  One process, R separately allocated rings (each what a bench rank owns: 64 4K frames in + out, its own
decoder), launches of 32 frames:
  serial      one stream walks ring 0, ring 1, ... (what one rank does, R times the memory)
  concurrent  ring r on its own stream r, all issued together (what R ranks on one GPU do)
usage: python tools/two_rings_lab.py [rings=2] [steps=30]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
W, H, RING, PER = 3840, 2160, 64, 32
ctx = mb.MetalRenderContext(0)
assert ctx.setupMetal()
lib, h = ctx.lib, ctx.handle
yb, cb, ob = W * H, W * H // 2, W * H * 4
in_stride, out_stride = (yb + cb + 255) // 256 * 256, ob
rings = []
rng = np.random.default_rng(1)
buf = rng.integers(0, 256, (1, yb + cb), dtype=np.uint8)
for r in range(R):
    d_in, d_out = C.c_void_p(), C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, in_stride * RING, C.byref(d_in)))
    _capi.check(lib.bt709hip_malloc(h, out_stride * RING, C.byref(d_out)))
    for i in range(RING):
        _capi.check(lib.bt709hip_upload(h, d_in.value + i * in_stride, buf.shape[1], buf.ctypes.data, buf.shape[1], buf.shape[1], 1, None))
        lib.bt709hip_stream_synchronize(h, None)  # the upload is asynchronous and `buf` is replaced in the next round
    _capi.check(lib.bt709hip_stream_synchronize(h, None))
    frames, surfs = (Frame * RING)(), (Surface * RING)()
    for i in range(RING):
        b = d_in.value + i * in_stride
        frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
        surfs[i] = Surface(d_out.value + i * out_stride, W * 4, W, H)
    dec = mb.MetalBT709Decoder()
    dec.metalRenderContext = ctx
    assert dec.setupMetal()
    s = C.c_void_p()
    _capi.check(lib.bt709hip_stream_create(h, C.byref(s)))
    rings.append((frames, surfs, dec, s.value))


def launch(r, first, stream):
    frames, surfs, dec, _ = rings[r]
    fp = C.cast(C.byref(frames, first * C.sizeof(Frame)), C.POINTER(Frame))
    sp = C.cast(C.byref(surfs, first * C.sizeof(Surface)), C.POINTER(Surface))
    _capi.check(lib.bt709hip_decode_batch(dec._handle, PER, fp, None, sp, stream, 0))


def sync_all():
    for *_, s in rings:
        _capi.check(lib.bt709hip_stream_synchronize(h, s))


def timed(concurrent):
    sync_all()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for first in range(0, RING, PER):
            for r in range(R):
                launch(r, first, rings[r][3] if concurrent else rings[0][3])
    sync_all()
    dt = time.perf_counter() - t0
    return STEPS * R * RING * W * H / dt / 1e9


for _ in range(2):
    timed(False), timed(True)
for mode in (False, True, False, True):
    vals = sorted(timed(mode) for _ in range(5))
    print("%d rings, %-10s  median %7.1f Gpixel/s  (%.1f .. %.1f)  %.3f of 8 TB/s" % (
        R, "concurrent" if mode else "serial", vals[2], vals[0], vals[-1], vals[2] * 5.5 / 8000))
