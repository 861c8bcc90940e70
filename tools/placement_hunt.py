#!/usr/bin/env python3
"""Lab: does the rate of the banded 256-frame launch depend on WHERE the slabs landed (two regimes were seen between processes:
0.816 with a 6.29 TB/s same-run copy, 0.745 with 5.96)?  One process allocates the ring several times, keeping every earlier
allocation alive, and times the same launch on each.  usage: python tools/placement_hunt.py [tries=6] [ring=256]"""
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

TRIES = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
rng = np.random.default_rng(1)
buf = rng.integers(0, 256, (1, yb + cb), dtype=np.uint8)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))
for t in range(TRIES):
    d_in, d_out = C.c_void_p(), C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, in_stride * RING, C.byref(d_in)))
    _capi.check(lib.bt709hip_malloc(h, ob * RING, C.byref(d_out)))
    for i in range(RING):
        _capi.check(lib.bt709hip_upload(h, d_in.value + i * in_stride, buf.shape[1], buf.ctypes.data, buf.shape[1], buf.shape[1], 1, None))
        lib.bt709hip_stream_synchronize(h, None)  # the upload is asynchronous and `buf` is replaced in the next round
    _capi.check(lib.bt709hip_stream_synchronize(h, None))
    frames, surfs = (Frame * RING)(), (Surface * RING)()
    for i in range(RING):
        b = d_in.value + i * in_stride
        frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
        surfs[i] = Surface(d_out.value + i * ob, W * 4, W, H)

    def run(n, per=None):
        per = per or RING
        for _ in range(n):
            for first in range(0, RING, per):
                fp = C.cast(C.byref(frames, first * C.sizeof(Frame)), C.POINTER(Frame))
                sp = C.cast(C.byref(surfs, first * C.sizeof(Surface)), C.POINTER(Surface))
                _capi.check(lib.bt709hip_decode_batch(dec._handle, per, fp, None, sp, None, 0))
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        run(1); lib.bt709hip_stream_synchronize(h, None)
    rates = []
    for _ in range(3):
        lib.bt709hip_event_record(h, e0, None); run(20); lib.bt709hip_event_record(h, e1, None)
        lib.bt709hip_stream_synchronize(h, None)
        ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
        rates.append(20 * RING * W * H / (ms.value / 1e3) / 1e9)
    plain = []
    dec.setOption(_capi.OPT_XCD_BANDS, 0)
    for _ in range(3):
        lib.bt709hip_event_record(h, e0, None); run(20, 32); lib.bt709hip_event_record(h, e1, None)
        lib.bt709hip_stream_synchronize(h, None)
        ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
        plain.append(20 * RING * W * H / (ms.value / 1e3) / 1e9)
    dec.setOption(_capi.OPT_XCD_BANDS, 1)
    print("              plain map, 32 per launch: %s (%.3f)" % (" ".join("%.1f" % r for r in plain), sorted(plain)[1] * 5.5 / 8000), flush=True)
    # copy probe on the output slab of this allocation
    half = (ob * RING // 2) // 4096 * 4096
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(8):
        _capi.check(lib.bt709hip_copy_probe(h, d_out.value + half, d_out.value, half, None))
    lib.bt709hip_event_record(h, e1, None); lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    copy = 8 * 2 * half / (ms.value / 1e3) / 1e9
    print("allocation %d: in 0x%x out 0x%x  decode %s Gpixel/s (%.3f)  copy %.0f GB/s" % (
        t, d_in.value, d_out.value, " ".join("%.1f" % r for r in rates), sorted(rates)[1] * 5.5 / 8000, copy), flush=True)
