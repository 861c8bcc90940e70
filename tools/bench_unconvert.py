#!/usr/bin/env python3
"""Throughput of bt709hip_unconvert (the GPU twin of +[BGRAToBT709Converter unconvert:...], packed 4:4:4 words in, BGRA words out:
4 B read + 4 B written per pixel) over a ring of resident 4K frames, one call per frame (the entry point has no batch form, as the
reference's has none), on 1 / 2 / 3 HIP streams.  Runs on the GPU box:  python tools/bench_unconvert.py [ring=64] [steps=400]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import gpu_helpers as gh  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd.decoder import DeviceBuffer  # noqa: E402

ring = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
if len(sys.argv) > 3:  # a variant build of the library (tools/lab_variants.py); must be the first load of the process
    _capi.load(os.path.abspath(sys.argv[3]))
W, H = 3840, 2160
ctx = gh.context(); lib, h = ctx.lib, ctx.handle
dec = gh.make_decoder(mb.MetalBT709GammaApple)
pitch = W * H * 4
src, dst = DeviceBuffer(ctx, ring * pitch), DeviceBuffer(ctx, ring * pitch)
words = np.random.default_rng(7).integers(0, 1 << 24, (H, W), dtype=np.uint32)
for i in range(ring):
    frame = np.ascontiguousarray(np.roll(words, i, axis=1)).view(np.uint8).reshape(H, W * 4)
    ctx._upload(src.ptr + i * pitch, W * 4, frame, None)
    ctx._sync(None)  # the upload is asynchronous: the host array must outlive it
surfs = [_capi.Surface(dst.ptr + i * pitch, W * 4, W, H, 0, 0) for i in range(ring)]
e0, e1 = C.c_void_p(), C.c_void_p(); lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))
for nstreams in (1, 2, 3):
    streams = [None]
    for _ in range(nstreams - 1):
        s = C.c_void_p(); _capi.check(lib.bt709hip_stream_create(h, C.byref(s))); streams.append(s.value)
    joins = [C.c_void_p() for _ in streams[1:]]
    for j in joins: lib.bt709hip_event_create(h, C.byref(j))

    def run(n):
        for k in range(n):
            i = k % ring
            _capi.check(lib.bt709hip_unconvert(dec._handle, src.ptr + i * pitch, W * 4, W, H, C.byref(surfs[i]), streams[k % nstreams], 0))
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        run(32)
        for s in streams: lib.bt709hip_stream_synchronize(h, s)
    lib.bt709hip_event_record(h, e0, None)
    run(steps)
    for s, j in zip(streams[1:], joins):
        lib.bt709hip_event_record(h, j, s); lib.bt709hip_stream_wait_event(h, None, j)
    lib.bt709hip_event_record(h, e1, None)
    for s in streams: lib.bt709hip_stream_synchronize(h, s)
    ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    us = ms.value * 1e3 / steps
    print("unconvert 4K, ring %d, one frame per call, %d stream(s): %.2f us per frame  %.1f Gpixel/s  %.3f of 8 TB/s (8 B per pixel)  %s"
          % (ring, nstreams, us, W * H / us / 1e3, 8 * W * H / us / 1e3 / 8000, lib.bt709hip_last_kernel_name().decode()))

# round 5: bt709hip_unconvert_batch -- the same frames, `per` of them per launch (evenly spaced), one stream
for per in (8, 32):
    n = ring - ring % per
    ptrs = (C.c_void_p * ring)(*[src.ptr + i * pitch for i in range(ring)])
    sarr = (_capi.Surface * ring)(*surfs)

    def run_batch(launches):
        for k in range(launches):
            first = (k * per) % n
            pp = C.cast(C.byref(ptrs, first * C.sizeof(C.c_void_p)), C.POINTER(C.c_void_p))
            sp = C.cast(C.byref(sarr, first * C.sizeof(_capi.Surface)), C.POINTER(_capi.Surface))
            _capi.check(lib.bt709hip_unconvert_batch(dec._handle, per, pp, W * 4, W, H, sp, None, 0))
    run_batch(8)
    lib.bt709hip_stream_synchronize(h, None)
    launches = max(8, steps // per)
    lib.bt709hip_event_record(h, e0, None)
    run_batch(launches)
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    us = ms.value * 1e3 / (launches * per)
    print("unconvert 4K, ring %d, %d frames per call (bt709hip_unconvert_batch), 1 stream: %.2f us per frame  %.1f Gpixel/s  %.3f of 8 TB/s (8 B per pixel)  %s"
          % (ring, per, us, W * H / us / 1e3, 8 * W * H / us / 1e3 / 8000, lib.bt709hip_last_kernel_name().decode()))
