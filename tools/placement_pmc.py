#!/usr/bin/env python3
"""Lab: what do the memory-side counters say about a slow output slab?  N output slabs, one input slab, K launches of the 256-frame
decode on each (nothing else launches that kernel), the launch's rate printed per slab.  Run under rocprofv3 --pmc (one process =
one placement = one pass; tools/placement_pmc.sh pairs the counters of dispatch k with slab k // K).
usage: python tools/placement_pmc.py [n=8] [k=4]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
import time  # noqa: E402
# clocks up before slab 0 is measured (the warm launches go through another kernel so that the dispatch numbering stays simple)
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    half = (ob * RING // 2) // 4096 * 4096
    _capi.check(lib.bt709hip_copy_probe(h, outs[0].value + half, outs[0].value, half, None))
    lib.bt709hip_stream_synchronize(h, None)
for k, d_out in enumerate(outs):
    frames, surfs = (Frame * RING)(), (Surface * RING)()
    for i in range(RING):
        b = d_in.value + i * in_stride
        frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
        surfs[i] = Surface(d_out.value + i * ob, W * 4, W, H)
    _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 1))  # launch 0 of the slab: warm
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(K - 1):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    print("SLAB %d out 0x%x frac %.4f" % (k, d_out.value, (K - 1) * RING * W * H * 5.5 / (ms.value / 1e3) / 8e12), flush=True)
