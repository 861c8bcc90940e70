#!/usr/bin/env python3
"""Lab: at what granularity does placement act?  N separate allocations of S MiB each; fill rate (bt709hip_memset) of every one.
usage: python tools/placement_chunks.py [n=64] [mib=512]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = (int(sys.argv[2]) if len(sys.argv) > 2 else 512) << 20
ctx = mb.MetalRenderContext(0)
assert ctx.setupMetal()
lib, h = ctx.lib, ctx.handle
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0))
lib.bt709hip_event_create(h, C.byref(e1))
chunks = []
for _ in range(N):
    p = C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, S, C.byref(p)))
    chunks.append(p)


def fill_rate(p, reps=8):
    for _ in range(2):
        lib.bt709hip_memset(h, p, 0, S, None)
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(reps):
        lib.bt709hip_memset(h, p, 0, S, None)
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    return reps * S / (ms.value / 1e3) / 1e9


for rnd in range(2):
    rates = [fill_rate(p) for p in chunks]
    print("pass %d, fill GB/s per chunk:" % rnd)
    for i in range(0, N, 16):
        print("  " + " ".join("%5.0f" % r for r in rates[i:i + 16]))
    srt = sorted(rates)
    print("  min %.0f  p25 %.0f  median %.0f  p75 %.0f  max %.0f" % (srt[0], srt[N // 4], srt[N // 2], srt[3 * N // 4], srt[-1]), flush=True)
