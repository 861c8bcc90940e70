#!/usr/bin/env python3
"""Lab: per-launch time of one ring launch as a function of how long the GPU has been kept busy -- n back-to-back launches
between two HIP events for n = 1, 2, 4, ... (each series after `idle` seconds of idle GPU), for the 2:1 kernel (VALU-bound) and
the 1:1 kernel (HBM-bound).  If the per-launch time grows with n, a sustained figure and a burst figure (a rocprof trace of
separated dispatches, a 15 ms placement probe) are different numbers.
usage: python tools/burst_vs_sustained.py [8k-half|4k] [frames=16] [idle=0.5]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "8k-half"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
W, H, half = (7680, 4320, True) if wl == "8k-half" else (3840, 2160, False)
ctx = mb.MetalRenderContext(0)
assert ctx.setupMetal()
lib, h = ctx.lib, ctx.handle
dec = mb.MetalBT709Decoder()
dec.metalRenderContext = ctx
assert dec.setupMetal()
ring = mb.FrameRing(dec, (W, H), frames, halfScale=half, tries=1)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0))
lib.bt709hip_event_create(h, C.byref(e1))
bytes_per_launch = (W * H * 3 // 2 + (W * H if half else W * H * 4)) * frames


def series(n):
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(n):
        assert ring.decode()
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    return ms.value


series(50)  # page tables, code, clocks once
print("%s, %d frames per launch; per-launch us by series length (each series after %.1f s idle):" % (wl, frames, idle))
n = 1
while True:
    time.sleep(idle)
    ms = series(n)
    print("  n %5d  total %9.2f ms  per launch %9.2f us  %7.1f GB/s" % (n, ms, ms * 1e3 / n, bytes_per_launch * n / ms / 1e6), flush=True)
    if ms > 2000.0 or n >= 16384:
        break
    n *= 2
# and a long series sampled in windows: per-launch time over the first 20 ms, ..., of ONE 2 s run
time.sleep(idle)
t0 = time.perf_counter()
marks = []
evs = []
total = int(2000.0 / (ms / n)) if ms > 0 else 1000
for k in range(total):
    if k % max(1, total // 20) == 0:
        e = C.c_void_p()
        lib.bt709hip_event_create(h, C.byref(e))
        lib.bt709hip_event_record(h, e, None)
        evs.append((k, e))
    assert ring.decode()
e = C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e))
lib.bt709hip_event_record(h, e, None)
evs.append((total, e))
lib.bt709hip_stream_synchronize(h, None)
print("one run of %d launches, per-launch us in 20 consecutive windows:" % total)
out = []
for (k0, a), (k1, b) in zip(evs, evs[1:]):
    m = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, a, b, C.byref(m))
    out.append("%.1f" % (m.value * 1e3 / (k1 - k0)))
print("  " + " ".join(out))
