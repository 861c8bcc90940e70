#!/usr/bin/env python3
"""Lab: does the ORDER or GROUPING of the two slabs' allocations decide the placement regime of a fresh process's first ring?
One fresh process per line: mode = in_out (input first: what every path does), out_in (output first), one (ONE allocation holding
both: input at its start, output behind it, 2 MiB aligned), spaced (a 3 GB allocation between the two, freed or kept).  The same
banded 256-frame 4K launch is timed on it.  usage: python tools/alloc_order_lab.py MODE"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "in_out"
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


def malloc(n):
    p = C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, n, C.byref(p)))
    return p.value


if MODE == "in_out":
    d_in, d_out = malloc(in_bytes), malloc(out_bytes)
elif MODE == "out_in":
    d_out, d_in = malloc(out_bytes), malloc(in_bytes)
elif MODE == "one":
    gap = (in_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    d_in = malloc(gap + out_bytes)
    d_out = d_in + gap
elif MODE == "spaced_kept":
    d_in, spacer, d_out = malloc(in_bytes), malloc(3 << 30), malloc(out_bytes)
elif MODE == "spaced_freed":
    d_in, spacer = malloc(in_bytes), malloc(3 << 30)
    d_out = malloc(out_bytes)
    lib.bt709hip_free(h, spacer)
elif MODE.startswith("far"):  # far32 / far96 / far160: a spacer of that many GB between the two, freed afterwards
    d_in, spacer = malloc(in_bytes), malloc(int(MODE[3:]) << 30)
    d_out = malloc(out_bytes)
    lib.bt709hip_free(h, spacer)
else:
    raise SystemExit("unknown mode")
frames, surfs = (Frame * RING)(), (Surface * RING)()
for i in range(RING):
    b = d_in + i * in_stride
    frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
    surfs[i] = Surface(d_out + i * ob, W * 4, W, H)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))


def run(n):
    for _ in range(n):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))


t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end:
    run(1); lib.bt709hip_stream_synchronize(h, None)
rates = []
for _ in range(3):
    lib.bt709hip_event_record(h, e0, None); run(20); lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_stream_synchronize(h, None)
    ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    rates.append(20 * RING * W * H / (ms.value / 1e3) / 1e9)
print("%-13s in 0x%x out 0x%x  %s Gpixel/s  frac %.4f" % (MODE, d_in, d_out, " ".join("%.1f" % r for r in rates), sorted(rates)[1] * 5.5 / 8000), flush=True)
