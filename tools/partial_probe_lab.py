#!/usr/bin/env python3
"""Round 6 lab: does probing a QUARTER of a candidate output slab predict how the whole slab streams?  (bt709hip_ring_create probes
every candidate with the whole ring's launch: 12 x 8.5 GB written and freed per hunt, which the driver wipes at ~40 GB/s in
batches -- profiles/r06_hunt_default.txt.)  For each of N fresh 8.5 GB output candidates under one input slab: the 256-frame 4K
launch's rate over the FIRST 64 frames only (first touch of that quarter), then over the LAST 64 frames, then over all 256."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as gh  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402

ctx = gh.context()
lib, h = ctx.lib, ctx.handle
dec = gh.make_decoder(mb.MetalBT709GammaApple)
W, H, N = 3840, 2160, 256
in_stride = (W * H * 3 // 2 + 255) // 256 * 256
out_stride = W * H * 4
d_in = C.c_void_p()
_capi.check(lib.bt709hip_malloc(h, in_stride * N, C.byref(d_in)))
lib.bt709hip_memset(h, d_in, 0x55, in_stride * N, None)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.bt709hip_event_create(h, C.byref(e0))
lib.bt709hip_event_create(h, C.byref(e1))
frames = (_capi.Frame * N)()
for i in range(N):
    base = d_in.value + i * in_stride
    frames[i] = _capi.Frame(base, W, base + W * H, W, W, H, 1, 1)


def rate(d_out, first, count, reps):
    surfs = (_capi.Surface * count)()
    for i in range(count):
        surfs[i] = _capi.Surface(d_out + (first + i) * out_stride, W * 4, W, H, 0, 0)
    fp = C.cast(C.byref(frames, first * C.sizeof(_capi.Frame)), C.POINTER(_capi.Frame))
    for _ in range(3):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, count, fp, None, surfs, None, 1))
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(reps):
        _capi.check(lib.bt709hip_decode_batch(dec._handle, count, fp, None, surfs, None, 0))
    lib.bt709hip_event_record(h, e1, None)
    lib.bt709hip_event_synchronize(h, e1)
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    return count * (W * H * 5.5) * reps / (ms.value * 1e-3) / 1e9


rows = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    p = C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, out_stride * N, C.byref(p)))
    q_first = rate(p.value, 0, 64, 24)
    q_last = rate(p.value, 192, 64, 24)
    full = rate(p.value, 0, 256, 8)
    q_first_again = rate(p.value, 0, 64, 24)
    rows.append((q_first, q_last, full, q_first_again))
    print("candidate %2d: first quarter %6.0f  last quarter %6.0f  whole %6.0f  first quarter again %6.0f GB/s" % (k, q_first, q_last, full, q_first_again))
    lib.bt709hip_free(h, p)
best_full = max(range(len(rows)), key=lambda i: rows[i][2])
best_q = max(range(len(rows)), key=lambda i: rows[i][0])
print("whole-slab winner: %d (%.0f); first-quarter winner: %d (whole %.0f = %.1f %% below)" % (best_full, rows[best_full][2], best_q, rows[best_q][2], 100 * (1 - rows[best_q][2] / rows[best_full][2])))
