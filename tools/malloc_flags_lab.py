#!/usr/bin/env python3
"""Lab: does the KIND of device allocation move the streaming rate of the 1:1 launch?  The placement hunt found
0.74-0.81 between plain hipMalloc allocations of one process; this tries the allocator's other flavours for the ring's slabs --
hipExtMallocWithFlags(default / fine-grained / uncached / physically contiguous) -- under the same 256-frame launch, in one process,
each flavour allocated REPS times (so that flavour and placement can be told apart).

    python tools/malloc_flags_lab.py [--ring 256] [--reps 3]"""
import argparse
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

FLAVOURS = [("hipMalloc", None), ("ext default", 0x0), ("ext fine-grained", 0x1), ("ext uncached", 0x3), ("ext contiguous", 0x4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ring", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="", help="comma list of flavour names to run")
    args = ap.parse_args()
    W, H, RING = 3840, 2160, args.ring
    ctx = mb.MetalRenderContext(0)
    assert ctx.setupMetal()
    lib, h = ctx.lib, ctx.handle
    dec = mb.MetalBT709Decoder()
    dec.metalRenderContext = ctx
    assert dec.setupMetal()
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    yb, cb, ob = W * H, W * H // 2, W * H * 4
    in_stride = (yb + cb + 255) // 256 * 256
    rng = np.random.default_rng(1)
    buf = rng.integers(0, 256, (1, yb + cb), dtype=np.uint8)
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.bt709hip_event_create(h, C.byref(e0))
    lib.bt709hip_event_create(h, C.byref(e1))

    def alloc(flag, nbytes):
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), nbytes) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag)
        return p if rc == 0 else None

    only = [s.strip() for s in args.only.split(",") if s.strip()]
    for rep in range(args.reps):
        for name, flag in FLAVOURS:
            if only and name not in only:
                continue
            d_in, d_out = alloc(flag, in_stride * RING), alloc(flag, ob * RING)
            if d_in is None or d_out is None:
                print("%-18s allocation refused" % name, flush=True)
                for p in (d_in, d_out):
                    if p is not None:
                        hip.hipFree(p)
                continue
            for i in range(RING):
                _capi.check(lib.bt709hip_upload(h, d_in.value + i * in_stride, buf.shape[1], buf.ctypes.data, buf.shape[1], buf.shape[1], 1, None))
                lib.bt709hip_stream_synchronize(h, None)  # the upload is asynchronous and `buf` is replaced in the next round
            _capi.check(lib.bt709hip_stream_synchronize(h, None))
            frames, surfs = (Frame * RING)(), (Surface * RING)()
            for i in range(RING):
                b = d_in.value + i * in_stride
                frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
                surfs[i] = Surface(d_out.value + i * ob, W * 4, W, H)

            def run(n):
                for _ in range(n):
                    _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                run(1)
                lib.bt709hip_stream_synchronize(h, None)
            rates = []
            for _ in range(3):
                lib.bt709hip_event_record(h, e0, None)
                run(20)
                lib.bt709hip_event_record(h, e1, None)
                lib.bt709hip_stream_synchronize(h, None)
                ms = C.c_float()
                lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
                rates.append(20 * RING * W * H / (ms.value / 1e3) / 1e9)
            half = (ob * RING // 2) // 4096 * 4096
            lib.bt709hip_event_record(h, e0, None)
            for _ in range(8):
                _capi.check(lib.bt709hip_copy_probe(h, d_out.value + half, d_out.value, half, None))
            lib.bt709hip_event_record(h, e1, None)
            lib.bt709hip_stream_synchronize(h, None)
            ms = C.c_float()
            lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
            copy = 8 * 2 * half / (ms.value / 1e3) / 1e9
            print("%-18s in 0x%x out 0x%x  decode %s Gpixel/s (%.4f)  copy %.0f GB/s" % (
                name, d_in.value, d_out.value, " ".join("%.1f" % r for r in rates), sorted(rates)[1] * 5.5 / 8000, copy), flush=True)
            hip.hipFree(d_in)
            hip.hipFree(d_out)


if __name__ == "__main__":
    main()
