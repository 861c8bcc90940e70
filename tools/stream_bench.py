#!/usr/bin/env python3
"""BASELINE config 2 as a real stream: host frames in, host frames out, one HIP stream per
in-flight frame (upload Y + CbCr with hipMemcpy2DAsync from pinned memory, decode, download
BGRA), `--inflight` frames pipelined.  Reports the PCIe-INCLUSIVE rate, which is never the
headline `value` of bench.py (that one is HBM-resident); DESIGN.md quotes this number.

    python tools/stream_bench.py [--width 1920 --height 1080] [--frames 600] [--inflight 4]
"""
import argparse
import ctypes as C
import json
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--check-every", type=int, default=100, help="byte-compare every Nth frame with the oracle")
    args = ap.parse_args()
    W, H, K = args.width, args.height, args.inflight
    ctx = mb.MetalRenderContext(0)
    assert ctx.setupMetal()
    lib, h = ctx.lib, ctx.handle
    dec = mb.MetalBT709Decoder()
    dec.metalRenderContext = ctx
    assert dec.setupMetal()

    in_bytes, out_bytes = W * H * 3 // 2, W * H * 4
    slots = []
    for k in range(K):
        s = C.c_void_p()
        _capi.check(lib.bt709hip_stream_create(h, C.byref(s)))
        hin, hout, din, dout = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _capi.check(lib.bt709hip_host_alloc(h, in_bytes, C.byref(hin)))
        _capi.check(lib.bt709hip_host_alloc(h, out_bytes, C.byref(hout)))
        _capi.check(lib.bt709hip_malloc(h, in_bytes, C.byref(din)))
        _capi.check(lib.bt709hip_malloc(h, out_bytes, C.byref(dout)))
        hin_np = np.ctypeslib.as_array(C.cast(hin, C.POINTER(C.c_uint8)), shape=(in_bytes,))
        hout_np = np.ctypeslib.as_array(C.cast(hout, C.POINTER(C.c_uint8)), shape=(out_bytes,))
        frame = Frame(din.value, W, din.value + W * H, W, W, H, 1, 1)
        surf = Surface(dout.value, W * 4, W, H)
        slots.append(dict(stream=s, hin=hin, hout=hout, din=din, dout=dout, hin_np=hin_np, hout_np=hout_np,
                          frame=frame, surf=surf, pending=None))

    # a ring of distinct source frames in ordinary host memory (the "decoder output" of a real pipeline)
    ring = [np.random.default_rng(0x709 + i).integers(0, 256, in_bytes, dtype=np.uint8) for i in range(16)]
    oracle = None
    if args.check_every:
        from oracle_lib import Oracle
        oracle = Oracle()
    checked = 0

    def retire(slot):
        nonlocal checked
        _capi.check(lib.bt709hip_stream_synchronize(h, slot["stream"]))
        n = slot["pending"]
        if oracle is not None and n % args.check_every == 0:
            src = ring[n % len(ring)]
            want = oracle.decode_nv12(0, src[:W * H].reshape(H, W)[:16], src[W * H:].reshape(H // 2, W)[:8])
            assert np.array_equal(slot["hout_np"][:16 * W * 4].reshape(16, W * 4), want), "frame %d differs" % n
            checked += 1
        slot["pending"] = None

    def submit(slot, n):
        slot["hin_np"][:] = ring[n % len(ring)]  # host copy into pinned memory (part of a real pipeline too)
        s = slot["stream"]
        _capi.check(lib.bt709hip_upload(h, slot["din"], in_bytes, slot["hin"], in_bytes, in_bytes, 1, s))
        rc = lib.bt709hip_decode(dec._handle, C.byref(slot["frame"]), None, C.byref(slot["surf"]), W, H, s, 0)
        _capi.check(rc, "decode")
        _capi.check(lib.bt709hip_download(h, slot["hout"], out_bytes, slot["dout"], out_bytes, out_bytes, 1, s))
        slot["pending"] = n

    for n in range(2 * K):  # warm-up
        slot = slots[n % K]
        if slot["pending"] is not None:
            retire(slot)
        submit(slot, n)
    for slot in slots:
        if slot["pending"] is not None:
            retire(slot)
    t0 = time.perf_counter()
    for n in range(args.frames):
        slot = slots[n % K]
        if slot["pending"] is not None:
            retire(slot)
        submit(slot, n)
    for slot in slots:
        if slot["pending"] is not None:
            retire(slot)
    dt = time.perf_counter() - t0
    fps = args.frames / dt
    print(json.dumps({
        "workload": "%dx%d NV12 host -> GPU decode -> BGRA host, %d frames, %d in flight (one HIP stream each)"
                    % (W, H, args.frames, K),
        "fps": round(fps, 1), "gpixel_per_s_pcie_inclusive": round(fps * W * H / 1e9, 3),
        "pcie_GBps_up_plus_down": round(fps * (in_bytes + out_bytes) / 1e9, 2),
        "frames_checked_against_oracle": checked, "realtime_60fps_streams": round(fps / 60.0, 1)}))


if __name__ == "__main__":
    main()
