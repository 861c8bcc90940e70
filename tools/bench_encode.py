#!/usr/bin/env python3
"""Throughput of the GPU encoder (BGRA -> NV12) on resident 4K frames.
Not the headline bench (that is bench.py); same method: ring of distinct frames in HBM,
HIP events on the launch stream, algorithmic bytes = 4 B read + 1.5 B written per pixel.

    python tools/bench_encode.py [--ring 32] [--steps 50] [--frames-per-launch 16]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ring", type=int, default=32)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=0, help="extra untimed steps after the 0.4 s pre-warm")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--threads", type=int, default=0, help="bt709hip_context_option ENCODE_THREADS (0 = from the width)")
    ap.add_argument("--row-pairs", type=int, default=0, help="bt709hip_context_option ENCODE_ROW_PAIRS (0 = sized per launch)")
    ap.add_argument("--xcd-bands", type=int, default=1, help="bt709hip_context_option XCD_BANDS")
    ap.add_argument("--placement-tries", type=int, default=1, help="bt709hip_malloc_streaming candidates per slab")
    ap.add_argument("--content", choices=("random", "smooth", "flat"), default="random",
                    help="random bytes (worst case for the LDS gathers), a smooth gradient, or one colour per picture")
    ap.add_argument("--frames-per-launch", type=int, default=1,
                    help="> 1: bt709hip_encode_batch over a ring carved from one allocation")
    ap.add_argument("--library", default=None, help="a variant build of libbt709hip.so (python -m metalbt709decoder_amd.build --variant)")
    args = ap.parse_args()
    W, H = args.width, args.height
    if args.library:
        from metalbt709decoder_amd import _capi as _c
        _c.load(os.path.abspath(args.library))
    ctx = mb.MetalRenderContext(0)
    assert ctx.setupMetal()
    lib, h = ctx.lib, ctx.handle
    from metalbt709decoder_amd import _capi
    _capi.check(lib.bt709hip_context_set_option(h, _capi.CTX_OPT_ENCODE_THREADS, args.threads))
    _capi.check(lib.bt709hip_context_set_option(h, _capi.CTX_OPT_ENCODE_ROW_PAIRS, args.row_pairs))
    _capi.check(lib.bt709hip_context_set_option(h, _capi.CTX_OPT_XCD_BANDS, args.xcd_bands))
    rng = np.random.default_rng(0x709)
    from metalbt709decoder_amd.decoder import DeviceBuffer
    fpl = max(1, args.frames_per_launch)
    args.ring = (args.ring + fpl - 1) // fpl * fpl
    in_pitch, out_pitch = W * H * 4, W * H * 3 // 2
    slab_in = DeviceBuffer(ctx, args.ring * in_pitch, args.placement_tries)
    slab_out = DeviceBuffer(ctx, args.ring * out_pitch, args.placement_tries)
    texs, bufs = [], []
    for i in range(args.ring):
        t = mb.BGRATexture(ctx, W, H, W * 4, ptr=slab_in.ptr + i * in_pitch)
        if args.content == "random":
            px = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        elif args.content == "flat":
            px = np.full(W * H, int(rng.integers(0, 1 << 32)), dtype=np.uint32)
        else:
            yy, xx = np.mgrid[0:H, 0:W].astype(np.uint32)
            px = ((((xx + i) >> 4) & 255) | ((((yy + 2 * i) >> 3) & 255) << 8) | ((((xx + yy) >> 5) & 255) << 16) | (255 << 24)).astype(np.uint32).ravel()
        ctx.fillBGRATexture(t, px)
        texs.append(t)
        base = slab_out.ptr + i * out_pitch
        bufs.append(mb.CVPixelBuffer(ctx, W, H, W, W, planes=(base, base + W * H)))
    surfs = (_capi.Surface * args.ring)(*[t.surface() for t in texs])
    frames = (_capi.Frame * args.ring)(*[b.frame() for b in bufs])
    s_size, f_size = C.sizeof(_capi.Surface), C.sizeof(_capi.Frame)

    def step():
        for i in range(0, args.ring, fpl):
            _capi.check(lib.bt709hip_encode_batch(h, fpl, C.cast(C.byref(surfs, i * s_size), C.POINTER(_capi.Surface)),
                                                  C.cast(C.byref(frames, i * f_size), C.POINTER(_capi.Frame)),
                                                  1, 0, None, 0))

    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        step()
        _capi.check(lib.bt709hip_stream_synchronize(h, None))
    for _ in range(args.warmup):
        step()
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.bt709hip_event_create(h, C.byref(e0))
    lib.bt709hip_event_create(h, C.byref(e1))
    lib.bt709hip_event_record(h, e0, None)
    for _ in range(args.steps):
        step()
    lib.bt709hip_event_record(h, e1, None)
    _capi.check(lib.bt709hip_stream_synchronize(h, None))
    ms = C.c_float()
    lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    n = args.steps * args.ring
    us = ms.value * 1e3 / n
    bytes_per_frame = W * H * 4 + W * H * 3 // 2
    print(json.dumps({"workload": "%dx%d BGRA -> NV12 encode (sRGB in, Apple gamma out), %d frame(s) per launch, %s content" % (W, H, fpl, args.content),
                      "us_per_frame": round(us, 3), "gpixel_per_s": round(W * H / us / 1e3, 1),
                      "algorithmic_GBps": round(bytes_per_frame / us / 1e3, 1),
                      "frac_of_8TBps": round(bytes_per_frame / us / 1e3 / 8000, 4),
                      "kernel": lib.bt709hip_last_kernel_name().decode(),
                      "placement": [slab_in.placement, slab_out.placement]}))


if __name__ == "__main__":
    main()
