#!/usr/bin/env python3
"""Throughput of the any-ratio fused decode + bilinear rescale (bt709hip_decode_scaled, SURVEY 8(f)
row 4) on resident frames; same method as bench.py (ring in HBM, HIP events on the launch stream).
One frame per launch (the view-fit path of the reference decodes one frame per command buffer).

    python tools/bench_scaled.py [--width 3840 --height 2160 --out-width 2560 --out-height 1440]
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

import gpu_helpers as gh  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ring", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--out-width", type=int, default=2560)
    ap.add_argument("--out-height", type=int, default=1440)
    args = ap.parse_args()
    W, H, OW, OH = args.width, args.height, args.out_width, args.out_height
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    bufs, texs = [], []
    for i in range(args.ring):
        y, c = gh.random_nv12(W, H, seed=0x709 + i)
        bufs.append(gh.make_buffer(y, c, dec.gamma))
        texs.append(ctx.makeBGRATexture((OW, OH)))
    frames = [b.frame() for b in bufs]
    surfs = [t.surface() for t in texs]

    def step():
        for f, s in zip(frames, surfs):
            _capi.check(lib.bt709hip_decode_scaled(dec._handle, C.byref(f), C.byref(s), None, 0))

    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        step()
        _capi.check(lib.bt709hip_stream_synchronize(h, None))
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
    us = ms.value * 1e3 / (args.steps * args.ring)
    nbytes = W * H * 3 // 2 + OW * OH * 4
    print(json.dumps({"workload": "%dx%d NV12 -> %dx%d BGRA, fused decode + bilinear rescale, 1 frame per launch" % (W, H, OW, OH),
                      "us_per_frame": round(us, 3), "out_gpixel_per_s": round(OW * OH / us / 1e3, 1),
                      "algorithmic_GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 4),
                      "kernel": lib.bt709hip_last_kernel_name().decode()}))


if __name__ == "__main__":
    main()
