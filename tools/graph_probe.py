#!/usr/bin/env python3
"""Per-frame cost of small-frame pipelines: N single-frame 1080p decodes issued one launch at a
time through the C ABI vs recorded once into a graph (bt709hip_graph_*) and replayed.

    python tools/graph_probe.py [--frames 64] [--width 1920 --height 1080]
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

import gpu_helpers as gh  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    W, H, n = args.width, args.height, args.frames
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    bufs, texs = [], []
    for i in range(n):
        y, c = gh.random_nv12(W, H, seed=i)
        bufs.append(gh.make_buffer(y, c, dec.gamma))
        texs.append(ctx.makeBGRATexture((W, H)))
    frames = [b.frame() for b in bufs]
    surfs = [t.surface() for t in texs]
    cb = ctx.commandQueue.commandBuffer(new_stream=True)

    def direct():
        for f, s in zip(frames, surfs):
            _capi.check(lib.bt709hip_decode(dec._handle, C.byref(f), None, C.byref(s), W, H, cb.stream, 0))

    cb.beginRecording()
    direct()
    rec = cb.endRecording()

    def timed(fn):
        for _ in range(20):
            fn()
        cb.waitUntilCompleted()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        t_issue = time.perf_counter() - t0
        cb.waitUntilCompleted()
        t_all = time.perf_counter() - t0
        return t_issue / (args.reps * n) * 1e6, t_all / (args.reps * n) * 1e6

    d_issue, d_all = timed(direct)
    g_issue, g_all = timed(lambda: rec.replay(cb))
    print(json.dumps({"workload": "%d single-frame %dx%d decodes per step" % (n, W, H),
                      "direct_us_per_frame": {"host_issue": round(d_issue, 2), "wall": round(d_all, 2)},
                      "graph_us_per_frame": {"host_issue": round(g_issue, 2), "wall": round(g_all, 2)},
                      "wall_speedup": round(d_all / g_all, 2)}))


if __name__ == "__main__":
    main()
