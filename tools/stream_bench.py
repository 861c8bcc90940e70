#!/usr/bin/env python3
"""BASELINE config 2 as a real stream: host frames in, host frames out through the C ABI's frame
pool (bt709hip_pool_*: one HIP stream per in-flight frame, pinned staging, upload -> decode ->
download per slot), `--inflight` frames pipelined.  Reports the PCIe-INCLUSIVE rate, which is never the
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
    pool = mb.InFlightFramePool(dec, (W, H), K)   # bt709hip_pool_*: per slot a stream, pinned staging, device buffers

    # a ring of distinct source frames in ordinary host memory (the "decoder output" of a real pipeline)
    ring = [np.random.default_rng(0x709 + i).integers(0, 256, in_bytes, dtype=np.uint8) for i in range(16)]
    oracle = None
    if args.check_every:
        from oracle_lib import Oracle
        oracle = Oracle()
    checked = 0
    pending = []  # (frame number, slot), oldest first

    def retire():
        nonlocal checked
        n, slot = pending.pop(0)
        out = pool.wait(slot)
        if oracle is not None and n % args.check_every == 0:
            src = ring[n % len(ring)]
            want = oracle.decode_nv12(0, src[:W * H].reshape(H, W)[:16], src[W * H:].reshape(H // 2, W)[:8])
            assert np.array_equal(out[:16], want), "frame %d differs" % n
            checked += 1

    def submit(n):
        if len(pending) == K:
            retire()
        slot, ybuf, cbuf = pool.acquire()
        src = ring[n % len(ring)]
        ybuf[:] = src[:W * H].reshape(H, W)              # host copy into pinned memory (part of a real pipeline too)
        cbuf[:] = src[W * H:].reshape(H // 2, W)
        pool.submit(slot)
        pending.append((n, slot))

    for n in range(2 * K):  # warm-up
        submit(n)
    while pending:
        retire()
    t0 = time.perf_counter()
    for n in range(args.frames):
        submit(n)
    while pending:
        retire()
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
