#!/usr/bin/env python3
"""Throughput of the paths next to the headline decode, on resident frames; same method as bench.py
(a ring carved from one allocation in HBM, HIP events on the launch stream, median of 5 regions):

    --path scaled     any-ratio fused decode + bilinear rescale (bt709hip_decode_scaled[_batch])
    --path rgba16f    pass 1 into an RGBA16Float target (bt709hip_decode_batch, format RGBA16F)
    --path render8    pass 2 alone from a BGRA8 sRGB intermediate (bt709hip_render_scaled[_batch])
    --path render16   pass 2 alone from an RGBA16Float intermediate

    python tools/bench_scaled.py [--path scaled --width 3840 --height 2160 --out-width 2560 --out-height 1440
                                  --frames-per-launch 8]
Takes --steps / --warmup like bench.py, so tools/profile_gpu.sh can drive it (PROFILE_PROG=tools/bench_scaled.py).
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
from metalbt709decoder_amd.decoder import DeviceBuffer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", default="scaled", choices=["scaled", "rgba16f", "render8", "render16"])
    ap.add_argument("--ring", type=int, default=64,
                    help="frames in the ring; 64 x 12.4 MB of 4K input is 3x the 256 MB Infinity Cache (a ring of 16 fitted it: round 3 found the RGBA16F path's 0.76 to be cache hits)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--out-width", type=int, default=2560)
    ap.add_argument("--out-height", type=int, default=1440)
    ap.add_argument("--frames-per-launch", type=int, default=1)
    ap.add_argument("--xcd-bands", type=int, default=1, help="decoder option BT709HIP_OPT_XCD_BANDS")
    ap.add_argument("--gamma", default="apple", choices=["apple", "srgb", "linear", "itu709"])
    ap.add_argument("--placement-tries", type=int, default=0, help="--path rgba16f: candidates per slab of the frame ring's hunt (0 = the product's default)")
    ap.add_argument("--library", default=None, help="a variant build of libbt709hip.so (python -m metalbt709decoder_amd.build --variant)")
    args = ap.parse_args()
    if args.library:
        _capi.load(os.path.abspath(args.library))
    W, H, path = args.width, args.height, args.path
    OW, OH = (W, H) if path == "rgba16f" else (args.out_width, args.out_height)
    fpl = max(1, min(args.frames_per_launch, args.ring))
    ring = args.ring - args.ring % fpl
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    gamma = {"apple": mb.MetalBT709GammaApple, "srgb": mb.MetalBT709GammaSRGB, "linear": mb.MetalBT709GammaLinear,
             "itu709": mb.MetalBT709GammaITU709}[args.gamma]
    dec = gh.make_decoder(gamma, options={_capi.OPT_XCD_BANDS: args.xcd_bands})

    in_px = {"render8": 4, "render16": 8}.get(path)          # bytes per input texel of pass 2 alone
    out_px = 8 if path == "rgba16f" else 4
    in_pitch = (W * H * (in_px or 0) if in_px else W * H * 3 // 2 + 255) // 256 * 256
    out_pitch = (OW * OH * out_px + 255) // 256 * 256
    frames, surfs, inters = (_capi.Frame * ring)(), (_capi.Surface * ring)(), (_capi.Surface * ring)()
    rng = np.random.default_rng(0x709)
    frame_ring = None
    if path == "rgba16f":
        # the product's own placement: a frame ring with RGBA16Float targets, hunted with the RGBA16F launch as the probe
        # (bt709hip_ring_options.format; rounds 2-5a took two single-slab allocations and this pattern's 0.70 / 0.77 lottery)
        frame_ring = mb.FrameRing(dec, (W, H), ring, tries=args.placement_tries, pixelFormat=mb.MTLPixelFormatRGBA16Float)
        for i in range(ring):
            y, c = gh.random_nv12(W, H, seed=0x709 + i)
            frame_ring.pixelBuffer(i).upload_planes(y, c)
            f, o = _capi.Frame(), _capi.Surface()
            _capi.check(lib.bt709hip_ring_frame(frame_ring.handle, i, C.byref(f), None, C.byref(o)), "ring frame")
            frames[i], surfs[i] = f, o
        slab_in = slab_out = None
    else:
        slab_in, slab_out = DeviceBuffer(ctx, ring * in_pitch), DeviceBuffer(ctx, ring * out_pitch)
    for i in range(ring if frame_ring is None else 0):
        base = slab_in.ptr + i * in_pitch
        if in_px:  # an intermediate as pass 1 leaves it (random bytes / random halves in [0, 1])
            if in_px == 4:
                buf = rng.integers(0, 256, (H, W * 4), dtype=np.uint8)
            else:
                buf = rng.random((H, W * 4), dtype=np.float32).astype(np.float16).view(np.uint8)
            ctx._upload(base, W * in_px, buf, None)
            inters[i] = _capi.Surface(base, W * in_px, W, H, _capi.FORMAT_RGBA16F if in_px == 8 else _capi.FORMAT_BGRA8_SRGB, 0)
        else:
            y, c = gh.random_nv12(W, H, seed=0x709 + i)
            ctx._upload(base, W, y, None)
            ctx._upload(base + W * H, W, c, None)
            frames[i] = _capi.Frame(base, W, base + W * H, W, W, H, 1, gh.TRANSFER_FOR_GAMMA[dec.gamma])
        ctx._sync(None)
        surfs[i] = _capi.Surface(slab_out.ptr + i * out_pitch, OW * out_px, OW, OH,
                                 _capi.FORMAT_RGBA16F if path == "rgba16f" else _capi.FORMAT_BGRA8_SRGB, 0)
    fsz, ssz = C.sizeof(_capi.Frame), C.sizeof(_capi.Surface)

    def step():
        for i in range(0, ring, fpl):
            fp = C.cast(C.byref(frames, i * fsz), C.POINTER(_capi.Frame))
            sp = C.cast(C.byref(surfs, i * ssz), C.POINTER(_capi.Surface))
            if path == "scaled":
                rc = lib.bt709hip_decode_scaled_batch(dec._handle, fpl, fp, None, sp, None, 0)
            elif path == "rgba16f":
                rc = lib.bt709hip_decode_batch(dec._handle, fpl, fp, None, sp, None, 0)
            elif fpl == 1:
                rc = lib.bt709hip_render_scaled(h, C.cast(C.byref(inters, i * ssz), C.POINTER(_capi.Surface)), sp, None, 0)
            else:
                rc = lib.bt709hip_render_scaled_batch(h, fpl, C.cast(C.byref(inters, i * ssz), C.POINTER(_capi.Surface)), sp, None, 0)
            _capi.check(rc, path)

    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        step()
        ctx._sync(None)
    for _ in range(args.warmup):
        step()
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.bt709hip_event_create(h, C.byref(e0))
    lib.bt709hip_event_create(h, C.byref(e1))
    regions = []
    for _ in range(5):
        ctx._sync(None)
        lib.bt709hip_event_record(h, e0, None)
        for _ in range(args.steps):
            step()
        lib.bt709hip_event_record(h, e1, None)
        ctx._sync(None)
        ms = C.c_float()
        lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
        regions.append(ms.value)
    regions.sort()
    us = regions[2] * 1e3 / (args.steps * ring)  # per frame, median region
    nbytes = (W * H * in_px if in_px else W * H * 3 // 2) + OW * OH * out_px
    if path.startswith("render"):  # every output pixel reads 4 taps; the algorithmic input is each texel once
        nbytes = min(W * H, 4 * OW * OH) * in_px + OW * OH * 4
    print(json.dumps({"path": path,
                      "workload": "%dx%d -> %dx%d, %d frame(s) per launch, ring %d" % (W, H, OW, OH, fpl, ring),
                      "us_per_frame": round(us, 3), "out_gpixel_per_s": round(OW * OH / us / 1e3, 1),
                      "algorithmic_bytes_per_frame": nbytes,
                      "algorithmic_GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 4),
                      "kernel": lib.bt709hip_last_kernel_name().decode()}))


if __name__ == "__main__":
    main()
