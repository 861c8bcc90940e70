#!/usr/bin/env python3
"""Lab: variant builds of libbt709hip.so against each other in ONE process on ONE ring (placement moves the rate of a launch
by up to 8 % between allocations, so process-against-process runs cannot resolve a 1 % difference).  The ring is allocated once
(placement hunt of the first library); every library then gets its own context and decoder over the same device pointers, and
the timed regions alternate between the libraries.

    python tools/ab_libs.py [--ring 256] [--per-launch 256] [--rounds 4] [--tries 4] LIB [LIB ...]
LIB = a path, or "shipped" for the in-tree build; LIB@K=V,K=V sets decoder options for that entry only."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402
from metalbt709decoder_amd._capi import Frame, Surface  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in _capi.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.bt709hip_abi_version() == _capi.ABI_VERSION, path
    return lib


def ok(lib, st):
    if st != 0:
        raise RuntimeError(lib.bt709hip_strerror(st).decode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ring", type=int, default=256)
    ap.add_argument("--per-launch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tries", type=int, default=4)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--contiguous", action="store_true", help="slabs from hipExtMallocWithFlags(hipDeviceMallocContiguous): one physical range, a deterministic placement")
    ap.add_argument("--in-pad", type=int, default=0, help="bytes added to the spacing of the input frames")
    ap.add_argument("--out-pad", type=int, default=0, help="bytes added to the spacing of the output frames")
    ap.add_argument("--out-shift", type=int, default=0, help="bytes the first output frame sits behind the start of its slab")
    ap.add_argument("--gamma", type=int, default=0, help="MetalBT709Gamma of every decoder: 0 Apple, 1 sRGB, 2 Linear, 3 ITU-709")
    ap.add_argument("--decoder-option", action="append", default=[], metavar="K=V")
    ap.add_argument("--format", default="bgra8", choices=["bgra8", "rgba16f"], help="render target of the launches (rgba16f: 8 bytes per pixel, linear-light halves)")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    W, H, RING = args.width, args.height, args.ring
    specs = [(p.split("@") + [""])[:2] for p in args.libs]  # LIB[@K=V,K=V]: decoder options of that entry only
    paths = [_capi.library_path() if p == "shipped" else os.path.abspath(p) for p, _ in specs]
    loaded = {}
    libs = [loaded.setdefault(p, None) or loaded.__setitem__(p, bind(p)) or loaded[p] for p in paths]
    ctxs, decs = [], []
    for lib, (_, own) in zip(libs, specs):
        h = C.c_void_p()
        ok(lib, lib.bt709hip_context_create(0, C.byref(h)))
        d = C.c_void_p()
        ok(lib, lib.bt709hip_decoder_create(h, args.gamma, 0, C.byref(d)))
        ok(lib, lib.bt709hip_decoder_setup(d))
        for kv in args.decoder_option + [x for x in own.split(",") if x]:
            k, v = kv.split("=")
            ok(lib, lib.bt709hip_decoder_set_option(d, int(k), int(v)))
        ctxs.append(h)
        decs.append(d)
    lib0, h0 = libs[0], ctxs[0]
    opx = 8 if args.format == "rgba16f" else 4  # bytes per output pixel
    yb, cb, ob = W * H, W * H // 2, W * H * opx
    in_stride = (yb + cb + 255) // 256 * 256 + args.in_pad
    out_stride = ob + args.out_pad
    d_in, d_out = C.c_void_p(), C.c_void_p()
    if args.contiguous:
        hip = C.CDLL("libamdhip64.so")
        hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        assert hip.hipExtMallocWithFlags(C.byref(d_in), in_stride * RING, 0x4) == 0
        assert hip.hipExtMallocWithFlags(C.byref(d_out), out_stride * RING + args.out_shift, 0x4) == 0
        print("contiguous slabs: in 0x%x out 0x%x, spacing %d / %d, out shift %d" % (d_in.value, d_out.value, in_stride, out_stride, args.out_shift))
    else:
        rates = (C.c_float * args.tries)()
        chosen = C.c_int()
        ok(lib0, lib0.bt709hip_malloc_streaming(h0, in_stride * RING, args.tries, C.byref(d_in), rates, C.byref(chosen)))
        print("input slab: %s -> %d" % (" ".join("%.0f" % r for r in rates), chosen.value))
        ok(lib0, lib0.bt709hip_malloc_streaming(h0, out_stride * RING + args.out_shift, args.tries, C.byref(d_out), rates, C.byref(chosen)))
        print("output slab: %s -> %d" % (" ".join("%.0f" % r for r in rates), chosen.value))
    d_out = C.c_void_p(d_out.value + args.out_shift)
    rng = np.random.default_rng(1)
    for i in range(RING):
        buf = rng.integers(0, 256, (1, yb + cb), dtype=np.uint8)
        ok(lib0, lib0.bt709hip_upload(h0, d_in.value + i * in_stride, buf.shape[1], buf.ctypes.data, buf.shape[1], buf.shape[1], 1, None))
        lib0.bt709hip_stream_synchronize(h0, None)  # the upload is asynchronous and `buf` is replaced in the next round
    ok(lib0, lib0.bt709hip_stream_synchronize(h0, None))
    frames, surfs = (Frame * RING)(), (Surface * RING)()
    for i in range(RING):
        b = d_in.value + i * in_stride
        frames[i] = Frame(b, W, b + yb, W, W, H, 1, {0: 1, 1: 2, 2: 3, 3: 1}[args.gamma])
        surfs[i] = Surface(d_out.value + i * out_stride, W * opx, W, H, 1 if opx == 8 else 0, 0)
    per = args.per_launch

    def run(k, n):
        lib, dec = libs[k], decs[k]
        for _ in range(n):
            for first in range(0, RING, per):
                fp = C.cast(C.byref(frames, first * C.sizeof(Frame)), C.POINTER(Frame))
                sp = C.cast(C.byref(surfs, first * C.sizeof(Surface)), C.POINTER(Surface))
                ok(lib, lib.bt709hip_decode_batch(dec, per, fp, None, sp, None, 0))

    evs = []
    for lib, h in zip(libs, ctxs):
        e0, e1 = C.c_void_p(), C.c_void_p()
        lib.bt709hip_event_create(h, C.byref(e0))
        lib.bt709hip_event_create(h, C.byref(e1))
        evs.append((e0, e1))
    # reference checksum from the first library: every variant must write the same bytes
    sums = []
    for k, (lib, h) in enumerate(zip(libs, ctxs)):
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            run(k, 1)
            lib.bt709hip_stream_synchronize(h, None)
        host = np.empty(ob, dtype=np.uint8)
        ok(lib, lib.bt709hip_download(h, host.ctypes.data, W * opx, d_out.value + (RING - 1) * out_stride, W * opx, W * opx, H, None))
        lib.bt709hip_stream_synchronize(h, None)
        sums.append(int(host.view(np.uint32).astype(np.uint64).sum()))
    print("output checksums equal:", len(set(sums)) == 1)
    table = [[] for _ in libs]
    for _ in range(args.rounds):
        for k, (lib, h) in enumerate(zip(libs, ctxs)):
            e0, e1 = evs[k]
            lib.bt709hip_event_record(h, e0, None)
            run(k, args.steps)
            lib.bt709hip_event_record(h, e1, None)
            lib.bt709hip_stream_synchronize(h, None)
            ms = C.c_float()
            lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
            table[k].append(args.steps * RING * W * H / (ms.value / 1e3) / 1e9)
    for p, r in zip(args.libs, table):
        med = sorted(r)[len(r) // 2]
        print("%-52s %s  median %.1f Gpixel/s = %.4f" % (p if p.startswith("shipped") else os.path.basename(p), " ".join("%.1f" % x for x in r), med, med * (1.5 + opx) / 8000))


if __name__ == "__main__":
    main()
