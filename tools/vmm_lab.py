#!/usr/bin/env python3
"""Lab: a ring whose slabs are COMPOSED from separately created physical chunks (HIP virtual memory management: one reserved
address range, hipMemCreate chunks mapped back to back) against plain hipMalloc slabs, same 256-frame launch, one process.
Question: small allocations all stream at the fast level (tools/placement_chunks.py) -- does a slab built from them?

    python tools/vmm_lab.py [--chunk-mib 512] [--reps 3]"""
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


class MemLocation(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class MemAllocationProp(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", MemLocation), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", AllocFlags)]


class MemAccessDesc(C.Structure):
    _fields_ = [("location", MemLocation), ("flags", C.c_int)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk-mib", type=int, nargs="+", default=[512])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ring", type=int, default=256)
    args = ap.parse_args()
    W, H, RING = 3840, 2160, args.ring
    ctx = mb.MetalRenderContext(0)
    assert ctx.setupMetal()
    lib, h = ctx.lib, ctx.handle
    dec = mb.MetalBT709Decoder()
    dec.metalRenderContext = ctx
    assert dec.setupMetal()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(MemAllocationProp), C.c_int]
    hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemAddressFree.argtypes = [C.c_void_p, C.c_size_t]
    hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(MemAllocationProp), C.c_ulonglong]
    hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
    hip.hipMemRelease.argtypes = [C.c_void_p]
    hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(MemAccessDesc), C.c_size_t]
    prop = MemAllocationProp()
    prop.type = 1  # hipMemAllocationTypePinned
    prop.location.type = 1  # hipMemLocationTypeDevice
    prop.location.id = 0
    gran = C.c_size_t()
    rc = hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1)
    print("hipMemGetAllocationGranularity rc %d: %d bytes" % (rc, gran.value), flush=True)
    if rc != 0:
        return
    acc = MemAccessDesc()
    acc.location.type, acc.location.id, acc.flags = 1, 0, 3

    class Composed:
        def __init__(self, nbytes, chunk):
            self.chunk = chunk
            self.size = (nbytes + chunk - 1) // chunk * chunk
            self.ptr = C.c_void_p()
            assert chunk <= (1 << 30), "2 GiB chunks reserved without an alignment faulted in the first launch (round 3): not supported here"
            assert hip.hipMemAddressReserve(C.byref(self.ptr), self.size, chunk, None, 0) == 0
            self.handles = []
            for off in range(0, self.size, chunk):
                hd = C.c_void_p()
                rc = hip.hipMemCreate(C.byref(hd), chunk, C.byref(prop), 0)
                assert rc == 0, "hipMemCreate %d" % rc
                assert hip.hipMemMap(C.c_void_p(self.ptr.value + off), chunk, 0, hd, 0) == 0
                self.handles.append(hd)
            assert hip.hipMemSetAccess(self.ptr, self.size, C.byref(acc), 1) == 0

        def free(self):
            hip.hipMemUnmap(self.ptr, self.size)
            for hd in self.handles:
                hip.hipMemRelease(hd)
            hip.hipMemAddressFree(self.ptr, self.size)

    class Plain:
        def __init__(self, nbytes):
            self.ptr = C.c_void_p()
            _capi.check(lib.bt709hip_malloc(h, nbytes, C.byref(self.ptr)))

        def free(self):
            lib.bt709hip_free(h, self.ptr)

    yb, cb, ob = W * H, W * H // 2, W * H * 4
    in_stride = (yb + cb + 255) // 256 * 256
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.bt709hip_event_create(h, C.byref(e0))
    lib.bt709hip_event_create(h, C.byref(e1))
    rng = np.random.default_rng(1)
    buf = rng.integers(0, 256, (1, yb + cb), dtype=np.uint8)

    def rate(d_in, d_out):
        frames, surfs = (Frame * RING)(), (Surface * RING)()
        for i in range(RING):
            b = d_in.value + i * in_stride
            frames[i] = Frame(b, W, b + yb, W, W, H, 1, 1)
            surfs[i] = Surface(d_out.value + i * ob, W * 4, W, H)
        # (no upload: hipMemcpy2DAsync refuses mapped ranges, and any bytes decode)
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
            _capi.check(lib.bt709hip_stream_synchronize(h, None))
        lib.bt709hip_event_record(h, e0, None)
        for _ in range(10):
            _capi.check(lib.bt709hip_decode_batch(dec._handle, RING, frames, None, surfs, None, 0))
        lib.bt709hip_event_record(h, e1, None)
        _capi.check(lib.bt709hip_stream_synchronize(h, None))
        ms = C.c_float()
        lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
        return 10 * RING * W * H * 5.5 / (ms.value / 1e3) / 8e12

    for rep in range(args.reps):
        a, b = Plain(in_stride * RING), Plain(ob * RING)
        print("plain hipMalloc slabs                 %.4f" % rate(a.ptr, b.ptr), flush=True)
        a.free()
        b.free()
        for mib in args.chunk_mib:
            chunk = max(gran.value, mib << 20)
            a, b = Composed(in_stride * RING, chunk), Composed(ob * RING, chunk)
            print("composed from %5d MiB chunks (%3d + %3d) %.4f" % (mib, len(a.handles), len(b.handles), rate(a.ptr, b.ptr)), flush=True)
            a.free()
            b.free()


if __name__ == "__main__":
    main()
