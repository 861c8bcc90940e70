#!/usr/bin/env python3
"""Lab (round 4): does the slow placement regime depend on how far apart the eight XCD streams of the banded launch are?
ONE process, two rings of 256 4K frames -- the first allocation (tries = 1, often a slow slab) and a hunted one -- and on EACH
ring the 256-frame launch under every work map: plain, contiguous bands (shipped: streams 32 frames = 1.06 GB apart), bands
interleaved frame by frame, bands in chunks of 2 / 4 / 8 / 16 frames (streams 66 MB ... 531 MB apart), and the plain map in
32-frame launches.  Same frames (random bytes) on both rings; every map's output checksummed against the shipped map's.
    python tools/band_chunks_lab.py [rounds=2]
NEEDS the chunked map, which lived in the kernel only for this experiment (BT709HIP_OPT_XCD_BANDS values 3..6 of commit 8045956;
the product clamps the option to 0..2 again): check that commit out to re-run it.  Result: profiles/r04_band_chunks.txt."""
import ctypes as C
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import gpu_helpers as gh  # noqa: E402
import metalbt709decoder_amd as mb  # noqa: E402
from metalbt709decoder_amd import _capi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, N = 3840, 2160, 256
ctx = gh.context(); lib, h = ctx.lib, ctx.handle
dec = gh.make_decoder(mb.MetalBT709GammaApple)
rings = {"first": mb.FrameRing(dec, (W, H), N, tries=1)}
rng = np.random.default_rng(0x709)
for i in range(N):
    buf = rng.integers(0, 256, (H * 3 // 2, W), dtype=np.uint8)
    rings["first"].pixelBuffer(i).upload_planes(buf[:H], buf[H:])
rings["hunted"] = mb.FrameRing(dec, (W, H), N, tries=6)
f0, f1 = _capi.Frame(), _capi.Frame()
lib.bt709hip_ring_frame(rings["first"].handle, 0, C.byref(f0), None, None)
lib.bt709hip_ring_frame(rings["hunted"].handle, 0, C.byref(f1), None, None)
_capi.check(lib.bt709hip_copy_probe(h, f1.y, f0.y, (W * H * 3 // 2) * N, None)); ctx._sync(None)
pl = rings["hunted"].placement()
print("hunted ring: first pairing %.0f, chosen %.0f GB/s; prescan %s" % (pl.first_GBps, pl.chosen_GBps, [round(v) for v in pl.out_prescan_GBps[:pl.out_candidates]]))
e0, e1 = C.c_void_p(), C.c_void_p(); lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))
MAPS = [("plain, 256 per launch", 0, 256), ("contiguous bands (shipped)", 1, 256), ("bands frame by frame", 2, 256), ("chunks of 2", 3, 256),
        ("chunks of 4", 4, 256), ("chunks of 8", 5, 256), ("chunks of 16", 6, 256), ("plain, 32 per launch", 0, 32)]


def step(ring, per):
    for i in range(0, N, per):
        assert ring.decode(i, per)


def checksum(ring):
    o = _capi.Surface(); lib.bt709hip_ring_frame(ring.handle, 0, None, None, C.byref(o))
    crc = 0
    for i in (0, 37, 100, 255):
        lib.bt709hip_ring_frame(ring.handle, i, None, None, C.byref(o))
        raw = np.empty((64, W * 4), np.uint8)
        _capi.check(lib.bt709hip_download(h, raw.ctypes.data, W * 4, o.bgra + 1000 * o.stride, o.stride, W * 4, 64, None)); ctx._sync(None)
        crc = zlib.crc32(raw.tobytes(), crc)
    return crc


ref = {}
for rnd in range(rounds):
    for name, ring in rings.items():
        for label, opt, per in MAPS:
            dec.setOption(_capi.OPT_XCD_BANDS, opt)
            t_end = time.perf_counter() + 0.15
            while time.perf_counter() < t_end:
                step(ring, per); ctx._sync(None)
            lib.bt709hip_event_record(h, e0, None)
            for _ in range(40):
                step(ring, per)
            lib.bt709hip_event_record(h, e1, None); ctx._sync(None)
            ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
            frac = 40 * N * W * H * 5.5 / (ms.value / 1e3) / 8e12
            crc = checksum(ring)
            ok = ref.setdefault(name, crc) == crc
            print("round %d  %-7s ring  %-28s frac %.4f  %s" % (rnd, name, label, frac, "same bytes" if ok else "DIFFERENT BYTES"), flush=True)
