"""4K 1:1 decode in the sRGB mode with and without an alpha plane and in the other gamma modes, over a ring of RING frames (default 256: the
alpha planes of a 32-frame ring are 265 MB, about the size of the Infinity Cache -- round 2's and the first round-3 figure for the
alpha decoder, 0.86, was measured on such a ring with cached alpha loads and was mostly cache hits), PER_LAUNCH frames per launch
(environment, default = the ring; from 64 on the XCD-aware work map applies), the ring made by bt709hip_ring_create with TRIES
candidates per slab (runs on the GPU box):
    [PER_LAUNCH=32] [ONLY_ALPHA=1] [ONLY_GAMMA=2] [BANDS=0] python tools/bench_alpha11.py [library|-] [ring=256] [tries=1]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_helpers as gh
import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
from metalbt709decoder_amd.decoder import DeviceBuffer
W, H = 3840, 2160
ring = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tries = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 1 and sys.argv[1] != "-": _capi.load(os.path.abspath(sys.argv[1]))
ctx = gh.context(); lib, h = ctx.lib, ctx.handle
for alpha, gamma in ((1, mb.MetalBT709GammaSRGB), (0, mb.MetalBT709GammaSRGB), (0, mb.MetalBT709GammaApple), (0, mb.MetalBT709GammaLinear), (0, mb.MetalBT709GammaITU709)):
    if os.environ.get("ONLY_ALPHA") and not alpha: continue
    if os.environ.get("ONLY_GAMMA") and (alpha or gamma != int(os.environ["ONLY_GAMMA"])): continue
    dec = gh.make_decoder(gamma, has_alpha=bool(alpha), options={_capi.OPT_XCD_BANDS: int(os.environ.get("BANDS", "1"))})
    # round 4: the ring is the product's (bt709hip_ring_create: the decoder's own launch as the placement probe, `tries` candidates per
    # slab; an alpha decoder's ring carries the alpha plane as the third plane of the input slab)
    fr = mb.FrameRing(dec, (W, H), ring, tries=tries)
    y, c = gh.random_nv12(W, H, seed=1); a = np.random.default_rng(2).integers(0, 256, (H, W), dtype=np.uint8)
    for i in range(ring):
        fr.pixelBuffer(i).upload_planes(y, c)
        if alpha:
            ab = fr.alphaPixelBuffer(i)
            ctx._upload(ab.y_ptr, ab.y_stride, a, None); ctx._sync(None)
    per = int(os.environ.get("PER_LAUNCH", ring))
    out_pitch = W * H * 4
    def step():
        for i in range(0, ring, per):
            assert fr.decode(i, per), dec.lastStatus
    import time
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        step(); ctx._sync(None)
    ctx._sync(None)
    e0, e1 = C.c_void_p(), C.c_void_p(); lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))
    lib.bt709hip_event_record(h, e0, None)
    n = max(3, 20 * 32 // ring)
    for _ in range(n): step()
    lib.bt709hip_event_record(h, e1, None); ctx._sync(None)
    ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
    us = ms.value * 1e3 / (n * ring)
    nbytes = W * H * 3 // 2 + (W * H if alpha else 0) + out_pitch
    pl = fr.placement()
    print("4K 1:1 ring %d x %d per launch gamma=%d alpha=%d: %.2f us/frame %.1f Gpx/s %.3f of 8TB/s %s  (ring placement: first pairing %.0f, chosen %.0f GB/s of %d x %d candidates)"
          % (ring, per, gamma, alpha, us, W * H / us / 1e3, nbytes / us / 1e3 / 8000, lib.bt709hip_last_kernel_name().decode(), pl.first_GBps, pl.chosen_GBps, pl.in_candidates, pl.out_candidates))
    fr.release()
