"""2:1 rescale with and without an alpha plane, batched (runs on the GPU box): python tools/bench_half_alpha.py W H frames alpha(0|1) [half kernel option: -1 auto, 0 per-tile, 1 persistent]"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_helpers as gh
import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
from metalbt709decoder_amd.decoder import DeviceBuffer
W, H = int(sys.argv[1]), int(sys.argv[2]); ring = int(sys.argv[3]); alpha = int(sys.argv[4])
ctx = gh.context(); lib, h = ctx.lib, ctx.handle
dec = gh.make_decoder(mb.MetalBT709GammaSRGB if alpha else mb.MetalBT709GammaApple, has_alpha=bool(alpha),
                      options={_capi.OPT_HALF_KERNEL: int(sys.argv[5]) if len(sys.argv) > 5 else -1})
in_pitch = (W * H * 3 // 2 + 255) // 256 * 256; a_pitch = W * H; out_pitch = (W // 2) * (H // 2) * 4
si, sa, so = DeviceBuffer(ctx, ring * in_pitch), DeviceBuffer(ctx, ring * a_pitch), DeviceBuffer(ctx, ring * out_pitch)
frames, alphas, surfs = (_capi.Frame * ring)(), (_capi.Frame * ring)(), (_capi.Surface * ring)()
tf = 3 if alpha else 1  # transfer tags: srgb for alpha decoders
for i in range(ring):
    y, c = gh.random_nv12(W, H, seed=i)
    b = si.ptr + i * in_pitch
    ctx._upload(b, W, y, None); ctx._upload(b + W * H, W, c, None)
    a = np.random.default_rng(i).integers(0, 256, (H, W), dtype=np.uint8)
    ctx._upload(sa.ptr + i * a_pitch, W, a, None); ctx._sync(None)
    buf = mb.CVPixelBuffer(ctx, W, H, W, W, planes=(b, b + W * H))
    buf.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2); buf.setAttachment("TransferFunction", gh.TRANSFER_FOR_GAMMA[dec.gamma])
    frames[i] = buf.frame()
    ab = mb.CVPixelBuffer(ctx, W, H, W, W, planes=(sa.ptr + i * a_pitch, b + W * H))
    ab.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2); ab.setAttachment("TransferFunction", mb.kCVImageBufferTransferFunction_Linear)
    alphas[i] = ab.frame()
    surfs[i] = _capi.Surface(so.ptr + i * out_pitch, (W // 2) * 4, W // 2, H // 2, _capi.FORMAT_BGRA8_SRGB, 0)
def step():
    rc = lib.bt709hip_decode_half_batch(dec._handle, ring, frames, alphas if alpha else None, surfs, None, 0); assert rc == 0, rc
for _ in range(20): step()
ctx._sync(None)
e0, e1 = C.c_void_p(), C.c_void_p(); lib.bt709hip_event_create(h, C.byref(e0)); lib.bt709hip_event_create(h, C.byref(e1))
lib.bt709hip_event_record(h, e0, None)
n = 30
for _ in range(n): step()
lib.bt709hip_event_record(h, e1, None); ctx._sync(None)
ms = C.c_float(); lib.bt709hip_event_elapsed_ms(h, e0, e1, C.byref(ms))
us = ms.value * 1e3 / (n * ring)
nbytes = W * H * 3 // 2 + (W * H if alpha else 0) + out_pitch
print("%dx%d -> half, alpha=%d, %d per launch: %.2f us/frame, %.1f Gpx/s out, %.3f of 8 TB/s, kernel %s" % (W, H, alpha, ring, us, (W // 2) * (H // 2) / us / 1e3, nbytes / us / 1e3 / 8000, lib.bt709hip_last_kernel_name().decode()))
