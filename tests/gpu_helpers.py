"""Helpers shared by the -m gpu parity tests, bench.py and smoke(): drive the product
through its host mirror (which calls the C ABI) and hand back numpy arrays."""
import numpy as np

import metalbt709decoder_amd as mb

TRANSFER_FOR_GAMMA = {
    mb.MetalBT709GammaApple: mb.kCVImageBufferTransferFunction_ITU_R_709_2,
    mb.MetalBT709GammaSRGB: mb.kCVImageBufferTransferFunction_sRGB,
    mb.MetalBT709GammaLinear: mb.kCVImageBufferTransferFunction_Linear,
    mb.MetalBT709GammaITU709: mb.kCVImageBufferTransferFunction_ITU_R_709_2,
}

_ctx = None


def context():
    """One MetalRenderContext per process (device 0)."""
    global _ctx
    if _ctx is None:
        c = mb.MetalRenderContext(0)
        if not c.setupMetal():
            raise RuntimeError("no HIP device: the product has no CPU fallback")
        _ctx = c
    return _ctx


def make_decoder(gamma=mb.MetalBT709GammaApple, has_alpha=False, alpha_fill=0xFF, options=None):
    """options: {_capi.OPT_*: value} kernel-selection knobs (bt709hip_decoder_set_option)."""
    d = mb.MetalBT709Decoder()
    d.metalRenderContext = context()
    d.gamma = gamma
    d.hasAlphaChannel = has_alpha
    d.alphaFill = alpha_fill
    for opt, val in (options or {}).items():
        d.setOption(opt, val)
    assert d.setupMetal(), d.lastStatus
    return d


def make_buffer(y, cbcr, gamma, y_stride=None, cbcr_stride=None, tag=True):
    """Upload tight numpy planes into a (possibly padded) 420v buffer."""
    h, w = y.shape
    buf = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(context(), (w, h), y_stride, cbcr_stride)
    if tag:
        buf.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2)
        buf.setAttachment("TransferFunction", TRANSFER_FOR_GAMMA[gamma])
    if w and h:
        buf.upload_planes(y, cbcr)
    return buf


def make_alpha_buffer(a):
    h, w = a.shape
    buf = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(context(), (w, h))
    buf.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2)
    buf.setAttachment("TransferFunction", mb.kCVImageBufferTransferFunction_Linear)
    buf.upload_planes(a, np.full((h // 2, w), 128, np.uint8))
    return buf


def gpu_decode(y, cbcr, gamma=mb.MetalBT709GammaApple, alpha=None, alpha_fill=0xFF, y_stride=None,
               cbcr_stride=None, out_stride=None, decoder=None):
    """Decode one frame on the GPU; returns (H, W*4) uint8 BGRA, or None if decodeBT709
    returned False."""
    ctx = context()
    h, w = y.shape
    dec = decoder or make_decoder(gamma, has_alpha=alpha is not None, alpha_fill=alpha_fill)
    buf = make_buffer(y, cbcr, dec.gamma, y_stride, cbcr_stride)
    abuf = make_alpha_buffer(alpha) if alpha is not None else None
    tex = ctx.makeBGRATexture((w, h), stride=out_stride)
    cb = ctx.commandQueue.commandBuffer()
    ok = dec.decodeBT709(buf, abuf, tex, cb, None, w, h, True)
    if not ok:
        return None
    return ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(h, w * 4)


def gpu_decode_half(y, cbcr, gamma=mb.MetalBT709GammaApple, decoder=None, alpha=None):
    return gpu_decode_scaled(y, cbcr, (y.shape[1] // 2, y.shape[0] // 2), gamma, decoder, alpha)


def gpu_decode_scaled(y, cbcr, out_size, gamma=mb.MetalBT709GammaApple, decoder=None, alpha=None):
    """Fused decode + rescale to out_size = (OW, OH) (the 2:1 kernels when that is exactly half)."""
    ctx = context()
    ow, oh = out_size
    dec = decoder or make_decoder(gamma, has_alpha=alpha is not None)
    buf = make_buffer(y, cbcr, dec.gamma)
    abuf = make_alpha_buffer(alpha) if alpha is not None else None
    tex = ctx.makeBGRATexture((ow, oh))
    if not dec.decodeBT709Scaled(buf, tex, ctx.commandQueue.commandBuffer(), True, alphaPixelBuffer=abuf):
        return None
    return ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(oh, ow * 4)


def random_nv12(w, h, seed, legal=False):
    """Seeded synthetic frame.  legal=False: every byte uniform in [0,255] (exercises
    saturation; the non-DEBUG reference accepts it).  legal=True: Y in [16,235], C in [16,240]."""
    rng = np.random.default_rng(seed)
    if legal:
        y = rng.integers(16, 236, (h, w), dtype=np.uint8)
        c = rng.integers(16, 241, (h // 2, w), dtype=np.uint8)
    else:
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        c = rng.integers(0, 256, (h // 2, w), dtype=np.uint8)
    return y, c


def exhaustive_frame():
    """One 4096x4096 frame containing every (Y,Cb,Cr) triple exactly once.
    Block (by,bx) of the 2048x2048 chroma grid carries Cb = bx&255, Cr = by&255 and the
    four luma values 4*g..4*g+3 with g = (by>>8)*8 + (bx>>8)."""
    by, bx = np.meshgrid(np.arange(2048, dtype=np.uint32), np.arange(2048, dtype=np.uint32), indexing="ij")
    cb, cr = (bx & 255).astype(np.uint8), (by & 255).astype(np.uint8)
    g = ((by >> 8) * 8 + (bx >> 8)).astype(np.uint32)
    cbcr = np.empty((2048, 4096), np.uint8)
    cbcr[:, 0::2], cbcr[:, 1::2] = cb, cr
    y = np.empty((4096, 4096), np.uint8)
    y[0::2, 0::2] = 4 * g
    y[0::2, 1::2] = 4 * g + 1
    y[1::2, 0::2] = 4 * g + 2
    y[1::2, 1::2] = 4 * g + 3
    return y, cbcr


def exhaustive_to_table(bgra, y, cbcr):
    """Reorder the decode of exhaustive_frame() into the (Y<<16)+(Cb<<8)+Cr table layout
    (3 bytes R,G,B per entry) the golden hashes are defined on."""
    h, w = y.shape
    px = bgra.reshape(h, w, 4)
    cb = np.repeat(np.repeat(cbcr[:, 0::2], 2, axis=0), 2, axis=1).astype(np.uint32)
    cr = np.repeat(np.repeat(cbcr[:, 1::2], 2, axis=0), 2, axis=1).astype(np.uint32)
    idx = ((y.astype(np.uint32) << 16) | (cb << 8) | cr).reshape(-1)
    table = np.zeros((1 << 24, 3), np.uint8)
    flat = px.reshape(-1, 4)
    table[idx, 0] = flat[:, 2]
    table[idx, 1] = flat[:, 1]
    table[idx, 2] = flat[:, 0]
    assert np.unique(idx).size == 1 << 24
    return table.reshape(-1)
