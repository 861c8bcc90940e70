"""Fused decode + rescale to any view size (reference pass 2 =
Renderer/MetalScaleRenderContext.m:55-105 + AAPLShaders.metal:73-85).  The reference leaves the
filtering arithmetic to the sampler hardware and has no test of it: PARITY UNPINNED, the oracle
holds our definition and the GPU is checked against it bit for bit."""
import numpy as np
import pytest

import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi


def _frame(w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h // 2, w), dtype=np.uint8)


# ------------------------------------------------------------------ CPU (definition)

def test_scaled_equals_half_at_exact_2_to_1(oracle):
    """Every weight is exactly 0.25 at a 2:1 ratio, so the general definition reproduces the
    2:1 kernel's (((a+b)+c)+d)*0.25f bit for bit."""
    for gamma in range(4):
        y, c = _frame(48, 24, 3 + gamma)
        assert np.array_equal(oracle.decode_nv12_scaled(gamma, y, c, 24, 12), oracle.decode_nv12_half(gamma, y, c))


def test_scaled_identity_is_plain_decode(oracle):
    """1:1: every sample falls on a texel centre (fx = fy = 0): linearise, re-encode = the byte itself."""
    y, c = _frame(32, 16, 9)
    for gamma in range(4):
        assert np.array_equal(oracle.decode_nv12_scaled(gamma, y, c, 32, 16), oracle.decode_nv12(gamma, y, c))


def test_scaled_flat_frame_any_ratio(oracle):
    y = np.full((20, 36), 150, np.uint8)
    c = np.full((10, 36), 128, np.uint8)
    px = oracle.decode_nv12(0, y, c).reshape(-1, 4)[0]
    for ow, oh in [(7, 5), (36, 20), (50, 33), (1, 1), (100, 3)]:
        out = oracle.decode_nv12_scaled(0, y, c, ow, oh).reshape(-1, 4)
        # weights sum to 1 only up to float rounding; a flat field may move by at most one code
        assert (np.abs(out.astype(int) - px.astype(int)) <= 1).all()


# ------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    gpu_helpers.context()
    return gpu_helpers


def gpu_scaled(gh, y, c, ow, oh, gamma, alpha=None):
    got = gh.gpu_decode_scaled(y, c, (ow, oh), gamma, alpha=alpha)
    assert got is not None
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("gamma", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [((64, 32), (40, 20)), ((64, 32), (17, 9)), ((30, 18), (64, 40)), ((1920, 64), (1280, 43)),
                                   ((50, 22), (1, 1)), ((48, 24), (24, 12)), ((32, 16), (32, 16))])
def test_gpu_scaled_matches_oracle(gh, oracle, gamma, shape):
    (w, h), (ow, oh) = shape
    y, c = _frame(w, h, w + h + ow + gamma)
    got = gpu_scaled(gh, y, c, ow, oh, gamma)
    assert np.array_equal(got, oracle.decode_nv12_scaled(gamma, y, c, ow, oh))


@pytest.mark.gpu
@pytest.mark.parametrize("strides", [(67, 65), (64, 66), (80, 64)])
def test_gpu_scaled_odd_strides_and_many_rows(gh, oracle, strides):
    """Odd plane strides take the byte-gather form of the kernel (an even CbCr plane: one 2-byte load
    per tap); a tall output makes a workgroup walk several rows."""
    ctx = gh.context()
    y, c = _frame(64, 600, 5)
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    buf = gh.make_buffer(y, c, dec.gamma, y_stride=strides[0], cbcr_stride=strides[1])
    tex = ctx.makeBGRATexture((300, 4000))
    assert dec.decodeBT709Scaled(buf, tex, ctx.commandQueue.commandBuffer(), True), dec.lastStatus
    got = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(4000, 300 * 4)
    assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, 300, 4000))


@pytest.mark.gpu
def test_gpu_view_fit_like_the_renderer(gh, oracle):
    """AAPLRenderer's case: a 1920x1080 frame into a view of another aspect and size
    (AAPLRenderer.m:891-977 takes the 2-pass route whenever the sizes differ)."""
    y, c = _frame(1920, 1080, 77)
    got = gpu_scaled(gh, y, c, 1366, 768, mb.MetalBT709GammaApple)
    assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, 1366, 768))


@pytest.mark.gpu
def test_gpu_scaled_bad_tags_and_missing_alpha(gh):
    ctx = gh.context()
    y, c = _frame(16, 8, 1)
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    srgb_tagged = gh.make_buffer(y, c, mb.MetalBT709GammaSRGB)
    assert not dec.decodeBT709Scaled(srgb_tagged, ctx.makeBGRATexture((5, 3)), None, True)
    assert dec.lastStatus == _capi.ERR_TRANSFER
    da = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)
    assert not da.decodeBT709Scaled(srgb_tagged, ctx.makeBGRATexture((5, 3)), None, True)  # an alpha decoder needs its alpha buffer
    assert da.lastStatus == _capi.ERR_INVALID_ARG
    rgba16 = ctx.makeBGRATexture((5, 3), pixelFormat=mb.MTLPixelFormatRGBA16Float)
    assert not dec.decodeBT709Scaled(gh.make_buffer(y, c, dec.gamma), rgba16, None, True)  # pass 2 writes the 8-bit view
    assert dec.lastStatus == _capi.ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [((64, 32), (40, 20)), ((30, 18), (64, 40)), ((1920, 64), (1280, 43)), ((48, 24), (24, 12)),
                                   ((50, 22), (1, 1))])
def test_gpu_scaled_with_alpha(gh, oracle, shape):
    """Alpha clips through the view-fit path: the alpha channel is filtered as a plain unorm with the
    same taps and weights (AAPLShaders.metal:411-438 -> MetalScaleRenderContext.m:55-105)."""
    (w, h), (ow, oh) = shape
    y, c = _frame(w, h, w + ow)
    a = np.random.default_rng(h + oh).integers(0, 256, (h, w), dtype=np.uint8)
    got = gpu_scaled(gh, y, c, ow, oh, mb.MetalBT709GammaSRGB, alpha=a)
    assert np.array_equal(got, oracle.decode_nv12_scaled(mb.MetalBT709GammaSRGB, y, c, ow, oh, alpha=a))


@pytest.mark.gpu
@pytest.mark.parametrize("count", [2, 5, 40])
def test_gpu_scaled_batch(gh, oracle, count):
    """bt709hip_decode_scaled_batch: `count` same-geometry frames into same-sized views, one launch
    (pointer table up to 32 frames, evenly spaced ring beyond)."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    (w, h), (ow, oh) = (96, 40), (61, 27)
    in_pitch, out_pitch = w * h * 3 // 2, ow * oh * 4
    slab_in, slab_out = DeviceBuffer(ctx, count * in_pitch), DeviceBuffer(ctx, count * out_pitch)
    frames = [_frame(w, h, 40 + i) for i in range(count)]
    bufs, texs = [], []
    for i, (y, c) in enumerate(frames):
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        mb.BGRAToBT709Converter.setBT709Attributes(b)
        b.upload_planes(y, c)
        bufs.append(b)
        texs.append(mb.BGRATexture(ctx, ow, oh, ow * 4, ptr=slab_out.ptr + i * out_pitch))
    assert dec.decodeBT709ScaledBatch(bufs, texs, ctx.commandQueue.commandBuffer(), True), dec.lastStatus
    assert ctx.lib.bt709hip_last_kernel_name() == b"decode_nv12_scaled"
    for (y, c), t in zip(frames, texs):
        got = ctx.getBGRATexturePixels(t).view(np.uint8).reshape(oh, ow * 4)
        assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, ow, oh))
    if count > _capi.MAX_BATCH:  # not evenly spaced any more: the pointer-table limit applies
        bufs[1], bufs[2] = bufs[2], bufs[1]
        assert not dec.decodeBT709ScaledBatch(bufs, texs, None, True) and dec.lastStatus == _capi.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_gpu_half_alpha_in_the_persistent_kernel(gh, oracle):
    """The persistent 2:1 kernel computes the alpha channel WITHOUT tables (decoded alpha byte =
    trunc(255 x + 0.5), byteNorm, sum * 63.75, trunc(. + 0.5)): every alpha byte as a constant 2x2 block, every
    byte against every other in one block, and random planes -- equal to the oracle and to the per-tile kernel
    (which goes through the byteNorm / quantiser tables)."""
    w, h = 1024, 64
    y, c = _frame(w, h, 4242)
    a = np.random.default_rng(77).integers(0, 256, (h, w), dtype=np.uint8)
    codes = np.arange(256, dtype=np.uint8)
    a[0:2, 0:512] = np.repeat(codes, 2)[None, :]            # 256 constant blocks
    a[2, 0:512:2], a[2, 1:512:2] = codes, codes[::-1]         # mixed blocks: (b, 255 - b) over (b, b)
    a[3, 0:512] = np.repeat(codes, 2)
    a[4:6, 512:1024] = np.repeat(np.roll(codes, 1), 2)[None, :] ^ np.tile(np.array([0, 1], np.uint8), 256)[None, :]
    want = oracle.decode_nv12_half(mb.MetalBT709GammaSRGB, y, c, alpha=a)
    got = {}
    for kernel in (1, 0):
        dec = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True, options={_capi.OPT_HALF_KERNEL: kernel, _capi.OPT_HALF_WORKGROUPS: 5})
        got[kernel] = gh.gpu_decode_half(y, c, mb.MetalBT709GammaSRGB, decoder=dec, alpha=a)
        name = gh.context().lib.bt709hip_last_kernel_name()
        assert name == (b"decode_nv12_half_rep<alpha>" if kernel else b"decode_nv12_half<wide,alpha>"), name
        assert np.array_equal(got[kernel], want), kernel
    assert np.array_equal(got[0], got[1])


@pytest.mark.gpu
def test_gpu_scaled_strips_with_a_ragged_last_wave(gh, oracle):
    """The vertical taps of a strip are worked out by its first lanes (lane i: row i) and read with
    v_readlane_b32: an output width that leaves the last wave with FEWER live columns than a strip has rows
    must not change that.  8 frames of 64x600 -> 65x600 / 130x300: several rows per strip, one live column in
    the last wave; pass 2 alone on the same shapes."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    scale = mb.MetalScaleRenderContext()
    assert scale.setupRenderPipelines(ctx)
    for (w, h), (ow, oh), count in (((64, 600), (65, 600), 8), ((64, 600), (129, 1100), 8), ((256, 64), (321, 2500), 1)):
        in_pitch, out_pitch = w * h * 3 // 2, ow * oh * 4
        slab_in, slab_out = DeviceBuffer(ctx, count * in_pitch), DeviceBuffer(ctx, count * out_pitch)
        frames = [_frame(w, h, 900 + i) for i in range(count)]
        bufs, texs = [], []
        for i, (y, c) in enumerate(frames):
            base = slab_in.ptr + i * in_pitch
            b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
            mb.BGRAToBT709Converter.setBT709Attributes(b)
            b.upload_planes(y, c)
            bufs.append(b)
            texs.append(mb.BGRATexture(ctx, ow, oh, ow * 4, ptr=slab_out.ptr + i * out_pitch))
        assert dec.decodeBT709ScaledBatch(bufs, texs, ctx.commandQueue.commandBuffer(), True), dec.lastStatus
        for (y, c), t in zip(frames, texs):
            got = ctx.getBGRATexturePixels(t).view(np.uint8).reshape(oh, ow * 4)
            assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, ow, oh)), (w, h, ow, oh)
        # pass 2 alone from the 8-bit intermediate of frame 0
        y, c = frames[0]
        inter, view = ctx.makeBGRATexture((w, h)), ctx.makeBGRATexture((ow, oh))
        assert dec.decodeBT709(bufs[0], None, inter, None, None, w, h, False)
        assert scale.renderScaled(ctx, view, ow, oh, None, None, inter, True)
        got = ctx.getBGRATexturePixels(view).view(np.uint8).reshape(oh, ow * 4)
        assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, ow, oh)), ("pass 2", w, h, ow, oh)


@pytest.mark.gpu
@pytest.mark.parametrize("rep", ["0", "1"])
def test_fuzzed_rescale_geometry(gh, oracle, rep):
    """Seeded fuzz over the rescale entry points: any 4-multiple source size, any plane pitch and
    byte alignment, any output pitch; exact 2:1 through the per-tile kernel (rep=0) or the persistent
    one with a random workgroup count (rep=1), and an arbitrary output size through the bilinear
    kernel.  Bytes must equal the oracle's and nothing outside the output rows may be written."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    rng = np.random.default_rng(709 + int(rep))
    for case in range(40):
        w = 4 * int(rng.integers(1, 120))
        hgt = 4 * int(rng.integers(1, 12))
        gamma = int(rng.integers(0, 4))
        aligned = bool(rng.integers(0, 2))
        ys = w + (int(rng.integers(0, 5)) * 4 if aligned else int(rng.integers(0, 37)))
        cs = w + (int(rng.integers(0, 5)) * 4 if aligned else int(rng.integers(0, 37)))
        oy, oc = (0, 0) if aligned else (int(rng.integers(0, 16)) for _ in range(2))
        exact = bool(rng.integers(0, 2))
        ow, oh = (w // 2, hgt // 2) if exact else (int(rng.integers(1, 2 * w)), int(rng.integers(1, 3 * hgt)))
        os_ = 4 * ow + (int(rng.integers(0, 3)) * 8 if aligned else 4 * int(rng.integers(0, 9)))
        oo = 0 if aligned else 4 * int(rng.integers(0, 4))
        use_alpha = bool(rng.integers(0, 4) == 0)
        dec = gh.make_decoder(gamma, has_alpha=use_alpha, options={_capi.OPT_HALF_KERNEL: int(rep),
                                                                  _capi.OPT_HALF_WORKGROUPS: int(rng.integers(1, 300))})
        gamma = dec.gamma  # an alpha decoder runs the sRGB mode
        y, c = _frame(w, hgt, 2000 + case)
        a = rng.integers(0, 256, (hgt, w), dtype=np.uint8) if use_alpha else None

        def plane(arr, pitch, off):
            buf = DeviceBuffer(ctx, pitch * arr.shape[0] + off + 64)
            ctx._upload(buf.ptr + off, pitch, np.ascontiguousarray(arr), None)
            return buf, buf.ptr + off

        by, py = plane(y, ys, oy)
        bc, pc = plane(c, cs, oc)
        src = mb.CVPixelBuffer(ctx, w, hgt, ys, cs, planes=(py, pc))
        src.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2)
        src.setAttachment("TransferFunction", gh.TRANSFER_FOR_GAMMA[dec.gamma])
        abuf = None
        if use_alpha:
            ba, pa = plane(a, ys, int(rng.integers(0, 16)) if not aligned else 0)
            abuf = mb.CVPixelBuffer(ctx, w, hgt, ys, cs, planes=(pa, pc))
            abuf.setAttachment("TransferFunction", mb.kCVImageBufferTransferFunction_Linear)
        out_bytes = os_ * oh + oo + 64
        bo = DeviceBuffer(ctx, out_bytes)
        _capi.check(lib.bt709hip_memset(h, bo.ptr, 0x5A, out_bytes, None))
        ctx._sync(None)
        tex = mb.BGRATexture(ctx, ow, oh, os_, ptr=bo.ptr + oo)
        assert dec.decodeBT709Scaled(src, tex, None, True, alphaPixelBuffer=abuf), (case, dec.lastStatus)
        raw = np.empty(out_bytes, np.uint8)
        _capi.check(lib.bt709hip_download(h, raw.ctypes.data, out_bytes, bo.ptr, out_bytes, out_bytes, 1, None))
        ctx._sync(None)
        rows = raw[oo:oo + os_ * oh].reshape(oh, os_)
        want = (oracle.decode_nv12_half(gamma, y, c, alpha=a) if exact
                else oracle.decode_nv12_scaled(gamma, y, c, ow, oh, alpha=a))
        info = (case, w, hgt, ow, oh, gamma, use_alpha, exact, ys, cs, os_, oy, oc, oo, lib.bt709hip_last_kernel_name())
        assert np.array_equal(rows[:, :4 * ow], want), info
        assert (rows[:, 4 * ow:] == 0x5A).all() and (raw[:oo] == 0x5A).all() and (raw[oo + os_ * oh:] == 0x5A).all(), info


@pytest.mark.gpu
@pytest.mark.parametrize("gamma", [0, 2])
@pytest.mark.parametrize("shape", [((640, 64), (1280, 128)), ((100, 40), (333, 77)), ((1900, 32), (2000, 36)), ((1920, 32), (2021, 40)),
                                   ((322, 18), (1287, 31)), ((64, 16), (4096, 16)), ((480, 270), (960, 540))])
def test_gpu_enlarging_wave_decodes_each_source_pixel_once(gh, oracle, gamma, shape):
    """Round 6: when the view is wider than the frame (AAPLRenderer.m:891-977, a 1080p clip in a larger view) the 64 lanes of a
    wave decode 64 consecutive SOURCE columns once and take their two taps from each other's registers (ds_bpermute) instead of
    each lane decoding both of its taps.  Shapes: exact 2x, an odd ratio with a ragged last wave, scale_x = 0.95 (the widest
    span the form accepts: 61 of its 64 columns) and just above it (per-lane taps again), a wave whose taps all sit on a
    handful of columns (64x), the right-edge clamp."""
    (w, h), (ow, oh) = shape
    y, c = _frame(w, h, w + h + ow + gamma)
    got = gpu_scaled(gh, y, c, ow, oh, gamma)
    assert np.array_equal(got, oracle.decode_nv12_scaled(gamma, y, c, ow, oh))


@pytest.mark.gpu
def test_gpu_enlarging_with_alpha_and_odd_luma_stride(gh, oracle):
    """The same form with an alpha plane (a fourth value travels between the lanes) and with a luma plane at an odd pitch and
    address (the form only needs the CbCr plane 2-byte aligned)."""
    ctx = gh.context()
    (w, h), (ow, oh) = (200, 24), (517, 50)
    y, c = _frame(w, h, 61)
    a = np.random.default_rng(62).integers(0, 256, (h, w), dtype=np.uint8)
    got = gpu_scaled(gh, y, c, ow, oh, mb.MetalBT709GammaSRGB, alpha=a)
    assert np.array_equal(got, oracle.decode_nv12_scaled(mb.MetalBT709GammaSRGB, y, c, ow, oh, alpha=a))
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    buf = gh.make_buffer(y, c, dec.gamma, y_stride=203, cbcr_stride=202)
    tex = ctx.makeBGRATexture((ow, oh))
    assert dec.decodeBT709Scaled(buf, tex, ctx.commandQueue.commandBuffer(), True), dec.lastStatus
    got = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(oh, ow * 4)
    assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, ow, oh))
