"""GPU parity tests (-m gpu): the HIP path, called through the C ABI via the host
mirror, against the CPU oracle and the committed golden fixtures.  Bar: bit-exact
B,G,R,A bytes -- this is integer/byte output, there is no tolerance anywhere below.

Reads like EmptyiOSTests/MetalBT709DecoderTests.m:189-277: make a 420v buffer, tag it
BT.709, copy packed pixels in, -decodeBT709:... waitUntilCompleted, read the texture.
"""
import hashlib

import ctypes as C

import numpy as np
import pytest

import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
from oracle_lib import GAMMA_NAMES

pytestmark = pytest.mark.gpu

GAMMAS = [mb.MetalBT709GammaApple, mb.MetalBT709GammaSRGB, mb.MetalBT709GammaLinear, mb.MetalBT709GammaITU709]


@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    gpu_helpers.context()
    return gpu_helpers


def test_native_library_is_loaded(gh):
    """The HIP extension is in-tree and actually the thing that runs."""
    import os
    lib = mb.load_library()
    assert os.path.samefile(lib._name, os.path.join(os.path.dirname(mb.__file__), "libbt709hip.so"))
    info = gh.context().info()
    assert info.arch.decode().startswith("gfx950"), info.arch
    assert info.wavefront_size == 64


# ------------------------------------------------------------------ reference vectors

def test_metal_decode_vectors(gh, vectors):
    """The reference's 28 Metal decode expectations, through the reference's own test
    flow (MetalBT709DecoderTests.m:189-277), default Apple gamma."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    for r in vectors["metal_decode"]:
        Y, Cb, Cr = r["ycbcr"]
        outBT709 = np.full(4, (Cr << 16) | (Cb << 8) | Y, dtype=np.uint32)
        bgraSRGBTexture = ctx.makeBGRATexture((2, 2))
        yCbCrBuffer = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (2, 2))
        mb.BGRAToBT709Converter.setBT709Attributes(yCbCrBuffer)
        mb.BGRAToBT709Converter.copyBT709ToCoreVideo(outBT709, yCbCrBuffer)
        commandBuffer = ctx.commandQueue.commandBuffer()
        worked = dec.decodeBT709(yCbCrBuffer, None, bgraSRGBTexture, commandBuffer, None, 2, 2, True)
        assert worked, r["test"]
        px = ctx.getBGRATexturePixels(bgraSRGBTexture)
        for word in px.reshape(-1):
            got = [(int(word) >> 16) & 0xFF, (int(word) >> 8) & 0xFF, int(word) & 0xFF]
            assert got == r["rgb_out"], (r["test"], r["src"], got)
            assert int(word) >> 24 == 0xFF


def test_converter_software_vectors(gh, vectors):
    """unconvertSoftware semantics: alpha_fill 0 reproduces its output words exactly."""
    dec = gh.make_decoder(mb.MetalBT709GammaApple, alpha_fill=0x00)
    for r in vectors["converter"]:
        if r["decode_type"] not in ("Software", "Metal"):
            continue
        Y, Cb, Cr = r["ycbcr"]
        y = np.full((2, 2), Y, np.uint8)
        c = np.array([[Cb, Cr]], np.uint8)
        out = gh.gpu_decode(y, c, decoder=dec).view(np.uint32)
        R, G, B = r["rgb_out"]
        assert (out == ((R << 16) | (G << 8) | B)).all(), r["test"]


# ------------------------------------------------------------------ exhaustive sweep

@pytest.mark.parametrize("gamma", GAMMAS)
def test_exhaustive_2_24_sweep(gh, oracle, refdata, gamma):
    """Every (Y,Cb,Cr) triple once: the GPU output, reordered into table layout, must
    hash to the value the REFERENCE HEADERS produced (tests/golden/reference.json) and
    equal the oracle's table byte for byte."""
    y, c = gh.exhaustive_frame()
    out = gh.gpu_decode(y, c, gamma)
    assert out is not None
    table = gh.exhaustive_to_table(out, y, c)
    assert hashlib.sha256(table.tobytes()).hexdigest() == refdata["table_sha256"][GAMMA_NAMES[gamma]]
    assert np.array_equal(table, oracle.decode_table(gamma))
    assert (out.reshape(-1, 4)[:, 3] == 0xFF).all()


def test_exhaustive_2_24_sweep_through_an_alpha_decoder(gh, oracle, refdata):
    """An alpha decoder runs the sRGB mode in ARITHMETIC (no table: quantise_byte for R, G, B and alpha): every
    (Y,Cb,Cr) triple must still hash to the reference headers' sRGB-mode table, and every alpha byte to the oracle's."""
    y, c = gh.exhaustive_frame()
    a = np.random.default_rng(24).integers(0, 256, y.shape, dtype=np.uint8)
    a[0, :256] = np.arange(256, dtype=np.uint8)
    out = gh.gpu_decode(y, c, mb.MetalBT709GammaSRGB, alpha=a)
    assert out is not None and gh.context().lib.bt709hip_last_kernel_name() == b"decode_nv12_quads<alpha>"
    table = gh.exhaustive_to_table(out, y, c)
    assert hashlib.sha256(table.tobytes()).hexdigest() == refdata["table_sha256"][GAMMA_NAMES[mb.MetalBT709GammaSRGB]]
    alpha_map = np.array([oracle.lib.bt709o_decode_alpha(int(v)) for v in range(256)], np.uint8)
    assert np.array_equal(out.reshape(y.shape[0], -1, 4)[:, :, 3], alpha_map[a])


# ------------------------------------------------------------------ frames vs oracle

@pytest.mark.parametrize("gamma", GAMMAS)
@pytest.mark.parametrize("size", [(2, 2), (4, 2), (2, 4), (6, 6), (18, 10), (64, 64), (250, 30), (1028, 6),
                                  (1920, 1080)])
def test_random_frames(gh, oracle, gamma, size):
    w, h = size
    y, c = gh.random_nv12(w, h, seed=w * 131 + h + gamma)
    out = gh.gpu_decode(y, c, gamma)
    assert np.array_equal(out, oracle.decode_nv12(gamma, y, c))


@pytest.mark.parametrize("strides", [(1936, 1936, None), (1923, 1925, None), (1920, 2048, 7684), (1921, 1920, 7680 + 64)])
def test_ragged_strides_and_fallback_kernel(gh, oracle, strides):
    """CoreVideo planes have bytesPerRow >= width (CVPixelBufferUtils.h:82-155).  Odd
    strides force the general kernel; the result must not change."""
    ys, cs, os_ = strides
    y, c = gh.random_nv12(1920, 54, seed=5)
    out = gh.gpu_decode(y, c, mb.MetalBT709GammaApple, y_stride=ys, cbcr_stride=cs, out_stride=os_)
    assert np.array_equal(out, oracle.decode_nv12(mb.MetalBT709GammaApple, y, c))
    name = mb.load_library().bt709hip_last_kernel_name().decode()
    if ys % 4 or cs % 4 or (os_ or 0) % 16:
        assert "blocks" in name
    else:
        assert "quads" in name


def test_fuzzed_geometry(gh, oracle):
    """Seeded fuzz over what CoreVideo may hand the decoder: any even size, any plane pitch
    >= width, any byte alignment of the plane bases, 4-byte aligned output with any pitch that is
    a multiple of 4; every gamma; with and without an alpha plane.  Whatever kernel the shim
    picks, the bytes must equal the oracle's and nothing outside the rows may be written."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    rng = np.random.default_rng(20261002)
    decs = {}
    for case in range(60):
        w = 2 * int(rng.integers(1, 200))
        hgt = 2 * int(rng.integers(1, 24))
        gamma = int(rng.integers(0, 4))
        use_alpha = bool(rng.integers(0, 4) == 0)
        aligned = bool(rng.integers(0, 2))
        ys = w + (int(rng.integers(0, 5)) * 4 if aligned else int(rng.integers(0, 37)))
        cs = w + (int(rng.integers(0, 5)) * 4 if aligned else int(rng.integers(0, 37)))
        os_ = 4 * w + (int(rng.integers(0, 3)) * 16 if aligned else 4 * int(rng.integers(0, 9)))
        oy, oc, oa = (0, 0, 0) if aligned else (int(rng.integers(0, 16)) for _ in range(3))
        oo = 0 if aligned else 4 * int(rng.integers(0, 4))
        key = (gamma, use_alpha)
        if key not in decs:
            decs[key] = gh.make_decoder(gamma, has_alpha=use_alpha)
        dec = decs[key]
        y, c = gh.random_nv12(w, hgt, seed=1000 + case)
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
            ba, pa = plane(a, ys, oa)
            abuf = mb.CVPixelBuffer(ctx, w, hgt, ys, cs, planes=(pa, pc))
            abuf.setAttachment("TransferFunction", mb.kCVImageBufferTransferFunction_Linear)
        out_bytes = os_ * hgt + oo + 64
        bo = DeviceBuffer(ctx, out_bytes)
        _capi.check(lib.bt709hip_memset(h, bo.ptr, 0x5A, out_bytes, None))
        ctx._sync(None)
        tex = mb.BGRATexture(ctx, w, hgt, os_, ptr=bo.ptr + oo)
        assert dec.decodeBT709(src, abuf, tex, None, None, w, hgt, True), (case, dec.lastStatus)
        raw = np.empty(out_bytes, np.uint8)
        _capi.check(lib.bt709hip_download(h, raw.ctypes.data, out_bytes, bo.ptr, out_bytes, out_bytes, 1, None))
        ctx._sync(None)
        rows = raw[oo:oo + os_ * hgt].reshape(hgt, os_)
        want = oracle.decode_nv12(dec.gamma, y, c, alpha=a)
        assert np.array_equal(rows[:, :4 * w], want), (case, w, hgt, gamma, use_alpha, ys, cs, os_, oy, oc, oo)
        assert (rows[:, 4 * w:] == 0x5A).all() and (raw[:oo] == 0x5A).all() and (raw[oo + os_ * hgt:] == 0x5A).all()


def test_video_legal_range(gh, oracle):
    y, c = gh.random_nv12(640, 360, seed=9, legal=True)
    for gamma in GAMMAS:
        assert np.array_equal(gh.gpu_decode(y, c, gamma), oracle.decode_nv12(gamma, y, c))


def test_bundled_pattern_crops(gh, refdata, patterns):
    """Crops of the reference's bundled images: GPU bytes == reference-header bytes."""
    for rec in refdata["patterns"]:
        y, c = patterns[rec["tag"] + "_y"], patterns[rec["tag"] + "_uv"]
        for gamma in GAMMAS:
            out = gh.gpu_decode(y, c, gamma)
            assert hashlib.sha256(out.tobytes()).hexdigest() == rec["bgra_sha256"][GAMMA_NAMES[gamma]], rec["tag"]
        assert np.array_equal(gh.gpu_decode(y, c, mb.MetalBT709GammaApple), patterns[rec["tag"] + "_bgra_apple"])


def test_whole_bundled_patterns(gh, full_patterns):
    """The WHOLE bundled test images (1920x1080 QuickTime pattern in both renditions = BASELINE config 1's frame,
    512x512 Image.tga): GPU bytes hash to what the reference headers produce, every gamma, 1:1 and through the
    fused exact 2:1 rescale."""
    images, planes = full_patterns
    for rec in images:
        y, c = planes[rec["tag"] + "_y"], planes[rec["tag"] + "_uv"]
        for gamma in GAMMAS:
            out = gh.gpu_decode(y, c, gamma)
            assert hashlib.sha256(out.tobytes()).hexdigest() == rec["bgra_sha256"][GAMMA_NAMES[gamma]], rec["tag"]
            half = gh.gpu_decode_half(y, c, gamma)
            assert hashlib.sha256(half.tobytes()).hexdigest() == rec["half_sha256"][GAMMA_NAMES[gamma]], rec["tag"]


def test_4k_full_size(gh, oracle):
    """BASELINE config 3 geometry, full size, against the oracle (threads split rows)."""
    from concurrent.futures import ThreadPoolExecutor
    w, h = 3840, 2160
    y, c = gh.random_nv12(w, h, seed=4)
    out = gh.gpu_decode(y, c, mb.MetalBT709GammaApple)
    want = np.zeros((h, w * 4), np.uint8)
    bands = [(r, min(r + 270, h)) for r in range(0, h, 270)]
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda b: oracle.decode_nv12(0, y, c, rows=b, out=want), bands))
    assert np.array_equal(out, want)


def test_8k_checksum_of_rows_property(gh, oracle):
    """BASELINE config 4 source geometry (7680x4320): a frame built by tiling a small
    tile must decode to the tiling of the tile's decode (size-independent property; the
    tile itself is checked against the oracle)."""
    ty, tc = gh.random_nv12(256, 16, seed=8)
    y = np.tile(ty, (270, 30))
    c = np.tile(tc, (270, 30))
    out = gh.gpu_decode(y, c, mb.MetalBT709GammaApple)
    tile = oracle.decode_nv12(0, ty, tc)
    assert np.array_equal(out, np.tile(tile, (270, 30)))


# ------------------------------------------------------------------ alpha channel

def test_alpha_channel(gh, oracle):
    """hasAlphaChannel: second Y-only buffer decoded as linear alpha, sRGB gamma forced
    (MetalBT709Decoder.m:165-169; AAPLShaders.metal:411-438)."""
    w, h = 72, 20
    y, c = gh.random_nv12(w, h, seed=21)
    a = np.random.default_rng(22).integers(0, 256, (h, w), dtype=np.uint8)
    dec = gh.make_decoder(mb.MetalBT709GammaApple, has_alpha=True)
    assert dec.gamma == mb.MetalBT709GammaSRGB
    out = gh.gpu_decode(y, c, alpha=a, decoder=dec)
    assert np.array_equal(out, oracle.decode_nv12(mb.MetalBT709GammaSRGB, y, c, alpha=a))
    # all 256 alpha codes
    a2 = np.arange(256, dtype=np.uint8).reshape(4, 64)
    y2, c2 = gh.random_nv12(64, 4, seed=23)
    out2 = gh.gpu_decode(y2, c2, alpha=a2, decoder=dec)
    assert np.array_equal(out2, oracle.decode_nv12(mb.MetalBT709GammaSRGB, y2, c2, alpha=a2))


def test_alpha_odd_width_fallback(gh, oracle):
    w, h = 18, 6
    y, c = gh.random_nv12(w, h, seed=24)
    a = np.random.default_rng(25).integers(0, 256, (h, w), dtype=np.uint8)
    dec = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)
    out = gh.gpu_decode(y, c, alpha=a, decoder=dec)
    assert np.array_equal(out, oracle.decode_nv12(mb.MetalBT709GammaSRGB, y, c, alpha=a))


# ------------------------------------------------------------------ batch / streams

def test_batch_equals_single(gh, oracle):
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 8, 320, 48
    frames = [gh.random_nv12(w, h, seed=100 + i) for i in range(n)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
    assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True)
    for (y, c), t in zip(frames, texs):
        got = ctx.getBGRATexturePixels(t).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, y, c))
    too_many = bufs * 5
    assert not dec.decodeBT709Batch(too_many, texs * 5) and dec.lastStatus == _capi.ERR_UNSUPPORTED


def test_frame_sharder_four_lanes_on_one_device(gh, oracle):
    """The in-process multi-GPU dispatcher (bt709hip_shard_*): frame i -> lane i mod n, each lane its own context,
    decoder and in-flight pool.  Here the four lanes all sit on device 0 (ordinals may repeat), 64 frames go
    through, every one is byte-compared with the oracle; then tag / size validation and the ticket window."""
    w, h, lanes, depth = 320, 64, 4, 2
    sh = mb.FrameSharder([0] * lanes, (w, h), gamma=mb.MetalBT709GammaApple, depth=depth)
    assert sh.handle, sh.lastStatus
    assert [sh.laneDevice(i) for i in range(lanes)] == [0] * lanes and sh.lib.bt709hip_shard_lanes(sh.handle) == lanes
    frames = [gh.random_nv12(w, h, seed=4000 + i) for i in range(64)]
    window = lanes * depth
    tickets = []
    for i, (y, c) in enumerate(frames):
        t = sh.submit(y, c)
        assert t == i, sh.lastStatus
        tickets.append(t)
        if i >= window - 1:  # collect the oldest frame still held: its slot is the next one to be reused
            j = i - (window - 1)
            assert np.array_equal(sh.wait(tickets[j]), oracle.decode_nv12(0, *frames[j])), j
    for j in range(64 - (window - 1), 64):
        assert np.array_equal(sh.wait(tickets[j]), oracle.decode_nv12(0, *frames[j])), j
    assert sh.wait(64) is None and sh.lastStatus == _capi.ERR_INVALID_ARG          # never submitted
    assert sh.wait(64 - window - 1) is None and sh.lastStatus == _capi.ERR_INVALID_ARG  # its slot has been reused
    y, c = frames[0]
    assert sh.submit(y[:32], c[:16]) is None and sh.lastStatus == _capi.ERR_SIZE_MISMATCH
    assert sh.submit(y, c, transfer=mb.kCVImageBufferTransferFunction_sRGB) is None and sh.lastStatus == _capi.ERR_TRANSFER
    assert sh.submit(y, c) == 64  # a refused frame takes no ticket and no lane
    assert np.array_equal(sh.wait(64), oracle.decode_nv12(0, y, c))
    # acquire + cancel consumes a slot of the lane without a ticket: frames are found by ticket, and a ticket whose slot
    # was recycled early by that is reported gone, never served from another frame's pixels
    lib, t = sh.lib, C.c_uint64()
    py, pc, ys, cs = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
    assert lib.bt709hip_shard_acquire(sh.handle, C.byref(t), C.byref(py), C.byref(ys), C.byref(pc), C.byref(cs), None, None) == _capi.OK
    assert t.value == 65 and lib.bt709hip_shard_commit(sh.handle, 64) == _capi.ERR_INVALID_ARG  # not the open ticket
    assert lib.bt709hip_shard_acquire(sh.handle, C.byref(t), C.byref(py), C.byref(ys), C.byref(pc), C.byref(cs), None, None) == _capi.ERR_INVALID_ARG
    assert lib.bt709hip_shard_cancel(sh.handle) == _capi.OK and lib.bt709hip_shard_cancel(sh.handle) == _capi.ERR_INVALID_ARG
    more = [gh.random_nv12(w, h, seed=5000 + i) for i in range(2 * window)]
    got_tickets = [sh.submit(yy, cc) for yy, cc in more]
    assert got_tickets == list(range(65, 65 + 2 * window))
    for k in range(window):  # the newest `window - lanes` frames are certainly still there; older ones are gone or exact
        j = 2 * window - 1 - k
        out = sh.wait(got_tickets[j])
        if k < window - lanes:
            assert out is not None
        if out is not None:
            assert np.array_equal(out, oracle.decode_nv12(0, *more[j])), j
    sh.release()
    # alpha decoder (sRGB forced), two lanes
    a = np.random.default_rng(5).integers(0, 256, (h, w), dtype=np.uint8)
    sh = mb.FrameSharder([0, 0], (w, h), hasAlphaChannel=True)
    assert sh.handle and sh.gamma == mb.MetalBT709GammaSRGB
    ts = [sh.submit(y, c, alpha=a) for y, c in frames[:5]]
    for t, (y, c) in zip(ts, frames[:5]):
        assert np.array_equal(sh.wait(t), oracle.decode_nv12(1, y, c, alpha=a))
    sh.release()


@pytest.mark.parametrize("case", [((3840, 8), 64, False), ((7680, 4), 72, False), ((1920, 12), 80, False), ((328, 10), 64, False),
                                  ((640, 6), 64, True), ((640, 6), 64, False, "sRGB"), ((644, 6), 72, False, "Linear"),
                                  ((1280, 4), 64, False, "ITU709"), ((640, 6), 70, False), ((328, 6), 77, True),
                                  # the LINEAR mode's log-bucket kernel (decode_nv12_quads_log) under the map and below it, odd row-pair
                                  # counts, stacked row pairs (1920 wide)
                                  ((644, 10), 72, False, "Linear"), ((1920, 12), 80, False, "Linear"), ((644, 10), 12, False, "Linear"),
                                  ((1920, 12), 9, False, "Linear"), ((3840, 6), 3, False, "Linear")])
def test_xcd_band_work_map(gh, oracle, case):
    """Launches of a multiple of 8 frames, 64 or more, use the XCD-aware work map (a longer launch of any other count: the map over the multiple
    of 8 and the plain map over the rest, two launches) (grid.x = 8 x tiles; each XCD class owns a
    contiguous band of the launch's frames; the frames sit evenly spaced in one slab, as a ring does): every frame distinct, one / two tiles per row, stacked row pairs, an alpha decoder;
    same bytes as the oracle and as the plain map (BT709HIP_OPT_XCD_BANDS = 0), nothing written outside the rows."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    (w, h), n, with_alpha = case[:3]
    ctx = gh.context()
    gamma = mb.MetalBT709GammaSRGB if with_alpha else mb.MetalBT709GammaApple
    if len(case) > 3:  # the other gamma modes (the sRGB mode is the table-free kernel variant)
        gamma = {"sRGB": mb.MetalBT709GammaSRGB, "Linear": mb.MetalBT709GammaLinear, "ITU709": mb.MetalBT709GammaITU709}[case[3]]
    frames = [gh.random_nv12(w, h, seed=7000 + 13 * i + w) for i in range(n)]
    alphas = [np.random.default_rng(i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)] if with_alpha else None
    stride = w * 4 + 16
    in_pitch, a_pitch, out_pitch = (w * h * 3 // 2 + 255) // 256 * 256, (w * h + 255) // 256 * 256, stride * h
    slab_in, slab_a, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * a_pitch), DeviceBuffer(ctx, n * out_pitch)
    bufs, abufs, texs = [], [], []
    for i, (y, c) in enumerate(frames):
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        b.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2)
        b.setAttachment("TransferFunction", gh.TRANSFER_FOR_GAMMA[gamma])
        b.upload_planes(y, c)
        bufs.append(b)
        if with_alpha:
            ab = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(slab_a.ptr + i * a_pitch, base + w * h))
            ab.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2)
            ab.setAttachment("TransferFunction", mb.kCVImageBufferTransferFunction_Linear)
            ctx._upload(slab_a.ptr + i * a_pitch, w, alphas[i], None)
            ctx._sync(None)
            abufs.append(ab)
        texs.append(mb.BGRATexture(ctx, w, h, stride, ptr=slab_out.ptr + i * out_pitch))
    outs = {}
    for banded in (1, 0, 2):  # 2: the bands interleaved frame by frame (lab form of the map, kept as an option)
        dec = gh.make_decoder(gamma, has_alpha=with_alpha, options={_capi.OPT_XCD_BANDS: banded})
        _capi.check(ctx.lib.bt709hip_memset(ctx.handle, slab_out.ptr, 0xC3, n * out_pitch, None))
        assert dec.decodeBT709Batch(bufs, texs, alphaPixelBuffers=abufs or None, waitUntilCompleted=True), dec.lastStatus
        raw = np.empty((n * h, stride), np.uint8)
        _capi.check(ctx.lib.bt709hip_download(ctx.handle, raw.ctypes.data, stride, slab_out.ptr, stride, stride, n * h, None))
        ctx._sync(None)
        assert (raw[:, w * 4:] == 0xC3).all()
        outs[banded] = raw[:, :w * 4].reshape(n, h, w * 4).copy()
    for i, (y, c) in enumerate(frames):
        want = oracle.decode_nv12(gamma, y, c, alpha=alphas[i] if with_alpha else None)
        assert np.array_equal(outs[1][i], want), i
        assert np.array_equal(outs[0][i], want), i
        assert np.array_equal(outs[2][i], want), i


def test_evenly_spaced_batch_beyond_table_limit(gh, oracle):
    """Frames carved at a constant pitch from one allocation (a ring) need no pointer table:
    one launch takes more than BT709HIP_MAX_BATCH of them."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 45, 64, 16
    in_pitch, out_pitch = w * h * 3 // 2, w * h * 4
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    frames = [gh.random_nv12(w, h, seed=300 + i) for i in range(n)]
    bufs, texs = [], []
    for i, (y, c) in enumerate(frames):
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        mb.BGRAToBT709Converter.setBT709Attributes(b)
        b.upload_planes(y, c)
        bufs.append(b)
        texs.append(mb.BGRATexture(ctx, w, h, w * 4, ptr=slab_out.ptr + i * out_pitch))
    assert n > _capi.MAX_BATCH
    assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True)
    for (y, c), t in zip(frames, texs):
        got = ctx.getBGRATexturePixels(t).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, y, c))
    # the same frames in shuffled order are not evenly spaced: the table limit applies again
    order = list(range(n))
    order[3], order[7] = order[7], order[3]
    assert not dec.decodeBT709Batch([bufs[i] for i in order], [texs[i] for i in order])
    assert dec.lastStatus == _capi.ERR_UNSUPPORTED


def test_malloc_streaming_placement_hunt(gh):
    """bt709hip_malloc_streaming: several candidates, the fastest-streaming one kept; the block is usable and freeable."""
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    p, rates, chosen = C.c_void_p(), (C.c_float * 3)(), C.c_int(-1)
    nbytes = 64 << 20
    assert lib.bt709hip_malloc_streaming(h, nbytes, 3, C.byref(p), rates, C.byref(chosen)) == _capi.OK
    assert p.value and 0 <= chosen.value < 3 and all(r > 100.0 for r in rates) and rates[chosen.value] == max(rates)
    _capi.check(lib.bt709hip_memset(h, p, 0x11, nbytes, None))
    back = np.empty((1, 4096), np.uint8)
    _capi.check(lib.bt709hip_download(h, back.ctypes.data, 4096, p.value + nbytes - 4096, 4096, 4096, 1, None))
    ctx._sync(None)
    assert (back == 0x11).all()
    _capi.check(lib.bt709hip_free(h, p))
    assert lib.bt709hip_malloc_streaming(h, 1 << 20, 0, C.byref(p), None, None) == _capi.ERR_INVALID_ARG
    assert lib.bt709hip_malloc_streaming(h, 4096, 1, C.byref(p), None, None) == _capi.OK and p.value  # one try: plain malloc
    _capi.check(lib.bt709hip_free(h, p))


def test_one_stream_per_in_flight_frame(gh, oracle):
    """North-star shape: each in-flight frame on its own HIP stream, no wait until the end."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, h = 640, 360
    frames = [gh.random_nv12(w, h, seed=200 + i) for i in range(6)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs = [ctx.makeBGRATexture((w, h)) for _ in frames]
    cbs = [ctx.commandQueue.commandBuffer(new_stream=True) for _ in frames]
    for b, t, cb in zip(bufs, texs, cbs):
        assert dec.decodeBT709(b, None, t, cb, None, w, h, False)
    for cb in cbs:
        cb.waitUntilCompleted()
    for (y, c), t, cb in zip(frames, texs, cbs):
        got = ctx.getBGRATexturePixels(t).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, y, c))
        cb.release()


def test_one_decoder_shared_by_host_threads(gh, oracle):
    """The decoder keeps no per-frame state (unlike the reference, MetalBT709Decoder.m:449-452),
    so host threads may share it, each with its own stream."""
    from concurrent.futures import ThreadPoolExecutor
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, h = 256, 32
    jobs = []
    for t in range(8):
        y, c = gh.random_nv12(w, h, seed=500 + t)
        jobs.append((y, c, gh.make_buffer(y, c, dec.gamma), ctx.makeBGRATexture((w, h)),
                     ctx.commandQueue.commandBuffer(new_stream=True)))

    def work(job):
        y, c, buf, tex, cb = job
        for _ in range(20):
            assert dec.decodeBT709(buf, None, tex, cb, None, w, h, True)
        return ctx.getBGRATexturePixels(tex, cb).view(np.uint8).reshape(h, w * 4)

    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(work, jobs))
    for (y, c, _, _, cb), got in zip(jobs, outs):
        assert np.array_equal(got, oracle.decode_nv12(0, y, c))
        cb.release()


def test_decode_is_idempotent_and_stateless(gh):
    """The decoder keeps no per-frame state: decoding A, then B, then A again gives A's bytes."""
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    ya, ca = gh.random_nv12(128, 32, seed=31)
    yb, cb = gh.random_nv12(128, 32, seed=32)
    a1 = gh.gpu_decode(ya, ca, decoder=dec)
    gh.gpu_decode(yb, cb, decoder=dec)
    a2 = gh.gpu_decode(ya, ca, decoder=dec)
    assert np.array_equal(a1, a2)


@pytest.mark.parametrize("gamma", [mb.MetalBT709GammaApple, mb.MetalBT709GammaSRGB])
def test_in_flight_frame_pool(gh, oracle, gamma):
    """Host frames through the pool: 3 slots, 8 frames, every frame submitted before the oldest is
    waited for; each result equals the oracle's and slots are recycled in order."""
    dec = gh.make_decoder(gamma)
    w, h, depth = 200, 34, 3
    pool = mb.InFlightFramePool(dec, (w, h), depth)
    frames = [gh.random_nv12(w, h, seed=700 + i) for i in range(8)]
    pending = []
    for i, (y, c) in enumerate(frames):
        if len(pending) == depth:  # the oldest frame's slot is about to be reused: collect it first
            j, slot = pending.pop(0)
            assert np.array_equal(pool.wait(slot), oracle.decode_nv12(gamma, *frames[j])), j
        slot, ybuf, cbuf = pool.acquire()
        assert slot == i % depth
        ybuf[:], cbuf[:] = y, c
        pool.submit(slot)
        pending.append((i, slot))
    for j, slot in pending:
        assert np.array_equal(pool.wait(slot), oracle.decode_nv12(gamma, *frames[j])), j
    # misuse is reported, not executed
    lib = gh.context().lib
    assert lib.bt709hip_pool_submit(pool.handle, 0) == _capi.ERR_INVALID_ARG      # not acquired
    assert lib.bt709hip_pool_submit(pool.handle, 99) == _capi.ERR_INVALID_ARG
    pool.release()


def test_in_flight_frame_pool_with_alpha(gh, oracle):
    """An alpha clip through the pool: every slot also owns an alpha plane (the reference hands
    -decodeBT709: a second CVPixelBuffer, MetalBT709Decoder.h:65-72)."""
    dec = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)
    w, h, depth = 72, 20, 2
    pool = mb.InFlightFramePool(dec, (w, h), depth)
    rng = np.random.default_rng(12)
    frames = [gh.random_nv12(w, h, seed=800 + i) + (rng.integers(0, 256, (h, w), dtype=np.uint8),) for i in range(5)]
    for i, (y, c, a) in enumerate(frames):
        slot, ybuf, cbuf = pool.acquire()
        ybuf[:], cbuf[:] = y, c
        pool.alphaPlane(slot)[:] = a
        pool.submit(slot)
        assert np.array_equal(pool.wait(slot), oracle.decode_nv12(mb.MetalBT709GammaSRGB, y, c, alpha=a)), i
    lib = gh.context().lib
    p, st = C.c_void_p(), C.c_size_t()
    assert lib.bt709hip_pool_alpha_plane(pool.handle, 0, C.byref(p), C.byref(st)) == _capi.ERR_INVALID_ARG  # not acquired
    pool.release()
    opaque = mb.InFlightFramePool(gh.make_decoder(mb.MetalBT709GammaApple), (w, h), 1)
    slot, _, _ = opaque.acquire()
    assert lib.bt709hip_pool_alpha_plane(opaque.handle, slot, C.byref(p), C.byref(st)) == _capi.ERR_UNSUPPORTED
    opaque.release()


def test_recorded_command_buffer_replays(gh, oracle):
    """Six single-frame decodes recorded into one graph (the HIP twin of a command buffer that is
    encoded once): replaying it decodes whatever the input buffers hold at that moment."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, h, n = 128, 36, 6
    frames = [gh.random_nv12(w, h, seed=400 + i) for i in range(n)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
    cb = ctx.commandQueue.commandBuffer(new_stream=True)
    cb.beginRecording()
    for b, t in zip(bufs, texs):
        assert dec.decodeBT709(b, None, t, cb, None, w, h, False), dec.lastStatus
    rec = cb.endRecording()
    for t in texs:  # nothing ran during the recording
        assert not ctx.getBGRATexturePixels(t).any()
    rec.replay(cb)
    cb.waitUntilCompleted()
    for (y, c), t in zip(frames, texs):
        assert np.array_equal(ctx.getBGRATexturePixels(t).view(np.uint8).reshape(h, w * 4), oracle.decode_nv12(0, y, c))
    # new content in the same buffers, same recording
    frames2 = [gh.random_nv12(w, h, seed=500 + i) for i in range(n)]
    for b, (y, c) in zip(bufs, frames2):
        b.upload_planes(y, c)
    rec.replay(cb)
    cb.waitUntilCompleted()
    for (y, c), t in zip(frames2, texs):
        assert np.array_equal(ctx.getBGRATexturePixels(t).view(np.uint8).reshape(h, w * 4), oracle.decode_nv12(0, y, c))
    rec.release()
    cb.release()


# ------------------------------------------------------------------ fused 2:1 rescale

@pytest.mark.parametrize("gamma", GAMMAS)
@pytest.mark.parametrize("size", [(4, 4), (8, 4), (12, 8), (40, 12), (256, 64), (1920, 1080 - 1080 % 4)])
def test_half_scale(gh, oracle, gamma, size):
    """Fused decode + 2:1 downscale against the oracle's two-pass-equivalent restatement
    (parity unpinned by the reference: it has no CPU twin of pass 2)."""
    w, h = size
    y, c = gh.random_nv12(w, h, seed=w + h + gamma)
    out = gh.gpu_decode_half(y, c, gamma)
    assert np.array_equal(out, oracle.decode_nv12_half(gamma, y, c))


@pytest.mark.parametrize("workgroups", [1, 3, 256])
@pytest.mark.parametrize("gamma", GAMMAS)
@pytest.mark.parametrize("size", [(4, 4), (40, 12), (256, 64), (1920, 1080 - 1080 % 4), (4104, 8), (8200, 12)])
def test_half_scale_persistent_kernel(gh, oracle, gamma, size, workgroups):
    """The persistent conflict-free form of the 2:1 kernel (replicated LDS tables, one workgroup per
    CU walking tile rows) forced on frames of every shape: rows narrower than a workgroup, rows
    of 2 and 3 tiles, more / fewer workgroups than tile rows, odd tile-row counts."""
    w, h = size
    y, c = gh.random_nv12(w, h, seed=w + h + gamma)
    dec = gh.make_decoder(gamma, options={_capi.OPT_HALF_KERNEL: 1, _capi.OPT_HALF_WORKGROUPS: workgroups})
    out = gh.gpu_decode_half(y, c, gamma, decoder=dec)
    assert gh.context().lib.bt709hip_last_kernel_name() == b"decode_nv12_half_rep"
    assert np.array_equal(out, oracle.decode_nv12_half(gamma, y, c))


@pytest.mark.parametrize("rep", [0, 1])
@pytest.mark.parametrize("count", [2, 5])
def test_half_scale_batch(gh, oracle, rep, count):
    """Several frames per launch (pointer table; the persistent kernel's cursor crosses frames)."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_HALF_KERNEL: rep, _capi.OPT_HALF_WORKGROUPS: 7})
    w, h = 72, 20
    frames = [gh.random_nv12(w, h, seed=900 + i) for i in range(count)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs = [ctx.makeBGRATexture((w // 2, h // 2)) for _ in range(count)]
    assert dec.decodeBT709ScaledBatch(bufs, texs, ctx.commandQueue.commandBuffer(), True), dec.lastStatus
    for (y, c), tex in zip(frames, texs):
        got = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(h // 2, (w // 2) * 4)
        assert np.array_equal(got, oracle.decode_nv12_half(0, y, c))


def test_half_scale_flat_frame_is_identity(gh):
    y = np.full((64, 64), 180, np.uint8)
    c = np.full((32, 64), 128, np.uint8)
    full = gh.gpu_decode(y, c)
    half = gh.gpu_decode_half(y, c)
    assert (half.reshape(-1, 4) == full.reshape(-1, 4)[0]).all()


@pytest.mark.parametrize("size", [(8, 4), (72, 20), (256, 64), (1028, 8)])
def test_half_scale_with_alpha(gh, oracle, size):
    """Alpha clips go through both passes in the reference (AAPLShaders.metal:411-438 into the
    intermediate, then MetalScaleRenderContext.m:55-105): rgb as for an opaque clip, alpha filtered
    as a plain unorm.  Aligned layouts take the wide kernel, everything else the narrow one."""
    w, h = size
    y, c = gh.random_nv12(w, h, seed=w + h)
    a = np.random.default_rng(w).integers(0, 256, (h, w), dtype=np.uint8)
    out = gh.gpu_decode_half(y, c, alpha=a)
    assert b"alpha" in gh.context().lib.bt709hip_last_kernel_name()
    assert np.array_equal(out, oracle.decode_nv12_half(mb.MetalBT709GammaSRGB, y, c, alpha=a))
    # all 256 alpha codes in flat 2x2 blocks come back unchanged
    a2 = np.repeat(np.repeat(np.arange(256, dtype=np.uint8).reshape(4, 64), 2, axis=0), 2, axis=1)
    y2, c2 = gh.random_nv12(128, 8, seed=3)
    out2 = gh.gpu_decode_half(y2, c2, alpha=a2).reshape(4, 64, 4)
    full = gh.gpu_decode(y2, c2, alpha=a2, decoder=gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)).reshape(8, 128, 4)
    assert np.array_equal(out2[..., 3], full[::2, ::2, 3])


def test_pass2_equals_reference_composed_fixture(gh, pass2, patterns):
    """GPU fused rescale against tests/golden/pass2.json: outputs of the REFERENCE's own inlines
    composed per SURVEY 8(a) row 9 (oracle/ref_harness.c), not of our restatement."""
    import hashlib
    import pass2_cases as pc
    decs = {}

    def dec_for(gamma, alpha):
        if (gamma, alpha) not in decs:
            decs[(gamma, alpha)] = gh.make_decoder(gamma, has_alpha=alpha)
        return decs[(gamma, alpha)]

    for kind, gamma, src, dst, seed, with_alpha in pc.CASES:
        y, c, a = pc.seeded_frame(src, seed, with_alpha)
        got = gh.gpu_decode_scaled(y, c, dst, gamma, dec_for(gamma, with_alpha), alpha=a)
        key = pc.case_key(kind, gamma, src, dst, seed, with_alpha)
        assert got is not None and hashlib.sha256(got.tobytes()).hexdigest() == pass2["cases"][key], key
    for tag, rec in pass2["patterns"].items():
        y, c = patterns[tag + "_y"], patterns[tag + "_uv"]
        for gamma in GAMMAS:
            got = gh.gpu_decode_half(y, c, gamma, dec_for(gamma, False))
            assert hashlib.sha256(got.tobytes()).hexdigest() == rec["half/g%d" % gamma], (tag, gamma)
            for (ow, oh) in pc.PATTERN_SIZES["scaled"]:
                got = gh.gpu_decode_scaled(y, c, (ow, oh), gamma, dec_for(gamma, False))
                assert hashlib.sha256(got.tobytes()).hexdigest() == rec["scaled/g%d/%dx%d" % (gamma, ow, oh)], (tag, gamma)


# ------------------------------------------------------------------ BASELINE configs at full size

def test_config4_8k_to_4k_full_size(gh, oracle):
    """BASELINE config 4 as written: 7680x4320 NV12 -> 3840x2160 through bt709hip_decode_half with
    default options.  The shim must pick the persistent conflict-free kernel at this geometry
    (17 tile rows per CU); the frame is a tiling of a small tile, so its decode must be the tiling
    of the tile's decode (size-independent property), the tile itself checked against the oracle."""
    ty, tc = gh.random_nv12(256, 16, seed=84)
    y, c = np.tile(ty, (270, 30)), np.tile(tc, (270, 30))
    out = gh.gpu_decode_half(y, c, mb.MetalBT709GammaApple)
    assert gh.context().lib.bt709hip_last_kernel_name() == b"decode_nv12_half_rep"
    tile = oracle.decode_nv12_half(0, ty, tc)
    assert out.shape == (2160, 3840 * 4) and np.array_equal(out, np.tile(tile, (270, 30)))
    # and the top rows of a NON-periodic 8K frame straight against the oracle
    y2, c2 = gh.random_nv12(7680, 4320, seed=85)
    out2 = gh.gpu_decode_half(y2, c2, mb.MetalBT709GammaApple)
    for r0 in (0, 2000, 4256):
        assert np.array_equal(out2[r0 // 2:r0 // 2 + 32], oracle.decode_nv12_half(0, y2[r0:r0 + 64], c2[r0 // 2:r0 // 2 + 32]))


def test_config4_batch_of_16_like_the_bench(gh, oracle):
    """bench.py's config-4 launch shape: 16 evenly spaced 8K frames, one decode_half_batch call."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 16, 7680, 4320
    in_pitch, out_pitch = w * h * 3 // 2, (w // 2) * (h // 2) * 4
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    tiles = [gh.random_nv12(512, 8, seed=870 + i) for i in range(n)]
    bufs, texs = [], []
    for i, (ty, tc) in enumerate(tiles):
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        mb.BGRAToBT709Converter.setBT709Attributes(b)
        b.upload_planes(np.tile(ty, (540, 15)), np.tile(tc, (540, 15)))
        bufs.append(b)
        texs.append(mb.BGRATexture(ctx, w // 2, h // 2, (w // 2) * 4, ptr=slab_out.ptr + i * out_pitch))
    assert dec.decodeBT709ScaledBatch(bufs, texs, ctx.commandQueue.commandBuffer(), True), dec.lastStatus
    assert ctx.lib.bt709hip_last_kernel_name() == b"decode_nv12_half_rep"
    for (ty, tc), t in zip(tiles, texs):
        got = ctx.getBGRATexturePixels(t).view(np.uint8).reshape(h // 2, (w // 2) * 4)
        assert np.array_equal(got, np.tile(oracle.decode_nv12_half(0, ty, tc), (540, 15)))


def test_config2_600_frame_1080p_stream(gh, oracle):
    """BASELINE config 2: a 1920x1080 stream of 600 frames (10 s at 60 fps) from a ring of 64
    distinct frames, host memory in, host memory out, through the in-flight frame pool (one HIP
    stream per in-flight frame); every 50th frame is byte-compared with the oracle, all of them by
    a checksum against the first decode of the same ring entry."""
    import zlib
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, h, depth, ring, total = 1920, 1080, 4, 64, 600
    pool = mb.InFlightFramePool(dec, (w, h), depth)
    frames = [gh.random_nv12(w, h, seed=6000 + i) for i in range(ring)]
    first_crc, pending, checked = {}, [], 0

    def collect(i, slot):
        nonlocal checked
        out = pool.wait(slot)
        crc = zlib.crc32(out.tobytes())
        assert first_crc.setdefault(i % ring, crc) == crc, i  # decoding is stateless: same frame, same bytes
        if i % 50 == 0:
            from concurrent.futures import ThreadPoolExecutor
            y, c = frames[i % ring]
            want = np.zeros((h, w * 4), np.uint8)
            with ThreadPoolExecutor(8) as ex:
                list(ex.map(lambda b: oracle.decode_nv12(0, y, c, rows=b, out=want), [(r, r + 108) for r in range(0, h, 108)]))
            assert np.array_equal(out, want), i
            checked += 1

    for i in range(total):
        if len(pending) == depth:
            collect(*pending.pop(0))
        slot, ybuf, cbuf = pool.acquire()
        ybuf[:], cbuf[:] = frames[i % ring]
        pool.submit(slot)
        pending.append((i, slot))
    for item in pending:
        collect(*item)
    assert checked == 12 and len(first_crc) == ring
    pool.release()


def test_config5_per_gpu_unit_8_x_4k(gh, oracle):
    """BASELINE config 5's step on one GPU: 8 x 4K frames, (a) as one decode_batch launch, (b) as 8
    single-frame launches on 8 streams (one HIP stream per in-flight frame); sampled row bands of
    every frame against the oracle, and (a) == (b) byte for byte."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 8, 3840, 2160
    frames = [gh.random_nv12(w, h, seed=5000 + i) for i in range(n)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs_a = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
    texs_b = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
    assert dec.decodeBT709Batch(bufs, texs_a, waitUntilCompleted=True), dec.lastStatus
    cbs = [ctx.commandQueue.commandBuffer(new_stream=True) for _ in range(n)]
    for b, t, cb in zip(bufs, texs_b, cbs):
        assert dec.decodeBT709(b, None, t, cb, None, w, h, False)
    for cb in cbs:
        cb.waitUntilCompleted()
    for i, ((y, c), ta, tb, cb) in enumerate(zip(frames, texs_a, texs_b, cbs)):
        a = ctx.getBGRATexturePixels(ta).view(np.uint8).reshape(h, w * 4)
        b = ctx.getBGRATexturePixels(tb).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(a, b), i
        for r0 in (0, 2 * (137 * (i + 1) % 1000), h - 16):
            assert np.array_equal(a[r0:r0 + 16], oracle.decode_nv12(0, y, c, rows=(r0, r0 + 16))[r0:r0 + 16]), (i, r0)
        cb.release()


# ------------------------------------------------------------------ error behaviour

def _decode_status(gh, dec, buf, tex, rw, rh, alpha=None):
    ok = dec.decodeBT709(buf, alpha, tex, gh.context().commandQueue.commandBuffer(), None, rw, rh, True)
    return ok, dec.lastStatus


@pytest.mark.parametrize("gamma", GAMMAS)
def test_unconvert_packed_444_words(gh, oracle, vectors, gamma):
    """+[BGRAToBT709Converter unconvert:...] (BGRAToBT709Converter.h:34-46, .m:146-198) on its actual input: packed
    4:4:4 words, every pixel its own chroma.  Random words in every gamma, the vectorised and the per-pixel layout, the
    reference's alpha byte 0 (unconvertSoftware's words), odd sizes refused; and -- default gamma -- the 28 (Y,Cb,Cr) ->
    (R,G,B) expectations the reference's XCTest file asserts, as one frame."""
    ctx = gh.context()
    rng = np.random.default_rng(40 + gamma)
    dec = gh.make_decoder(gamma, alpha_fill=0)
    # (3840, 6) / (1028, 4): the vectorised kernel's partial last tile (a lane past the row's end stages the table and stores
    # nothing), several two-row groups; (4100, 2): more than one tile plus a ragged end
    for w, h in ((64, 8), (30, 6), (1024, 2), (3840, 6), (1028, 4), (4100, 2)):
        words = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
        tex = ctx.makeBGRATexture((w, h))
        assert mb.BGRAToBT709Converter.unconvert(dec, words, tex, w, h), dec.lastStatus
        assert (b"<vec>" in ctx.lib.bt709hip_last_kernel_name()) == (w % 4 == 0)
        assert np.array_equal(ctx.getBGRATexturePixels(tex).reshape(-1), oracle.unconvert_packed(gamma, words, w, h))
    tex = ctx.makeBGRATexture((5, 4))
    assert not mb.BGRAToBT709Converter.unconvert(dec, np.zeros((4, 5), np.uint32), tex, 5, 4) and dec.lastStatus == _capi.ERR_ODD_DIMENSIONS
    if gamma == mb.MetalBT709GammaApple:
        recs = vectors["metal_decode"]
        words = np.array([[r["ycbcr"][0] | (r["ycbcr"][1] << 8) | (r["ycbcr"][2] << 16) for r in recs]] * 2, np.uint32)
        tex = ctx.makeBGRATexture((len(recs), 2))
        assert mb.BGRAToBT709Converter.unconvert(dec, words, tex, len(recs), 2)
        got = ctx.getBGRATexturePixels(tex)
        want = np.array([(r["rgb_out"][0] << 16) | (r["rgb_out"][1] << 8) | r["rgb_out"][2] for r in recs], np.uint32)
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def test_unconvert_batch(gh, oracle):
    """bt709hip_unconvert_batch: several frames of one size in ONE launch (grid.z = frame) -- through the pointer table (separately
    allocated textures) and as evenly spaced frames (one slab, more than the table holds) -- every frame compared with the oracle,
    the vectorised and the per-pixel layout; geometry and format errors as the single-frame entry point reports them."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    lib = ctx.lib
    dec = gh.make_decoder(mb.MetalBT709GammaApple, alpha_fill=0)
    rng = np.random.default_rng(77)
    for (w, h), n in (((64, 8), 5), ((30, 6), 3), ((1028, 4), 7)):
        frames = [rng.integers(0, 1 << 24, (h, w), dtype=np.uint32) for _ in range(n)]
        texs = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
        assert mb.BGRAToBT709Converter.unconvertBatch(dec, frames, texs, w, h), dec.lastStatus
        assert (b"<vec>" in lib.bt709hip_last_kernel_name()) == (w % 4 == 0)
        for words, tex in zip(frames, texs):
            assert np.array_equal(ctx.getBGRATexturePixels(tex).reshape(-1), oracle.unconvert_packed(0, words, w, h))
    # evenly spaced: 40 frames (> BT709HIP_MAX_BATCH) carved from two slabs
    w, h, n = 256, 4, 40
    frames = [rng.integers(0, 1 << 24, (h, w), dtype=np.uint32) for _ in range(n)]
    slab_in, slab_out = DeviceBuffer(ctx, n * w * h * 4), DeviceBuffer(ctx, n * w * h * 4)
    for i, words in enumerate(frames):
        ctx._upload(slab_in.ptr + i * w * h * 4, w * 4, words.view(np.uint8).reshape(h, w * 4), None)
    ctx._sync(None)
    ptrs = (C.c_void_p * n)(*[slab_in.ptr + i * w * h * 4 for i in range(n)])
    surfs = (_capi.Surface * n)(*[_capi.Surface(slab_out.ptr + i * w * h * 4, w * 4, w, h, 0, 0) for i in range(n)])
    assert lib.bt709hip_unconvert_batch(dec._handle, n, ptrs, w * 4, w, h, surfs, None, 1) == _capi.OK
    got = np.empty((n * h, w * 4), np.uint8)
    _capi.check(lib.bt709hip_download(ctx.handle, got.ctypes.data, w * 4, slab_out.ptr, w * 4, w * 4, n * h, None))
    ctx._sync(None)
    for i, words in enumerate(frames):
        assert np.array_equal(got[i * h:(i + 1) * h].view(np.uint32).reshape(-1), oracle.unconvert_packed(0, words, w, h)), i
    # 40 frames that are NOT evenly spaced exceed the pointer table; mixed sizes / formats / alpha decoders are refused
    ptrs[1], ptrs[2] = ptrs[2], ptrs[1]
    assert lib.bt709hip_unconvert_batch(dec._handle, n, ptrs, w * 4, w, h, surfs, None, 1) == _capi.ERR_UNSUPPORTED
    surfs[3].width = w - 2
    assert lib.bt709hip_unconvert_batch(dec._handle, 8, ptrs, w * 4, w, h, surfs, None, 1) == _capi.ERR_SIZE_MISMATCH
    surfs[3].width = w
    surfs[5].format = _capi.FORMAT_RGBA16F
    assert lib.bt709hip_unconvert_batch(dec._handle, 8, ptrs, w * 4, w, h, surfs, None, 1) == _capi.ERR_UNSUPPORTED
    adec = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)
    assert lib.bt709hip_unconvert_batch(adec._handle, 2, ptrs, w * 4, w, h, surfs, None, 1) == _capi.ERR_UNSUPPORTED
    assert lib.bt709hip_unconvert_batch(dec._handle, 0, ptrs, w * 4, w, h, surfs, None, 1) == _capi.OK
    assert lib.bt709hip_unconvert_batch(dec._handle, 2, None, w * 4, w, h, surfs, None, 1) == _capi.ERR_INVALID_ARG


def test_one_pass_route_nil_texture_and_render_pass_descriptor(gh, oracle):
    """AAPLRenderer.m:927-934: -decodeBT709: with bgraSRGBTexture:nil and the view's render pass descriptor -- the
    decoder renders into colorAttachments[0].texture (MetalBT709Decoder.m:272-281, 462-466).  Here the drawable is
    larger than the frame: the frame fills the renderWidth x renderHeight viewport at its origin, nothing else."""
    ctx = gh.context()
    w, h = 64, 16
    y, c = gh.random_nv12(w, h, seed=77)
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    buf = gh.make_buffer(y, c, dec.gamma)
    drawable = ctx.makeBGRATexture((w + 16, h + 4))
    _capi.check(ctx.lib.bt709hip_memset(ctx.handle, drawable.ptr, 0x3C, drawable.stride * drawable.height, None))
    rpd = mb.MTLRenderPassDescriptor(drawable)
    assert dec.decodeBT709(buf, None, None, ctx.commandQueue.commandBuffer(), rpd, w, h, True), dec.lastStatus
    got = ctx.getBGRATexturePixels(drawable).view(np.uint8).reshape(h + 4, (w + 16) * 4)
    assert np.array_equal(got[:h, :w * 4], oracle.decode_nv12(0, y, c))
    assert (got[h:] == 0x3C).all() and (got[:h, w * 4:] == 0x3C).all()
    # nil texture and nil descriptor; a drawable smaller than the frame; render size != frame size
    assert not dec.decodeBT709(buf, None, None, None, None, w, h, True) and dec.lastStatus == _capi.ERR_INVALID_ARG
    small = mb.MTLRenderPassDescriptor(ctx.makeBGRATexture((w - 2, h)))
    assert not dec.decodeBT709(buf, None, None, None, small, w, h, True) and dec.lastStatus == _capi.ERR_SIZE_MISMATCH
    assert not dec.decodeBT709(buf, None, None, None, rpd, w + 16, h + 4, True) and dec.lastStatus == _capi.ERR_SIZE_MISMATCH


def test_error_behaviour(gh):
    """FALSE on every validation failure, in the reference's order
    (MetalBT709Decoder.m:272-368)."""
    ctx = gh.context()
    y, c = gh.random_nv12(8, 4, seed=1)
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    good = gh.make_buffer(y, c, dec.gamma)
    tex = ctx.makeBGRATexture((8, 4))
    assert _decode_status(gh, dec, good, tex, 8, 4) == (True, _capi.OK)
    # output texture of another size (.m:272-282)
    assert _decode_status(gh, dec, good, ctx.makeBGRATexture((8, 6)), 8, 4) == (False, _capi.ERR_SIZE_MISMATCH)
    # render size differs (.m:284-290)
    assert _decode_status(gh, dec, good, tex, 4, 4) == (False, _capi.ERR_SIZE_MISMATCH)
    # untagged / BT.601 matrix (.m:311-318)
    bad = gh.make_buffer(y, c, dec.gamma)
    bad.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_601_4)
    assert _decode_status(gh, dec, bad, tex, 8, 4) == (False, _capi.ERR_MATRIX)
    # transfer tag vs configured gamma (.m:335-353)
    srgb_tagged = gh.make_buffer(y, c, mb.MetalBT709GammaSRGB)
    assert _decode_status(gh, dec, srgb_tagged, tex, 8, 4) == (False, _capi.ERR_TRANSFER)
    for gamma in (mb.MetalBT709GammaSRGB, mb.MetalBT709GammaLinear):
        d2 = gh.make_decoder(gamma)
        assert _decode_status(gh, d2, good, tex, 8, 4) == (False, _capi.ERR_TRANSFER)
        assert _decode_status(gh, d2, gh.make_buffer(y, c, gamma), tex, 8, 4) == (True, _capi.OK)
    # alpha buffer of another size (.m:294-306) and not tagged linear (.m:357-368)
    da = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)
    sb = gh.make_buffer(y, c, mb.MetalBT709GammaSRGB)
    a_small = gh.make_alpha_buffer(np.zeros((2, 8), np.uint8))
    assert _decode_status(gh, da, sb, tex, 8, 4, a_small) == (False, _capi.ERR_SIZE_MISMATCH)
    a_bad = gh.make_alpha_buffer(np.zeros((4, 8), np.uint8))
    a_bad.setAttachment("TransferFunction", mb.kCVImageBufferTransferFunction_sRGB)
    assert _decode_status(gh, da, sb, tex, 8, 4, a_bad) == (False, _capi.ERR_ALPHA_TRANSFER)
    assert _decode_status(gh, da, sb, tex, 8, 4, None) == (False, _capi.ERR_INVALID_ARG)
    # odd dimensions (BGRAToBT709Converter.m:69-74)
    odd = mb.CVPixelBuffer(ctx, 7, 4)
    mb.BGRAToBT709Converter.setBT709Attributes(odd)
    assert _decode_status(gh, dec, odd, ctx.makeBGRATexture((7, 4)), 7, 4) == (False, _capi.ERR_ODD_DIMENSIONS)
    # a decoder without a render context cannot set up (.m:48-54)
    orphan = mb.MetalBT709Decoder()
    assert orphan.setupMetal() is False and orphan.lastStatus == _capi.ERR_NOT_SETUP
    assert orphan.decodeBT709(good, None, tex, None, None, 8, 4, True) is False


def test_launch_limits_are_reported_not_launched(gh):
    """The row-pair dimension of every kernel lives in gridDim.y (<= 65535): taller surfaces are refused
    with ERR_UNSUPPORTED instead of failing as an opaque HIP launch error."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, h = 4, 2 * 65536 + 4
    buf = mb.CVPixelBuffer(ctx, w, h)
    mb.BGRAToBT709Converter.setBT709Attributes(buf)
    assert not dec.decodeBT709(buf, None, ctx.makeBGRATexture((w, h)), None, None, w, h, True)
    assert dec.lastStatus == _capi.ERR_UNSUPPORTED
    ok = mb.CVPixelBuffer(ctx, w, 2 * 65535)  # the tallest frame one launch takes
    mb.BGRAToBT709Converter.setBT709Attributes(ok)
    assert dec.decodeBT709(ok, None, ctx.makeBGRATexture((w, 2 * 65535)), None, None, w, 2 * 65535, True), dec.lastStatus
    small = gh.make_buffer(*gh.random_nv12(8, 4, seed=2), dec.gamma)
    assert not dec.decodeBT709Scaled(small, ctx.makeBGRATexture((2, 65536)), None, True)  # view-fit: one output row per gridDim.y
    assert dec.lastStatus == _capi.ERR_UNSUPPORTED
    tex = ctx.makeBGRATexture((4, 2 * 65536 + 4))
    assert not mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(tex, buf, mb.MetalBT709GammaSRGB, mb.MetalBT709GammaApple)
    # the any-ratio kernels form row offsets in 32 bits: a plane of 2 GiB or more (here: by its pitch; nothing is
    # dereferenced) is refused, not wrapped
    import ctypes as C
    view = ctx.makeBGRATexture((8, 8))
    like = small.frame()
    wide = _capi.Frame(like.y, 1 << 22, like.cbcr, 1 << 22, 16, 1024, like.matrix, like.transfer)
    surf = _capi.Surface(view.ptr, 32, 8, 8, _capi.FORMAT_BGRA8_SRGB, 0)
    assert ctx.lib.bt709hip_decode_scaled(dec._handle, C.byref(wide), None, C.byref(surf), None, 1) == _capi.ERR_UNSUPPORTED


def test_lazily_built_tables_are_refused_inside_a_capture(gh, oracle):
    """hipMalloc and blocking copies are illegal while a stream records a graph: an entry point that
    would have to build its tables there returns ERR_NOT_SETUP; after the matching *_prepare call the
    same recording works and replays correctly."""
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    w, hgt = 64, 16
    y, c = gh.random_nv12(w, hgt, seed=11)
    dec = gh.make_decoder(mb.MetalBT709GammaITU709)  # a decoder of its own: its RGBA16F table is not built yet
    buf = gh.make_buffer(y, c, dec.gamma)
    half = ctx.makeBGRATexture((w, hgt), pixelFormat=mb.MTLPixelFormatRGBA16Float)
    cb = ctx.commandQueue.commandBuffer(new_stream=True)
    cb.beginRecording()
    assert not dec.decodeBT709(buf, None, half, cb, None, w, hgt, False) and dec.lastStatus == _capi.ERR_NOT_SETUP
    cb.endRecording().release()
    assert lib.bt709hip_decoder_prepare_format(dec._handle, _capi.FORMAT_RGBA16F) == _capi.OK
    cb.beginRecording()
    assert dec.decodeBT709(buf, None, half, cb, None, w, hgt, False), dec.lastStatus
    rec = cb.endRecording()
    rec.replay(cb)
    cb.waitUntilCompleted()
    got = ctx.getBGRATexturePixels(half)
    assert np.array_equal(got.view(np.uint16), oracle.decode_nv12_rgba16f(mb.MetalBT709GammaITU709, y, c).view(np.uint16))
    rec.release()
    # a fresh decoder inside a capture: -setupMetal has not run yet
    fresh = mb.MetalBT709Decoder()
    fresh.metalRenderContext = ctx
    hnd = C.c_void_p()
    assert lib.bt709hip_decoder_create(h, 0, 0, C.byref(hnd)) == _capi.OK
    f8, s8 = gh.make_buffer(y, c, 0).frame(), ctx.makeBGRATexture((w, hgt)).surface()
    cb.beginRecording()
    assert lib.bt709hip_decode(hnd, C.byref(f8), None, C.byref(s8), w, hgt, cb.stream, 0) == _capi.ERR_NOT_SETUP
    cb.endRecording().release()
    assert lib.bt709hip_decode(hnd, C.byref(f8), None, C.byref(s8), w, hgt, cb.stream, 1) == _capi.OK
    lib.bt709hip_decoder_destroy(hnd)
    cb.release()


def test_rescale_entry_points_record_into_a_graph(gh, oracle):
    """The any-ratio kernel (whose launcher sizes a persistent grid with an occupancy query on first use), the exact
    2:1 kernels and pass 2 alone inside a recorded graph: first use INSIDE the recording, replayed twice."""
    ctx = gh.context()
    w, hgt = 256, 96
    y, c = gh.random_nv12(w, hgt, seed=21)
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    scale = mb.MetalScaleRenderContext()
    assert scale.setupRenderPipelines(ctx)
    buf = gh.make_buffer(y, c, dec.gamma)
    views = {(200, 50): ctx.makeBGRATexture((200, 50)), (128, 48): ctx.makeBGRATexture((128, 48)), (300, 130): ctx.makeBGRATexture((300, 130))}
    inter, view2 = ctx.makeBGRATexture((w, hgt)), ctx.makeBGRATexture((77, 31))
    cb = ctx.commandQueue.commandBuffer(new_stream=True)
    cb.beginRecording()
    for (ow, oh), tex in views.items():
        assert dec.decodeBT709Scaled(buf, tex, cb, False), dec.lastStatus
    assert dec.decodeBT709(buf, None, inter, cb, None, w, hgt, False)
    assert scale.renderScaled(ctx, view2, 77, 31, cb, None, inter, False)
    rec = cb.endRecording()
    for _ in range(2):
        rec.replay(cb)
    cb.waitUntilCompleted()
    for (ow, oh), tex in views.items():
        got = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(oh, ow * 4)
        want = oracle.decode_nv12_half(0, y, c) if (ow, oh) == (w // 2, hgt // 2) else oracle.decode_nv12_scaled(0, y, c, ow, oh)
        assert np.array_equal(got, want), (ow, oh)
    got = ctx.getBGRATexturePixels(view2).view(np.uint8).reshape(31, 77 * 4)
    assert np.array_equal(got, oracle.decode_nv12_scaled(0, y, c, 77, 31))
    rec.release()
    cb.release()


def test_copy_probe_and_stream_join(gh):
    """The benchmark's copy ceiling really copies (every byte, odd multiples of 16 included, nothing
    beyond), and bt709hip_stream_wait_event orders two streams without blocking the host."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    rng = np.random.default_rng(5)
    for nbytes in (16, 16 * 1023, 16 * 1025, 16 * 70001):
        src = rng.integers(0, 256, (1, nbytes), dtype=np.uint8)
        a, b = DeviceBuffer(ctx, nbytes), DeviceBuffer(ctx, nbytes + 64)
        ctx._upload(a.ptr, nbytes, src, None)
        _capi.check(lib.bt709hip_memset(h, b.ptr, 0x5A, nbytes + 64, None))
        _capi.check(lib.bt709hip_copy_probe(h, b.ptr, a.ptr, nbytes, None))
        out = np.empty((1, nbytes + 64), np.uint8)
        _capi.check(lib.bt709hip_download(h, out.ctypes.data, nbytes + 64, b.ptr, nbytes + 64, nbytes + 64, 1, None))
        ctx._sync(None)
        assert np.array_equal(out[0, :nbytes], src[0]) and (out[0, nbytes:] == 0x5A).all()
    assert lib.bt709hip_copy_probe(h, b.ptr, a.ptr, 24, None) == _capi.ERR_INVALID_ARG      # not a multiple of 16
    assert lib.bt709hip_copy_probe(h, b.ptr + 4, a.ptr, 16, None) == _capi.ERR_INVALID_ARG  # misaligned
    # producer on stream 1, consumer on stream 2 joined by an event: decode A -> copy A's pixels elsewhere
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, hgt = 256, 64
    y, c = gh.random_nv12(w, hgt, seed=91)
    buf, tex = gh.make_buffer(y, c, dec.gamma), ctx.makeBGRATexture((w, hgt))
    dst = DeviceBuffer(ctx, w * hgt * 4)
    cb1, cb2 = ctx.commandQueue.commandBuffer(new_stream=True), ctx.commandQueue.commandBuffer(new_stream=True)
    ev = C.c_void_p()
    _capi.check(lib.bt709hip_event_create(h, C.byref(ev)))
    for _ in range(20):
        assert dec.decodeBT709(buf, None, tex, cb1, None, w, hgt, False)
    _capi.check(lib.bt709hip_event_record(h, ev, cb1.stream))
    _capi.check(lib.bt709hip_stream_wait_event(h, cb2.stream, ev))
    _capi.check(lib.bt709hip_copy_probe(h, dst.ptr, tex.ptr, w * hgt * 4, cb2.stream))
    cb2.waitUntilCompleted()
    got = np.empty((hgt, w * 4), np.uint8)
    _capi.check(lib.bt709hip_download(h, got.ctypes.data, w * 4, dst.ptr, w * 4, w * 4, hgt, cb2.stream))
    cb2.waitUntilCompleted()
    assert np.array_equal(got, gh.gpu_decode(y, c, decoder=dec))
    assert lib.bt709hip_stream_wait_event(h, cb2.stream, None) == _capi.ERR_INVALID_ARG
    lib.bt709hip_event_destroy(h, ev)
    cb1.release()
    cb2.release()


def test_empty_frame_is_a_noop(gh):
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    buf = mb.CVPixelBuffer(ctx, 0, 0)
    mb.BGRAToBT709Converter.setBT709Attributes(buf)
    assert dec.decodeBT709(buf, None, ctx.makeBGRATexture((0, 0)), None, None, 0, 0, True)


def test_output_padding_is_untouched(gh):
    """Only width*4 bytes of each output row are written."""
    ctx = gh.context()
    w, h, stride = 64, 8, 64 * 4 + 64
    y, c = gh.random_nv12(w, h, seed=77)
    dec = gh.make_decoder()
    tex = ctx.makeBGRATexture((w, h), stride=stride)
    _capi.check(ctx.lib.bt709hip_memset(ctx.handle, tex.ptr, 0xAB, stride * h, None))
    assert dec.decodeBT709(gh.make_buffer(y, c, dec.gamma), None, tex, None, None, w, h, True)
    raw = np.empty((h, stride), np.uint8)
    _capi.check(ctx.lib.bt709hip_download(ctx.handle, raw.ctypes.data, stride, tex.ptr, stride, stride, h, None))
    ctx._sync(None)
    assert (raw[:, w * 4:] == 0xAB).all()


def test_cpp_host_mirror_host_memory_overload(gh, vectors, tmp_path):
    """The unchanged 8-argument selector for a caller whose CVPixelBuffer and texture are in HOST
    memory (what AAPLRenderer.m:927-957 and MetalBT709DecoderTests.m:248-255 hold): the C++ mirror's
    overload moves them through the in-flight pool, three frames in flight; the 28 reference vectors."""
    import subprocess
    from test_host_cpu import build_cpp_selftest
    exe = build_cpp_selftest(tmp_path)
    args = ["--host"]
    for r in vectors["metal_decode"]:
        args += [str(v) for v in r["ycbcr"] + r["rgb_out"]]
    res = subprocess.run([exe] + args, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "28 host vectors, 0 failures" in res.stdout


def test_cpp_host_mirror_two_pass_pipeline(gh, tmp_path):
    """C++ mirror of MetalScaleRenderContext: pass 1 into a BGRA8 / RGBA16Float intermediate, pass 2 on its
    own; the BGRA8 route must equal the fused decodeBT709Scaled."""
    import subprocess
    from test_host_cpu import build_cpp_selftest
    res = subprocess.run([build_cpp_selftest(tmp_path), "--two-pass"], capture_output=True, text=True)
    assert res.returncode == 0 and "two-pass pipeline, 0 failures" in res.stdout, res.stdout + res.stderr


def test_cpp_host_mirror_reference_vectors(gh, vectors, tmp_path):
    """The C++ twin of the reference's Metal decode test (host/decoder_selftest.cpp over
    host/MetalBT709Decoder.hpp) on the 28 reference vectors."""
    import subprocess
    from test_host_cpu import build_cpp_selftest
    exe = build_cpp_selftest(tmp_path)
    args = []
    for r in vectors["metal_decode"]:
        args += [str(v) for v in r["ycbcr"] + r["rgb_out"]]
    res = subprocess.run([exe] + args, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "28 vectors, 0 failures" in res.stdout


# ------------------------------------------------------------------ round 4: the launch the headline times, the ring, coalescing

def _tiled_ring(gh, ctx, gamma, n, w, h, tile_w, tile_h, seed, in_gap=0, out_gap=4096):
    """n frames of w x h at a constant pitch in two slabs (as a ring), frame i = a tiling of its OWN small random tile; a
    canary gap after every output frame.  Returns (slabs, bufs, texs, tiles, pitches)."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    in_pitch = (w * h * 3 // 2 + in_gap + 255) // 256 * 256
    out_pitch = (w * h * 4 + out_gap + 255) // 256 * 256
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    tiles = [gh.random_nv12(tile_w, tile_h, seed=seed + i) for i in range(n)]
    bufs, texs = [], []
    for i, (ty, tc) in enumerate(tiles):
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        b.setAttachment("YCbCrMatrix", mb.kCVImageBufferYCbCrMatrix_ITU_R_709_2)
        b.setAttachment("TransferFunction", gh.TRANSFER_FOR_GAMMA[gamma])
        b.upload_planes(np.tile(ty, (h // tile_h, w // tile_w)), np.tile(tc, (h // tile_h, w // tile_w)))
        bufs.append(b)
        texs.append(mb.BGRATexture(ctx, w, h, w * 4, ptr=slab_out.ptr + i * out_pitch))
    return (slab_in, slab_out), bufs, texs, tiles, (in_pitch, out_pitch)


def _launch_info(ctx):
    info = _capi.LaunchInfo()
    _capi.check(ctx.lib.bt709hip_last_launch_info(C.byref(info)))
    return info


@pytest.mark.parametrize("count", [256, 252])
def test_config3_256_frame_banded_launch_like_the_bench(gh, oracle, count):
    """BASELINE config 3 in the EXACT shape bench.py times: 256 evenly spaced 3840x2160 frames in ONE bt709hip_decode_batch
    (the XCD-aware work map: grid.x = 8 x tiles, grid.z = 32 frames per band, 64-bit frame offsets up to 8.5 GB), every
    frame a tiling of its own small tile (the tile checked against the oracle), EVERY frame compared, canaries between the
    output frames untouched.  252 frames: the map over 248 plus the plain map over the last 4 (two launches); the frames
    beyond the count stay unwritten.  Reference test flow: EmptyiOSTests/MetalBT709DecoderTests.m:189-277."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 256, 3840, 2160
    (slab_in, slab_out), bufs, texs, tiles, (_, out_pitch) = _tiled_ring(gh, ctx, mb.MetalBT709GammaApple, n, w, h, 256, 8, 30000)
    _capi.check(ctx.lib.bt709hip_memset(ctx.handle, slab_out.ptr, 0xC3, n * out_pitch, None))
    assert dec.decodeBT709Batch(bufs[:count], texs[:count], waitUntilCompleted=True), dec.lastStatus
    assert ctx.lib.bt709hip_last_kernel_name() == b"decode_nv12_quads<nt>"
    info = _launch_info(ctx)
    head = count - count % 8
    assert info.xcd_bands == 1 and info.launches == (1 if count % 8 == 0 else 2)
    assert list(info.grid) == [8 * 1, h // 2, head // 8] and list(info.block) == [512, 1, 1]  # 3840 wide: one 512-lane tile per row pair
    frame_bytes = w * h * 4
    raw = np.empty(out_pitch, np.uint8)
    for i, (ty, tc) in enumerate(tiles):
        _capi.check(ctx.lib.bt709hip_download(ctx.handle, raw.ctypes.data, out_pitch, slab_out.ptr + i * out_pitch, out_pitch,
                                              out_pitch, 1, None))
        ctx._sync(None)
        assert (raw[frame_bytes:] == 0xC3).all(), "canary after frame %d" % i
        got = raw[:frame_bytes].reshape(h, w * 4)
        if i < count:
            want = np.tile(oracle.decode_nv12(0, ty, tc), (h // 8, w // 256))
            assert np.array_equal(got, want), "frame %d" % i
        else:
            assert (got == 0xC3).all(), "frame %d beyond the launch was written" % i


def test_frame_ring_with_placement_hunt(gh, oracle):
    """bt709hip_ring_create: 64 x 1080p (0.7 GB: large enough to hunt), two candidates per slab asked for.  The hunt's
    bookkeeping is consistent, every frame of the ring it kept decodes to the oracle's bytes in one bt709hip_ring_decode
    launch (XCD-aware map), a sub-range decodes too, and tries = 1 allocates exactly one candidate per slab."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 64, 1920, 1080
    ring = mb.FrameRing(dec, (w, h), n, tries=2)
    p = ring.placement()
    assert p.tries == 2 and 1 <= p.in_candidates <= 2 and 2 <= p.out_candidates <= 6
    assert 0 <= p.chosen_in < p.in_candidates and 0 <= p.chosen_out < p.out_candidates
    kept = [k for k in p.out_kept if k >= 0]
    assert p.chosen_out in kept and len(kept) == min(2, p.out_candidates) and p.probes == p.in_candidates * len(kept)
    scan = list(p.out_prescan_GBps[:p.out_candidates])
    assert all(v > 100.0 for v in scan) and p.first_GBps == scan[0]
    assert p.worst_GBps <= p.best_GBps and p.chosen_GBps > 100.0
    # The default hunt is frugal (round 6): two outputs are alive at a time and the slower one makes room for the next candidate,
    # so what reaches the pairing probes is the FASTEST output of the prescan plus whichever candidate came last
    assert max(range(len(scan)), key=lambda o: scan[o]) in kept
    # with room for every candidate (a named budget) the kept outputs are the fastest of the prescan
    roomy = mb.FrameRing(dec, (w, h), n, tries=2, maxBytes=8 * n * (w * h * 3 // 2 + w * h * 4))
    pr = roomy.placement()
    rkept, rscan = [k for k in pr.out_kept if k >= 0], list(pr.out_prescan_GBps[:pr.out_candidates])
    assert pr.evicted == 0 and sorted(rkept) == sorted(sorted(range(len(rscan)), key=lambda o: -rscan[o])[:len(rkept)])
    roomy.release()
    tiles = [gh.random_nv12(240, 8, seed=41000 + i) for i in range(n)]
    for i, (ty, tc) in enumerate(tiles):
        ring.pixelBuffer(i).upload_planes(np.tile(ty, (h // 8, w // 240)), np.tile(tc, (h // 8, w // 240)))
    assert ring.decode(waitUntilCompleted=True)
    info = _launch_info(ctx)
    assert info.xcd_bands == 1 and info.launches == 1 and list(info.grid) == [8, h // 2 // 2, n // 8]  # 1080p: two row pairs per workgroup
    for i, (ty, tc) in enumerate(tiles):
        got = ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, np.tile(oracle.decode_nv12(0, ty, tc), (h // 8, w // 240))), i
    # a sub-range; argument errors
    _capi.check(ctx.lib.bt709hip_memset(ctx.handle, ring.texture(0).ptr, 0, n * w * h * 4, None))
    assert ring.decode(5, 3, waitUntilCompleted=True)
    for i in (4, 5, 7, 8):
        got = ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(h, w * 4)
        ty, tc = tiles[i]
        assert np.array_equal(got, np.tile(oracle.decode_nv12(0, ty, tc), (h // 8, w // 240))) == (5 <= i < 8)
    assert not ring.decode(60, 5) and dec.lastStatus == _capi.ERR_INVALID_ARG
    ring.release()
    one = mb.FrameRing(dec, (w, h), n, tries=1)
    q = one.placement()
    assert (q.tries, q.in_candidates, q.out_candidates, q.probes, q.chosen_in, q.chosen_out) == (1, 1, 1, 0, 0, 0)
    one.release()
    # a ring that fits the Infinity Cache never probes; a half-scale ring with an alpha decoder decodes through the 2:1 path
    adec = gh.make_decoder(mb.MetalBT709GammaSRGB, has_alpha=True)
    small = mb.FrameRing(adec, (64, 16), 3, halfScale=True, tries=6)
    assert small.placement().tries == 1
    fr = [gh.random_nv12(64, 16, seed=900 + i) for i in range(3)]
    al = [np.random.default_rng(950 + i).integers(0, 256, (16, 64), dtype=np.uint8) for i in range(3)]
    for i in range(3):
        small.pixelBuffer(i).upload_planes(*fr[i])
        ab = small.alphaPixelBuffer(i)
        ctx._upload(ab.y_ptr, ab.y_stride, al[i], None)
        ctx._sync(None)
    assert small.decode(waitUntilCompleted=True)
    for i in range(3):
        got = ctx.getBGRATexturePixels(small.texture(i)).view(np.uint8).reshape(8, 32 * 4)
        assert np.array_equal(got, oracle.decode_nv12_half(1, fr[i][0], fr[i][1], alpha=al[i])), i
    small.release()
    # a ring the device cannot hold (60 000 x 4K = 2.7 TB): a clean BT709HIP_ERR_HIP, nothing leaked, with and without a hunt
    lib = ctx.lib
    free0, free1, h_ = C.c_size_t(), C.c_size_t(), C.c_void_p()
    _capi.check(lib.bt709hip_mem_info(ctx.handle, C.byref(free0), None))
    for t in (1, 3):
        assert lib.bt709hip_ring_create(dec._handle, 3840, 2160, 60000, 0, t, C.byref(h_)) == _capi.ERR_HIP and not h_.value
    _capi.check(lib.bt709hip_mem_info(ctx.handle, C.byref(free1), None))
    assert abs(free1.value - free0.value) < (64 << 20), (free0.value, free1.value)
    # ... and the failed allocation does not linger as the thread's "last HIP error": the next decode succeeds (it used to
    # fail with a stale out-of-memory, found by this very test)
    ys, cs = gh.random_nv12(64, 16, seed=5)
    assert np.array_equal(gh.gpu_decode(ys, cs, mb.MetalBT709GammaApple, decoder=dec), oracle.decode_nv12(0, ys, cs))
    assert lib.bt709hip_malloc(ctx.handle, 1 << 50, C.byref(h_)) == _capi.ERR_HIP and lib.bt709hip_last_hip_error() != 0
    assert np.array_equal(gh.gpu_decode(ys, cs, mb.MetalBT709GammaApple, decoder=dec), oracle.decode_nv12(0, ys, cs))
    # argument errors, no GPU work
    assert lib.bt709hip_ring_create(dec._handle, 63, 16, 4, 0, 1, C.byref(h_)) == _capi.ERR_ODD_DIMENSIONS
    assert lib.bt709hip_ring_create(dec._handle, 66, 16, 4, 1, 1, C.byref(h_)) == _capi.ERR_ODD_DIMENSIONS
    assert lib.bt709hip_ring_create(dec._handle, 64, 16, 0, 0, 1, C.byref(h_)) == _capi.ERR_INVALID_ARG
    assert lib.bt709hip_ring_create(None, 64, 16, 4, 0, 1, C.byref(h_)) == _capi.ERR_INVALID_ARG


def test_coalescing_submit(gh, oracle):
    """BT709HIP_OPT_COALESCE: one-frame -decodeBT709: calls with waitUntilCompleted = FALSE are validated at once and queued;
    the queue goes out as ONE batch launch when it is full, when a differing call arrives, and through every stream-taking
    entry point (a read-back, a synchronize).  Same bytes as the oracle through every route; a failing call reports its own
    status immediately and disturbs nothing."""
    ctx = gh.context()
    lib = ctx.lib
    dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 8})
    w, h, n = 320, 24, 21
    frames = [gh.random_nv12(w, h, seed=52000 + i) for i in range(n)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]          # separate allocations: the pointer-table batch
    texs = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
    cb = ctx.commandQueue.commandBuffer()
    for t in texs:
        _capi.check(lib.bt709hip_memset(ctx.handle, t.ptr, 0x5A, w * h * 4, None))
    ctx._sync(None)
    queued_names = []
    for i in range(n):
        assert dec.decodeBT709(bufs[i], None, texs[i], cb, None, w, h, False), dec.lastStatus
        queued_names.append(lib.bt709hip_last_kernel_name())
    # 21 calls, 8 per launch: calls 8 and 16 issued a launch, the last five are still queued
    assert [i for i, nm in enumerate(queued_names) if nm != b"(queued: coalescing submit)"] == [7, 15]
    info = _launch_info(ctx)
    assert info.launches == 1 and info.grid[2] == 8
    # a call that fails validation: its own status, now; the queue is untouched
    bad = gh.make_buffer(*frames[0], dec.gamma, tag=False)
    assert not dec.decodeBT709(bad, None, texs[0], cb, None, w, h, False) and dec.lastStatus == _capi.ERR_MATRIX
    assert not dec.decodeBT709(bufs[0], None, texs[0], cb, None, w + 2, h, False) and dec.lastStatus == _capi.ERR_SIZE_MISMATCH
    # read-back of a queued frame's texture: the download flushes the queue first (stream order kept)
    got = ctx.getBGRATexturePixels(texs[n - 1]).view(np.uint8).reshape(h, w * 4)
    assert np.array_equal(got, oracle.decode_nv12(0, *frames[n - 1]))
    assert _launch_info(ctx).grid[2] == 5
    for i in range(n):
        got = ctx.getBGRATexturePixels(texs[i]).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, *frames[i])), i
    # a frame of another size flushes what is queued and starts a new queue; explicit flush; waitUntilCompleted = TRUE
    y2, c2 = gh.random_nv12(128, 16, seed=53000)
    b2, t2 = gh.make_buffer(y2, c2, dec.gamma), ctx.makeBGRATexture((128, 16))
    for i in range(3):
        assert dec.decodeBT709(bufs[i], None, texs[i], cb, None, w, h, False)
    assert dec.decodeBT709(b2, None, t2, cb, None, 128, 16, False)
    assert _launch_info(ctx).grid[2] == 3                       # the three queued frames went out
    assert dec.flush(cb) and _launch_info(ctx).grid[2] == 1     # then the odd one
    assert dec.decodeBT709(bufs[3], None, texs[3], cb, None, w, h, False)
    assert dec.decodeBT709(bufs[4], None, texs[4], cb, None, w, h, True)  # waits: queue first, then this frame
    assert lib.bt709hip_last_kernel_name().startswith(b"decode_nv12_quads")
    assert np.array_equal(ctx.getBGRATexturePixels(t2).view(np.uint8).reshape(16, 128 * 4), oracle.decode_nv12(0, y2, c2))
    # evenly spaced frames (a ring) gather into the uniform batch form; a second stream has a queue of its own
    ring = mb.FrameRing(dec, (w, h), 12, tries=1)
    for i in range(12):
        ring.pixelBuffer(i).upload_planes(*frames[i])
    cb2 = ctx.commandQueue.commandBuffer(new_stream=True)
    for i in range(12):
        assert dec.decodeBT709(ring.pixelBuffer(i), None, ring.texture(i), cb if i % 2 == 0 else cb2, None, w, h, False)
    cb.waitUntilCompleted()
    cb2.waitUntilCompleted()
    for i in range(12):
        got = ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, *frames[i])), i
    # turning the option off issues what is queued; afterwards calls launch at once
    assert dec.decodeBT709(bufs[5], None, texs[5], cb, None, w, h, False)
    dec.setOption(_capi.OPT_COALESCE, 0)
    assert _launch_info(ctx).grid[2] == 1
    assert dec.decodeBT709(bufs[6], None, texs[6], cb, None, w, h, False)
    assert lib.bt709hip_last_kernel_name().startswith(b"decode_nv12_quads")
    cb.waitUntilCompleted()
    cb2.release()
    ring.release()
    # the in-flight pool (raw stream copies behind the decode) stays correct with the option on
    pdec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 4})
    pool = mb.InFlightFramePool(pdec, (w, h), 3)
    slots = []
    for i in range(3):
        slot, yv, cv = pool.acquire()
        yv[:], cv[:] = frames[i]
        pool.submit(slot)
        slots.append(slot)
    for i, slot in enumerate(slots):
        assert np.array_equal(pool.wait(slot), oracle.decode_nv12(0, *frames[i]))
    pool.release()


def test_cpp_host_mirror_frame_ring_and_coalescing(gh, vectors, tmp_path):
    """C++ mirror of the two round-4 extensions (host/MetalBT709Decoder.hpp FrameRing, BT709HIP_OPT_COALESCE): the 28
    reference vectors as the frames of a device-resident ring, decoded in one launch and then by 28 one-frame
    -decodeBT709: calls (waitUntilCompleted FALSE) gathered four to a launch; the last frame's read-back issues the rest."""
    import subprocess
    from test_host_cpu import build_cpp_selftest
    exe = build_cpp_selftest(tmp_path)
    args = ["--ring"]
    for r in vectors["metal_decode"]:
        args += [str(v) for v in r["ycbcr"] + r["rgb_out"]]
    res = subprocess.run([exe] + args, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "28 ring frames, 0 failures" in res.stdout


def test_coalescing_submit_in_a_recorded_graph_and_from_two_threads(gh, oracle):
    """Queued frames and command-buffer recording: frames queued BEFORE beginRecording are issued (not recorded), frames
    submitted during the recording are recorded -- as one batch launch -- when the recording ends, and a replay decodes them
    again.  Then two threads share one coalescing decoder, each on a stream of its own (a queue per stream)."""
    import threading
    ctx = gh.context()
    lib = ctx.lib
    dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 16})
    w, h, n = 256, 16, 6
    frames = [gh.random_nv12(w, h, seed=61000 + i) for i in range(n + 1)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs = [ctx.makeBGRATexture((w, h)) for _ in range(n + 1)]
    cb = ctx.commandQueue.commandBuffer(new_stream=True)
    assert dec.decodeBT709(bufs[n], None, texs[n], cb, None, w, h, False)   # queued before the recording
    cb.beginRecording()                                                      # ... and issued by it, unrecorded
    assert _launch_info(ctx).grid[2] == 1
    for i in range(n):
        assert dec.decodeBT709(bufs[i], None, texs[i], cb, None, w, h, False)
        assert lib.bt709hip_last_kernel_name() == b"(queued: coalescing submit)"
    rec = cb.endRecording()                                                  # the queue is recorded here: one launch of n frames
    assert _launch_info(ctx).grid[2] == n
    cb.waitUntilCompleted()
    got = ctx.getBGRATexturePixels(texs[n]).view(np.uint8).reshape(h, w * 4)
    assert np.array_equal(got, oracle.decode_nv12(0, *frames[n]))
    for t in texs[:n]:                                                        # nothing of the recording has run yet
        _capi.check(lib.bt709hip_memset(ctx.handle, t.ptr, 0x11, w * h * 4, cb.stream))
    cb.waitUntilCompleted()
    assert (ctx.getBGRATexturePixels(texs[0]).view(np.uint8) == 0x11).all()
    for _ in range(2):
        rec.replay(cb)
    cb.waitUntilCompleted()
    for i in range(n):
        got = ctx.getBGRATexturePixels(texs[i]).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, *frames[i])), i
    rec.release()
    cb.release()

    # two threads, one decoder, a stream each
    per, rounds = 24, 3
    errs = []

    def worker(k):
        try:
            mine = ctx.commandQueue.commandBuffer(new_stream=True)
            fr = [gh.random_nv12(w, h, seed=62000 + 100 * k + i) for i in range(per)]
            bb = [gh.make_buffer(y, c, dec.gamma) for y, c in fr]
            tt = [ctx.makeBGRATexture((w, h)) for _ in range(per)]
            for _ in range(rounds):
                for i in range(per):
                    if not dec.decodeBT709(bb[i], None, tt[i], mine, None, w, h, False):
                        errs.append((k, i, dec.lastStatus))
                mine.waitUntilCompleted()
            for i in range(per):
                raw = np.empty((h, w * 4), np.uint8)
                _capi.check(lib.bt709hip_download(ctx.handle, raw.ctypes.data, w * 4, tt[i].ptr, tt[i].stride, w * 4, h, mine.stream))
                mine.waitUntilCompleted()
                if not np.array_equal(raw, oracle.decode_nv12(0, *fr[i])):
                    errs.append((k, i, "pixels"))
            mine.release()
        except Exception as exc:  # noqa: BLE001
            errs.append((k, repr(exc)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs[:5]


def test_bench_line_on_the_gpu():
    """bench.py end to end on the GPU with a short ring (48 x 4K: its input is still twice the Infinity Cache): one JSON line,
    the step's kernel named (not a sentinel or a probe), the product ring's placement report, both roofline fractions -- the
    hunted ring's and the first allocation's --, the 8-band parity spot check, a same-run copy figure.  (The headline itself
    is the driver's to run; this keeps the wiring under the GPU test step.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ring", "48", "--steps", "3", "--warmup", "1",
                        "--placement-tries", "2", "--no-cpu-baseline", "--no-smooth-leg"], capture_output=True, text=True,
                       timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    rf = d["roofline"]
    assert d["parity_spot_check"] == "ok" and len(d["parity_spot_frames"]) == 8 and d["value"] > 100.0
    assert rf["kernel"] == "decode_nv12_quads<nt>" and rf["bound"] == "hbm" and 0.3 < rf["frac"] < 1.0
    assert 0.3 < rf["first_allocation_frac"] < 1.0 and rf["same_run_copy_GBps"] > 1000.0
    assert rf["algorithmic_bytes_per_launch"] == 48 * 45_619_200
    pl = d["config"]["placement"]
    assert pl["tries"] == 2 and pl["rings_resident"] == ["first", "hunted"] and pl["pairings_probed"] >= 2
    assert abs(d["value"] - 48 * 3840 * 2160 / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 1e-3


@pytest.mark.gpu
def test_upload_and_download_own_nothing_of_a_pageable_buffer(gh):
    """bt709hip_upload / _download with ordinary (pageable) host memory are complete on return, like the reference's
    -fillBGRATexture: / -getBGRATexturePixels: (Renderer/MetalRenderContext.m:122-160): the source is overwritten and
    RELEASED right after the call -- 3.1 MB numpy arrays, i.e. mmap'ed blocks that go back to the kernel when freed: the
    pattern that gave a lab tool a GPU memory access fault in round 5 while the copy was still asynchronous -- and the bytes
    downloaded into a fresh buffer are checked without a synchronise in between.  Pinned memory stays asynchronous."""
    import ctypes as C
    ctx = gh.context()
    lib, h = ctx.lib, ctx.handle
    n, rounds = 3110400, 24
    dev = C.c_void_p()
    _capi.check(lib.bt709hip_malloc(h, n * rounds, C.byref(dev)))
    rng = np.random.default_rng(31)
    sums = []
    for i in range(rounds):
        buf = rng.integers(0, 256, n, dtype=np.uint8)
        sums.append(int(buf.astype(np.uint32).sum()))
        _capi.check(lib.bt709hip_upload(h, dev.value + i * n, n, buf.ctypes.data, n, n, 1, None))
        buf[:] = 0
        del buf  # the block is unmapped here
    for i in range(rounds):
        back = np.empty(n, np.uint8)
        _capi.check(lib.bt709hip_download(h, back.ctypes.data, n, dev.value + i * n, n, n, 1, None))
        assert int(back.astype(np.uint32).sum()) == sums[i], i  # filled on return
        del back
    pinned = C.c_void_p()
    _capi.check(lib.bt709hip_host_alloc(h, n, C.byref(pinned)))
    host = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(n,))
    host[:] = 0
    _capi.check(lib.bt709hip_download(h, pinned, n, dev.value, n, n, 1, None))
    _capi.check(lib.bt709hip_stream_synchronize(h, None))
    assert int(host.astype(np.uint32).sum()) == sums[0]
    del host
    _capi.check(lib.bt709hip_host_free(h, pinned))
    _capi.check(lib.bt709hip_free(h, dev))
