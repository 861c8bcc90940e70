"""RGBA16Float render targets and the stand-alone pass 2.

The reference renders pass 1 into RGBA16Float where sRGB texture writes are unavailable
(Renderer/AAPLRenderer.m:143-170) and always runs pass 2 as its own render pass
(Renderer/MetalScaleRenderContext.m:55-105).  Neither has a CPU twin or a test in the reference;
the pin is tests/golden/pass2.json: half codes of all 2^24 (Y,Cb,Cr) computed from the reference's own
matrix step and curve functions (oracle/ref_harness.c), and the fused-rescale goldens the two-pass
route must reproduce."""
import hashlib
import os

import numpy as np
import pytest

import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
from oracle_lib import GAMMA_NAMES

GAMMAS = [0, 1, 2, 3]


def _frame(w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h // 2, w), dtype=np.uint8)


# ------------------------------------------------------------------ CPU: oracle vs the reference-composed pin

@pytest.mark.parametrize("gamma", GAMMAS)
def test_oracle_half_table_equals_reference_composed_hash(oracle, pass2, gamma):
    table = oracle.half_table(gamma)
    assert hashlib.sha256(table.tobytes()).hexdigest() == pass2["rgba16f_table_sha256"][GAMMA_NAMES[gamma]]


def test_oracle_alpha_half_and_layout(oracle, pass2):
    y, c = _frame(64, 8, 1)
    a = np.arange(256, dtype=np.uint8).repeat(2).reshape(8, 64)
    out = oracle.decode_nv12_rgba16f(1, y, c, alpha=a)
    assert out.shape == (8, 64, 4) and out.dtype == np.float16
    assert [int(v) for v in out.view(np.uint16)[..., 3].reshape(-1)[::2]] == pass2["alpha_half_map"]
    opaque = oracle.decode_nv12_rgba16f(1, y, c)
    assert (opaque[..., 3] == np.float16(1.0)).all() and np.array_equal(opaque[..., :3], out[..., :3])
    # no curve: the half of the saturated non-linear value itself
    lin = oracle.decode_nv12_rgba16f(2, y, c)
    n = oracle.ycbcr_to_rgbn(int(y[3, 5]), int(c[1, 4]), int(c[1, 5]))
    assert np.array_equal(lin[3, 5, :3], n.astype(np.float16))


def test_oracle_two_passes_equal_the_fused_definition(oracle):
    """decode to 8-bit sRGB, then pass 2 alone == the fused decode+rescale the goldens pin."""
    for gamma, (w, h), (ow, oh) in [(0, (64, 32), (40, 20)), (1, (30, 18), (64, 40)), (3, (48, 24), (24, 12)), (2, (50, 22), (1, 1))]:
        y, c = _frame(w, h, w + ow)
        a = np.random.default_rng(h).integers(0, 256, (h, w), dtype=np.uint8) if gamma == 1 else None
        inter = oracle.decode_nv12(gamma, y, c, alpha=a)
        assert np.array_equal(oracle.render_scaled(inter, ow, oh), oracle.decode_nv12_scaled(gamma, y, c, ow, oh, alpha=a))


def test_oracle_render_scaled_from_rgba16f(oracle):
    """A float intermediate skips the 8-bit quantisation between the passes: flat fields survive
    exactly, and the result stays within one code of the 8-bit route on random content."""
    flat = np.zeros((6, 10, 4), np.float16)
    flat[...] = [0.25, 0.5, 1.0, 0.5]
    out = oracle.render_scaled(flat, 7, 4).reshape(-1, 4)
    assert (out == [255, 188, 137, 128]).all()  # B, G, R = sRGB(1.0, 0.5, 0.25), A = round(127.5)
    y, c = _frame(64, 32, 5)
    via16 = oracle.render_scaled(oracle.decode_nv12_rgba16f(0, y, c), 40, 20)
    via8 = oracle.decode_nv12_scaled(0, y, c, 40, 20)
    assert np.abs(via16.astype(int) - via8.astype(int)).max() <= 1


def test_half_lookup_correction_step(pass2):
    """The kernel's exactness argument, replayed on the host from the product's own table: the fast
    candidate is biased downward, so it lands on H(x) or H(x) - 1, and the one threshold above it gives
    H(x) back -- checked next to every threshold, for both candidates; a candidate above H is refused."""
    import ctypes as C
    lib = mb.load_library()
    fn = lib.bt709hip_half_lookup
    for gamma in GAMMAS:
        n = C.c_int()
        assert fn(gamma, 0.5, 0, C.byref(n)) >= 0
        if gamma == 2:
            assert n.value == 0  # no curve, no table
            assert fn(gamma, 0.5, 0, None) == 0x3800
            continue
        assert 5000 < n.value < 9000
        thr = (C.c_float * n.value)()
        assert lib.bt709hip_half_thresholds(gamma, thr, n.value) == n.value
        bits = np.array(list(thr), np.float32)
        bits = bits[np.isfinite(bits) & (bits > 0)].view(np.uint32)
        for b in bits[:: max(1, len(bits) // 400)]:
            for d in (-1, 0, 1):
                x = float(np.array([int(b) + d], np.uint32).view(np.float32)[0])
                want = fn(gamma, x, 0, None)
                assert fn(gamma, x, -1, None) == want, (gamma, x)
        assert fn(gamma, 0.5, 1, None) < 0


def test_half_candidate_is_H_or_H_minus_1_for_every_float(tmp_path, oracle):
    """The RGBA16Float kernel takes its candidate from a table of tangents indexed by a round-toward-zero binary16 bucket of
    the scaled channel (ONE fma; bucket 0 = everything below the curve's split point = the reference's exact product).  The
    host replay of exactly those operations over the product's own table, for EVERY float from 0 to 1.0 (1.07 billion per
    gamma), against the oracle's curve with the reference's double pow: the candidate never exceeds the true value, its half
    is H or H - 1 -- what the single-threshold settlement needs -- and below the split point it is H itself, the bucket
    boundary sitting exactly on the split (tests/native/half_candidate_sweep.cpp).  The GPU sweeps over all 2^24 triples
    check the kernel's end result."""
    import ctypes as C
    import subprocess
    out = str(tmp_path / "libhalf_candidate_sweep.so")
    HERE = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(HERE)
    odir, csrc = os.path.join(root, "oracle"), os.path.join(root, "metalbt709decoder_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-Wall", "-Werror",
                        "-I", csrc, "-I", odir, os.path.join(HERE, "native", "half_candidate_sweep.cpp"),
                        os.path.join(csrc, "transfer_tables.cpp"), "-o", out, "-L", odir, "-loracle", "-Wl,-rpath," + odir,
                        "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(out)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    for gamma in (0, 1, 3):  # Apple, sRGB, ITU-709: the curves (LINEAR has no curve and no candidate)
        res = (C.c_uint64 * 6)()
        assert lib.sweep_half_candidate(gamma, threads, res) == 0
        swept, above, outside, below, first, curve = list(res)
        assert swept == 0x3f800001 and curve > 30_000_000 and above == 0 and outside == 0, (gamma, swept, above, outside, hex(first))
        assert 0 < below < curve // 4  # the tangent is low by up to a fifth of a half's spacing: H - 1 happens, not often
    assert lib.sweep_half_candidate(2, threads, (C.c_uint64 * 6)()) == -1


# ------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    gpu_helpers.context()
    return gpu_helpers


def gpu_decode_rgba16f(gh, y, c, gamma, alpha=None, stride=None, y_stride=None, cbcr_stride=None):
    ctx = gh.context()
    h, w = y.shape
    dec = gh.make_decoder(gamma, has_alpha=alpha is not None)
    buf = gh.make_buffer(y, c, dec.gamma, y_stride, cbcr_stride)
    abuf = gh.make_alpha_buffer(alpha) if alpha is not None else None
    tex = ctx.makeBGRATexture((w, h), stride=stride, pixelFormat=mb.MTLPixelFormatRGBA16Float)
    assert dec.decodeBT709(buf, abuf, tex, ctx.commandQueue.commandBuffer(), None, w, h, True), dec.lastStatus
    assert b"rgba16f" in ctx.lib.bt709hip_last_kernel_name()
    return ctx.getBGRATexturePixels(tex)


@pytest.mark.gpu
@pytest.mark.parametrize("gamma", GAMMAS)
def test_gpu_rgba16f_exhaustive_sweep(gh, pass2, gamma):
    """Every (Y,Cb,Cr) once into an RGBA16Float target: the three half codes must hash to what the
    REFERENCE's matrix step and curve functions produce (tests/golden/pass2.json)."""
    y, c = gh.exhaustive_frame()
    out = gpu_decode_rgba16f(gh, y, c, gamma).view(np.uint16)  # (4096, 4096, 4)
    assert (out[..., 3] == 0x3C00).all()
    cb = np.repeat(np.repeat(c[:, 0::2], 2, axis=0), 2, axis=1).astype(np.uint32)
    cr = np.repeat(np.repeat(c[:, 1::2], 2, axis=0), 2, axis=1).astype(np.uint32)
    idx = ((y.astype(np.uint32) << 16) | (cb << 8) | cr).reshape(-1)
    table = np.zeros((1 << 24, 3), np.uint16)
    table[idx] = out.reshape(-1, 4)[:, :3]
    assert hashlib.sha256(table.tobytes()).hexdigest() == pass2["rgba16f_table_sha256"][GAMMA_NAMES[gamma]]


@pytest.mark.gpu
@pytest.mark.parametrize("gamma", GAMMAS)
@pytest.mark.parametrize("size", [(2, 2), (6, 4), (250, 30), (1028, 6), (1920, 64)])
def test_gpu_rgba16f_frames(gh, oracle, gamma, size):
    w, h = size
    y, c = _frame(w, h, w + h + gamma)
    got = gpu_decode_rgba16f(gh, y, c, gamma)
    assert np.array_equal(got.view(np.uint16), oracle.decode_nv12_rgba16f(gamma, y, c).view(np.uint16))


@pytest.mark.gpu
def test_gpu_rgba16f_alpha_strides_and_padding(gh, oracle):
    """Linear alpha stored unquantised; odd plane pitches (byte loads), an 8-byte-aligned target pitch
    (8-byte stores), nothing written outside the rows."""
    ctx = gh.context()
    w, h = 70, 12
    y, c = _frame(w, h, 9)
    a = np.random.default_rng(3).integers(0, 256, (h, w), dtype=np.uint8)
    got = gpu_decode_rgba16f(gh, y, c, mb.MetalBT709GammaSRGB, alpha=a)
    assert np.array_equal(got.view(np.uint16), oracle.decode_nv12_rgba16f(1, y, c, alpha=a).view(np.uint16))
    for ys, cs, os_ in ((71, 73, w * 8 + 8), (72, 70, w * 8 + 24), (70, 70, w * 8)):
        dec = gh.make_decoder(mb.MetalBT709GammaApple)
        buf = gh.make_buffer(y, c, dec.gamma, ys, cs)
        tex = ctx.makeBGRATexture((w, h), stride=os_, pixelFormat=mb.MTLPixelFormatRGBA16Float)
        _capi.check(ctx.lib.bt709hip_memset(ctx.handle, tex.ptr, 0x5A, os_ * h, None))
        assert dec.decodeBT709(buf, None, tex, None, None, w, h, True), dec.lastStatus
        raw = np.empty((h, os_), np.uint8)
        _capi.check(ctx.lib.bt709hip_download(ctx.handle, raw.ctypes.data, os_, tex.ptr, os_, os_, h, None))
        ctx._sync(None)
        want = oracle.decode_nv12_rgba16f(0, y, c).view(np.uint8).reshape(h, w * 8)
        assert np.array_equal(raw[:, :w * 8], want) and (raw[:, w * 8:] == 0x5A).all(), (ys, cs, os_)


@pytest.mark.gpu
def test_gpu_rgba16f_batch_4k(gh, oracle):
    """The reference's fallback intermediate at BASELINE config 3's geometry: 4 x 4K frames, one launch."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 4, 3840, 2160
    frames = [_frame(w, h, 60 + i) for i in range(n)]
    bufs = [gh.make_buffer(y, c, dec.gamma) for y, c in frames]
    texs = [ctx.makeBGRATexture((w, h), pixelFormat=mb.MTLPixelFormatRGBA16Float) for _ in range(n)]
    assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True), dec.lastStatus
    for (y, c), t in zip(frames, texs):
        got = ctx.getBGRATexturePixels(t).view(np.uint16)
        for r0 in (0, 1000, h - 8):
            want = oracle.decode_nv12_rgba16f(0, y[r0:r0 + 8], c[r0 // 2:r0 // 2 + 4]).view(np.uint16)
            assert np.array_equal(got[r0:r0 + 8], want)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(128, 8), (252, 62)])
@pytest.mark.parametrize("n", [72, 77])
def test_gpu_rgba16f_xcd_band_work_map(gh, oracle, n, size):
    """72 evenly spaced frames in one launch (a multiple of 8, past the threshold), and 77 (72 under the map + 5 plain): the
    XCD-aware work map of the RGBA16Float kernel against the plain order and the oracle.  128 x 8: few workgroups, the small
    shape (2 blocks x 3 row pairs per lane); 252 x 62: >= 4 workgroups per CU, the large shape (4 x 2), with a ragged last tile
    (126 blocks on 4 x 64 lanes) and a ragged last row-pair group (31 row pairs in groups of 2)."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    w, h = size
    in_pitch, out_pitch = w * h * 3 // 2, w * h * 8
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    frames = [_frame(w, h, 500 + i) for i in range(n)]
    bufs, texs = [], []
    for i, (y, c) in enumerate(frames):
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        mb.BGRAToBT709Converter.setBT709Attributes(b)
        b.upload_planes(y, c)
        bufs.append(b)
        texs.append(mb.BGRATexture(ctx, w, h, w * 8, ptr=slab_out.ptr + i * out_pitch, pixelFormat=mb.MTLPixelFormatRGBA16Float))
    for bands in (1, 0):
        dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_XCD_BANDS: bands})
        _capi.check(ctx.lib.bt709hip_memset(ctx.handle, slab_out.ptr, 0, n * out_pitch, None))
        assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True), dec.lastStatus
        for (y, c), t in zip(frames, texs):
            assert np.array_equal(ctx.getBGRATexturePixels(t).view(np.uint16), oracle.decode_nv12_rgba16f(0, y, c).view(np.uint16)), bands


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (frames, (width, height), band option) -> (grid, block, bands used): the shape bt709_rgba16f.hip's launcher must pick
    (16, (1920, 1080), 1, ((8, 270, 2), 256, 1)),   # large shape (4 blocks x 2 row pairs), 240 of 256 lanes busy, XCD map from 8 frames on
    (16, (1920, 1080), 0, ((1, 270, 16), 256, 0)),  # the same launch under the plain map
    (6, (1920, 1080), 1, ((1, 270, 6), 256, 0)),    # not a multiple of 8 and short: plain map, one launch
    (288, (64, 1026), 1, ((8, 22, 36), (64, 8), 1)),  # narrow shape (4 x 3): >= 48 workgroups per slot on rows of <= 1 024 blocks; 8 slices of one wave, 171 groups on 176 slices
    (64, (1280, 720), 1, ((8, 90, 8), (192, 2), 1)),  # rows of 160 busy lanes: two SLICES per workgroup share the staged table
    (96, (640, 362), 0, ((1, 23, 96), (128, 4), 0)),  # four slices of 128 lanes, 91 row-pair groups: the last workgroup's last slice is past the frame
    (3, (3840, 2160), 1, ((1, 540, 3), 512, 0)),     # 1 620 workgroups in the large shape, past one per slot: large; 480 of 512 lanes busy
    (1, (3840, 2160), 1, ((1, 360, 1), 960, 0)),    # a single 4K frame: the small shape (2 x 3, 960 lanes)
])
def test_gpu_rgba16f_work_shapes(gh, oracle, case):
    """The three work shapes of the RGBA16Float kernel, its slices and both work maps, each asserted through bt709hip_last_launch_info
    and checked against the oracle: rows of whole frames at the start, in the middle and at the end of the launch (every
    frame for the small ones)."""
    import ctypes as C
    from metalbt709decoder_amd.decoder import DeviceBuffer
    n, (w, h), bands, (grid, block, used) = case
    ctx = gh.context()
    if ctx.info().compute_units != 256:
        pytest.skip("shape thresholds asserted for 256 CUs")
    in_pitch, out_pitch = w * h * 3 // 2, w * h * 8
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    distinct = [_frame(w, h, 900 + i) for i in range(min(n, 5))]
    bufs, texs = [], []
    for i in range(n):
        y, c = distinct[i % len(distinct)]
        base = slab_in.ptr + i * in_pitch
        b = mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h))
        mb.BGRAToBT709Converter.setBT709Attributes(b)
        b.upload_planes(y, c)
        bufs.append(b)
        texs.append(mb.BGRATexture(ctx, w, h, w * 8, ptr=slab_out.ptr + i * out_pitch, pixelFormat=mb.MTLPixelFormatRGBA16Float))
    dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_XCD_BANDS: bands})
    _capi.check(ctx.lib.bt709hip_memset(ctx.handle, slab_out.ptr, 0, n * out_pitch, None))
    assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True), dec.lastStatus
    info = _capi.LaunchInfo()
    _capi.check(ctx.lib.bt709hip_last_launch_info(C.byref(info)))
    block = block if isinstance(block, tuple) else (block, 1)
    assert (tuple(info.grid), (info.block[0], info.block[1]), info.xcd_bands, info.launches) == (grid, block, used, 1)
    want = [oracle.decode_nv12_rgba16f(0, y, c).view(np.uint16) for y, c in distinct]
    for i in sorted(set(range(n)) if n <= 16 else {0, 1, n // 8 - 1, n // 8, n // 2, n - n // 8 - 1, n - 2, n - 1}):
        assert np.array_equal(ctx.getBGRATexturePixels(texs[i]).view(np.uint16), want[i % len(distinct)]), i


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(0, (64, 32), (40, 20)), (1, (30, 18), (64, 40)), (3, (1920, 64), (1280, 43)),
                                  (2, (50, 22), (1, 1)), (0, (48, 24), (24, 12))])
def test_gpu_two_passes_equal_fused(gh, oracle, case):
    """The reference's literal pipeline -- -decodeBT709 into an intermediate, then -renderScaled: --
    through both intermediate formats; the BGRA8 route must equal the fused kernel bit for bit."""
    gamma, (w, h), (ow, oh) = case
    ctx = gh.context()
    y, c = _frame(w, h, w + ow + gamma)
    a = np.random.default_rng(ow).integers(0, 256, (h, w), dtype=np.uint8) if gamma == 1 else None
    dec = gh.make_decoder(gamma, has_alpha=a is not None)
    buf = gh.make_buffer(y, c, dec.gamma)
    abuf = gh.make_alpha_buffer(a) if a is not None else None
    scale = mb.MetalScaleRenderContext()
    assert scale.setupRenderPipelines(ctx)
    view = ctx.makeBGRATexture((ow, oh))
    fused = gh.gpu_decode_scaled(y, c, (ow, oh), gamma, dec, alpha=a)
    for fmt in (mb.MTLPixelFormatBGRA8Unorm_sRGB, mb.MTLPixelFormatRGBA16Float):
        inter = ctx.makeBGRATexture((w, h), pixelFormat=fmt)
        cb = ctx.commandQueue.commandBuffer()
        assert dec.decodeBT709(buf, abuf, inter, cb, None, w, h, False), dec.lastStatus
        assert scale.renderScaled(ctx, view, ow, oh, cb, None, inter, True), scale.lastStatus
        got = ctx.getBGRATexturePixels(view).view(np.uint8).reshape(oh, ow * 4)
        if fmt == mb.MTLPixelFormatBGRA8Unorm_sRGB:
            assert np.array_equal(got, fused)
            assert np.array_equal(got, oracle.render_scaled(oracle.decode_nv12(gamma, y, c, alpha=a), ow, oh))
        else:
            assert np.array_equal(got, oracle.render_scaled(oracle.decode_nv12_rgba16f(gamma, y, c, alpha=a), ow, oh))
    assert not scale.renderScaled(ctx, view, ow + 1, oh, None, None, inter, True) and scale.lastStatus == _capi.ERR_SIZE_MISMATCH


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [mb.MTLPixelFormatBGRA8Unorm_sRGB, mb.MTLPixelFormatRGBA16Float])
def test_gpu_render_scaled_batch(gh, oracle, fmt):
    """bt709hip_render_scaled_batch: pass 2 over a ring of intermediates in ONE launch equals one call per surface and the
    oracle, writes nothing outside its rows; unevenly spaced or mismatched surfaces are refused."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    n, (w, h), (ow, oh) = 5, (96, 40), (61, 27)
    bpp = 8 if fmt == mb.MTLPixelFormatRGBA16Float else 4
    in_stride, out_stride = w * bpp + 32, ow * 4 + 16
    in_pitch, out_pitch = in_stride * h + 64, out_stride * oh
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    dec = gh.make_decoder(0)
    frames = [_frame(w, h, 900 + i) for i in range(n)]
    inters = [mb.BGRATexture(ctx, w, h, in_stride, ptr=slab_in.ptr + i * in_pitch, pixelFormat=fmt) for i in range(n)]
    views = [mb.BGRATexture(ctx, ow, oh, out_stride, ptr=slab_out.ptr + i * out_pitch) for i in range(n)]
    for (y, c), t in zip(frames, inters):
        assert dec.decodeBT709(gh.make_buffer(y, c, dec.gamma), None, t, None, None, w, h, True), dec.lastStatus
    scale = mb.MetalScaleRenderContext()
    assert scale.setupRenderPipelines(ctx)
    _capi.check(ctx.lib.bt709hip_memset(ctx.handle, slab_out.ptr, 0x5A, n * out_pitch, None))
    assert scale.renderScaledBatch(ctx, views, None, inters, True), scale.lastStatus
    raw = np.empty((n * oh, out_stride), np.uint8)
    _capi.check(ctx.lib.bt709hip_download(ctx.handle, raw.ctypes.data, out_stride, slab_out.ptr, out_stride, out_stride, n * oh, None))
    ctx._sync(None)
    assert (raw[:, ow * 4:] == 0x5A).all()
    got = raw[:, :ow * 4].reshape(n, oh, ow * 4)
    for i, (y, c) in enumerate(frames):
        inter = oracle.decode_nv12_rgba16f(0, y, c) if bpp == 8 else oracle.decode_nv12(0, y, c)
        assert np.array_equal(got[i], oracle.render_scaled(inter, ow, oh)), i
        single = ctx.makeBGRATexture((ow, oh))
        assert scale.renderScaled(ctx, single, ow, oh, None, None, inters[i], True), scale.lastStatus
        assert np.array_equal(ctx.getBGRATexturePixels(single).view(np.uint8).reshape(oh, -1), got[i]), i
    # not evenly spaced: refused before anything is launched
    assert not scale.renderScaledBatch(ctx, [views[0], views[2], views[3]], None, [inters[0], inters[1], inters[2]], True)
    assert scale.lastStatus == _capi.ERR_UNSUPPORTED
    other = ctx.makeBGRATexture((w + 2, h), pixelFormat=fmt)
    assert not scale.renderScaledBatch(ctx, views[:2], None, [inters[0], other], True) and scale.lastStatus == _capi.ERR_SIZE_MISMATCH
    assert not scale.renderScaledBatch(ctx, views[:2], None, inters[:1], True) and scale.lastStatus == _capi.ERR_INVALID_ARG


@pytest.mark.gpu
def test_gpu_rgba16f_frame_ring(gh, oracle):
    """A frame ring with RGBA16Float render targets (bt709hip_ring_options.format): descriptors carry the format and the 8-byte
    texels, the placement hunt runs with the RGBA16F launch as its probe (the ring is large enough to be hunted), every frame of
    one launch equals the oracle; the half-scale combination is refused."""
    import ctypes as C
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    w, h, n = 1920, 1080, 32  # 32 x (3.1 + 16.6) MB = 630 MB: above the hunt's 256 MB floor
    ring = mb.FrameRing(dec, (w, h), n, tries=2, pixelFormat=mb.MTLPixelFormatRGBA16Float)
    frames = [_frame(w, h, 1200 + i) for i in range(3)]
    for i in range(n):
        ring.pixelBuffer(i).upload_planes(*frames[i % 3])
    o = _capi.Surface()
    _capi.check(ctx.lib.bt709hip_ring_frame(ring.handle, 1, None, None, C.byref(o)))
    assert o.format == _capi.FORMAT_RGBA16F and o.stride == w * 8 and (o.width, o.height) == (w, h)
    pl = ring.placement()
    assert pl.tries == 2 and pl.out_candidates >= 2 and pl.chosen_GBps > 100.0 and pl.probes >= 2
    assert ring.decode(0, n, waitUntilCompleted=True), dec.lastStatus
    assert b"rgba16f" in ctx.lib.bt709hip_last_kernel_name()
    want = [oracle.decode_nv12_rgba16f(0, y, c).view(np.uint16) for y, c in frames]
    for i in (0, 1, 2, 7, 8, 16, n - 1):
        assert np.array_equal(ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint16), want[i % 3]), i
    ring.release()
    with pytest.raises(RuntimeError):
        mb.FrameRing(dec, (64, 32), 4, halfScale=True, pixelFormat=mb.MTLPixelFormatRGBA16Float)
