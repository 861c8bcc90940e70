"""BGRA -> NV12 encoder (the step before the decode path).

CPU part: the product's host-built encoder tables against the oracle, and the oracle's
encoder against numbers the reference asserts / produces.  GPU part (-m gpu): the HIP
encoder through the C ABI against the oracle, bit-exact, plus the full GPU round trip
BGRA -> NV12 -> BGRA against the reference's own exhaustive-histogram semantics.
"""
import ctypes as C

import numpy as np
import pytest

import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
from oracle_lib import GAMMA_APPLE, GAMMA_LINEAR, GAMMA_SRGB

PAIRS = [(GAMMA_SRGB, GAMMA_APPLE), (GAMMA_SRGB, GAMMA_SRGB), (GAMMA_LINEAR, GAMMA_LINEAR),
         (GAMMA_APPLE, GAMMA_APPLE), (GAMMA_SRGB, GAMMA_LINEAR)]
TABLE_ENCODE_APPLE = 4


# ------------------------------------------------------------------ CPU

def test_apple_encode_composite_is_a_threshold_function(oracle):
    """round(255*Apple196enc(v)) is monotone in v and equals its threshold table for every
    float in [0,1] -- the property the encoder's BT709_from_linear lookup rests on."""
    assert oracle.check_thresholds(TABLE_ENCODE_APPLE) == 0


def test_host_built_encode_thresholds_equal_oracle(oracle):
    lib = mb.load_library()
    t = np.zeros(255, np.float32)
    assert lib.bt709hip_gamma_thresholds(TABLE_ENCODE_APPLE, t.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert np.array_equal(t.view(np.uint32), oracle.thresholds(TABLE_ENCODE_APPLE).view(np.uint32))


def test_reference_subsample_vectors(oracle, refdata):
    """BT709_average_pixel_values outputs produced by the reference headers (golden)."""
    for b in refdata["subsample_blocks"]:
        assert list(oracle.subsample_block(b["rgb"], b["in"], b["out"])) == b["y4cbcr"]


def test_flat_block_equals_per_pixel_encode(oracle):
    """A 2x2 block of one colour subsamples to that colour's own (Y,Cb,Cr): the encode
    expectations of the reference's Metal tests are built this way
    (MetalBT709DecoderTests.m:137-185, encode type 'VImage' = the subsample path)."""
    rng = np.random.default_rng(11)
    for R, G, B in rng.integers(0, 256, (300, 3)):
        y4cbcr = oracle.subsample_block([int(R), int(G), int(B)] * 4, GAMMA_SRGB, GAMMA_APPLE)
        assert len(set(y4cbcr[:4])) == 1


# ------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    gpu_helpers.context()
    return gpu_helpers


def gpu_encode(gh, bgra_words, w, h, in_gamma, out_gamma, stride=None, y_stride=None, c_stride=None):
    ctx = gh.context()
    tex = ctx.makeBGRATexture((w, h), pixels=bgra_words, stride=stride)
    buf = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (w, h), y_stride, c_stride)
    ok = mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(tex, buf, in_gamma, out_gamma)
    return buf.download_planes() if ok else None


@pytest.mark.gpu
@pytest.mark.parametrize("pair", PAIRS)
@pytest.mark.parametrize("size", [(2, 2), (4, 2), (6, 4), (64, 16), (250, 6), (1920, 64)])
def test_gpu_encoder_matches_oracle(gh, oracle, pair, size):
    w, h = size
    rng = np.random.default_rng(w * 7 + h + pair[0] * 3 + pair[1])
    bgra = rng.integers(0, 1 << 32, w * h, dtype=np.uint32)
    got = gpu_encode(gh, bgra, w, h, *pair)
    assert got is not None
    want = oracle.encode_nv12(bgra & 0xFFFFFF, w, h, *pair)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.gpu
def test_gpu_encoder_general_layout(gh, oracle):
    """Odd strides / width % 4 != 0 take the scalar kernel; same bytes."""
    w, h = 18, 6
    bgra = np.random.default_rng(5).integers(0, 1 << 24, w * h, dtype=np.uint32)
    got = gpu_encode(gh, bgra, w, h, GAMMA_SRGB, GAMMA_APPLE, y_stride=19, c_stride=21)
    want = oracle.encode_nv12(bgra, w, h, GAMMA_SRGB, GAMMA_APPLE)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert "blocks" in mb.load_library().bt709hip_last_kernel_name().decode()


@pytest.mark.gpu
def test_gpu_encoder_reference_blocks(gh, refdata):
    """The reference-produced BT709_average_pixel_values vectors, through the GPU."""
    for b in refdata["subsample_blocks"]:
        rgb = b["rgb"]
        words = np.array([(rgb[3 * i] << 16) | (rgb[3 * i + 1] << 8) | rgb[3 * i + 2] for i in range(4)], np.uint32)
        y, c = gpu_encode(gh, words, 2, 2, b["in"], b["out"])
        assert [int(y[0, 0]), int(y[0, 1]), int(y[1, 0]), int(y[1, 1]), int(c[0, 0]), int(c[0, 1])] == b["y4cbcr"]


@pytest.mark.gpu
def test_gpu_encoder_xctest_average_of_4_vectors(gh, vectors):
    """The reference's own asserted 2x2-averaging numbers (CoreImageMetalFilterTests.m:1683-2096), through the GPU encoder."""
    for b in vectors["average_blocks"]:
        rgb = b["rgb"]
        words = np.array([(rgb[3 * i] << 16) | (rgb[3 * i + 1] << 8) | rgb[3 * i + 2] for i in range(4)], np.uint32)
        y, c = gpu_encode(gh, words, 2, 2, b["in"], b["out"])
        assert [int(y[0, 0]), int(y[0, 1]), int(y[1, 0]), int(y[1, 1]), int(c[0, 0]), int(c[0, 1])] == b["y4cbcr"], b["test"]


@pytest.mark.gpu
def test_gpu_encoder_all_grey_levels_and_primaries(gh, oracle):
    cols = [(v, v, v) for v in range(256)] + [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (0, 255, 255),
                                              (255, 0, 255)]
    w, h = 2 * len(cols), 2
    bgra = np.zeros((h, w), np.uint32)
    for i, (r, g, b) in enumerate(cols):
        bgra[:, 2 * i:2 * i + 2] = (r << 16) | (g << 8) | b
    for pair in PAIRS:
        got = gpu_encode(gh, bgra.reshape(-1), w, h, *pair)
        want = oracle.encode_nv12(bgra.reshape(-1), w, h, *pair)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.gpu
def test_gpu_round_trip_on_device(gh, oracle):
    """Encode then decode without leaving the GPU: flat 2x2 blocks of sampled colours come
    back within the error the reference's exhaustive Apple196 round trip allows (max
    channel error 2: CoreImageMetalFilterTests.m:676-691) and equal the oracle's round trip."""
    ctx = gh.context()
    rng = np.random.default_rng(42)
    n = 4096
    cols = rng.integers(0, 256, (n, 3))
    w, h = 2 * n, 2
    bgra = np.zeros((h, w), np.uint32)
    words = (cols[:, 0].astype(np.uint32) << 16) | (cols[:, 1].astype(np.uint32) << 8) | cols[:, 2].astype(np.uint32)
    bgra[:, 0::2] = words
    bgra[:, 1::2] = words
    tex = ctx.makeBGRATexture((w, h), pixels=bgra.reshape(-1))
    buf = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (w, h))
    mb.BGRAToBT709Converter.setBT709Attributes(buf)
    assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(tex, buf, mb.MetalBT709GammaSRGB, mb.MetalBT709GammaApple)
    out = ctx.makeBGRATexture((w, h))
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    assert dec.decodeBT709(buf, None, out, None, None, w, h, True)
    px = ctx.getBGRATexturePixels(out)
    got = np.stack([(px[0, 0::2] >> 16) & 0xFF, (px[0, 0::2] >> 8) & 0xFF, px[0, 0::2] & 0xFF], axis=1).astype(int)
    # the subsample path quantises to gamma-encoded bytes before the matrix, so it is a little
    # lossier than the per-pixel encoder's exhaustive bound of 2 (CoreImageMetalFilterTests.m:676-691)
    assert np.abs(got - cols).max() <= 4
    for i in range(0, n, 16):
        y4cbcr = oracle.subsample_block([int(v) for v in cols[i]] * 4, GAMMA_SRGB, GAMMA_APPLE)
        assert tuple(got[i]) == oracle.decode_pixel(GAMMA_APPLE, y4cbcr[0], y4cbcr[4], y4cbcr[5])


@pytest.mark.gpu
def test_constant_division_shortcut_is_exact_for_every_float(gh, tmp_path):
    """The encoder divides by 1.8556f / 1.5748f as q0 = x*rc, q = fma(fma(-c,q0,x), rc, q0).
    tools/div_exact.hip compares that with __fdiv_rn for all 2^32 float bit patterns; it must
    report zero mismatches for 1e-30 <= |x| <= 4 (the encoder's operands are within [-1.1, 1.1])."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "div_exact")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                        os.path.join(root, "tools", "div_exact.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout
    lines = [l for l in out.splitlines() if l.startswith("c=")]
    assert len(lines) == 2, out
    for l in lines:
        assert l.rstrip().endswith("<=4: 0"), l


@pytest.mark.gpu
def test_gpu_encoder_errors(gh):
    ctx = gh.context()
    tex = ctx.makeBGRATexture((8, 4))
    assert not mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(tex, mb.CVPixelBuffer(ctx, 8, 6), 1, 0)
    assert not mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(tex, mb.CVPixelBuffer(ctx, 8, 4), 3, 0)  # ITU: not an encoder gamma
    odd = ctx.makeBGRATexture((7, 4))
    assert not mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(odd, mb.CVPixelBuffer(ctx, 7, 4), 1, 0)
    assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(ctx.makeBGRATexture((0, 0)), mb.CVPixelBuffer(ctx, 0, 0), 1, 0)


@pytest.mark.gpu
def test_gpu_encoder_batch_matches_single(gh, oracle):
    """bt709hip_encode_batch: separately allocated pictures (pointer table) and a ring carved
    from one allocation (evenly spaced, beyond the table limit) give the single-call bytes."""
    from metalbt709decoder_amd.decoder import DeviceBuffer
    ctx = gh.context()
    w, h = 64, 12
    rng = np.random.default_rng(41)
    # pointer table
    pics = [rng.integers(0, 1 << 32, w * h, dtype=np.uint32) for _ in range(5)]
    texs = [ctx.makeBGRATexture((w, h), pixels=p) for p in pics]
    bufs = [mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (w, h)) for _ in pics]
    assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffers(texs, bufs, 1, 0)
    for p, b in zip(pics, bufs):
        want = oracle.encode_nv12(p & 0xFFFFFF, w, h, 1, 0)
        got = b.download_planes()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # evenly spaced ring, more pictures than BT709HIP_MAX_BATCH
    n = _capi.MAX_BATCH + 9
    in_pitch, out_pitch = w * h * 4, w * h * 3 // 2
    slab_in, slab_out = DeviceBuffer(ctx, n * in_pitch), DeviceBuffer(ctx, n * out_pitch)
    pics = [rng.integers(0, 1 << 32, w * h, dtype=np.uint32) for _ in range(n)]
    texs, bufs = [], []
    for i, p in enumerate(pics):
        t = mb.BGRATexture(ctx, w, h, w * 4, ptr=slab_in.ptr + i * in_pitch)
        ctx.fillBGRATexture(t, p)
        texs.append(t)
        base = slab_out.ptr + i * out_pitch
        bufs.append(mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h)))
    assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffers(texs, bufs, 1, 0)
    for p, b in zip(pics, bufs):
        want = oracle.encode_nv12(p & 0xFFFFFF, w, h, 1, 0)
        got = b.download_planes()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # 64 pictures (a multiple of 8, at the threshold): the XCD-aware work map; 70: 64 under the map + 6 plain; same bytes as the plain order
    for bands, n64 in ((1, 64), (0, 64), (1, 70)):
        _capi.check(ctx.lib.bt709hip_context_set_option(ctx.handle, _capi.CTX_OPT_XCD_BANDS, bands))
        s_in, s_out = DeviceBuffer(ctx, n64 * in_pitch), DeviceBuffer(ctx, n64 * out_pitch)
        pics64 = [rng.integers(0, 1 << 32, w * h, dtype=np.uint32) for _ in range(n64)]
        t64, b64 = [], []
        for i, p in enumerate(pics64):
            t = mb.BGRATexture(ctx, w, h, w * 4, ptr=s_in.ptr + i * in_pitch)
            ctx.fillBGRATexture(t, p)
            t64.append(t)
            base = s_out.ptr + i * out_pitch
            b64.append(mb.CVPixelBuffer(ctx, w, h, w, w, planes=(base, base + w * h)))
        assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffers(t64, b64, 1, 0)
        for p, b in zip(pics64, b64):
            want = oracle.encode_nv12(p & 0xFFFFFF, w, h, 1, 0)
            got = b.download_planes()
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), bands
    _capi.check(ctx.lib.bt709hip_context_set_option(ctx.handle, _capi.CTX_OPT_XCD_BANDS, 1))
    # shuffled: not evenly spaced any more -> the table limit applies; mixed sizes are refused
    order = list(range(n))
    order[2], order[5] = order[5], order[2]
    assert not mb.BGRAToBT709Converter.convertIntoCoreVideoBuffers([texs[i] for i in order], [bufs[i] for i in order], 1, 0)
    other = ctx.makeBGRATexture((w, h + 2))
    assert not mb.BGRAToBT709Converter.convertIntoCoreVideoBuffers(
        [texs[0], other], [bufs[0], mb.CVPixelBuffer(ctx, w, h + 2)], 1, 0)
