"""Pins the CPU oracle (oracle/bt709_oracle.c) to the reference:

  * numbers asserted by the reference's own XCTest files (tests/golden/vectors.json,
    parsed by tests/golden/make_golden.py, file:line kept per record),
  * outputs of the reference's header functions run in the build container
    (tests/golden/reference.json + patterns.npz),
  * and, when /root/reference is present, the reference headers themselves, live.

CPU only.  The oracle is the checker for the GPU parity tests, so it is pinned first.
"""
import hashlib

import os

import numpy as np
import pytest

from oracle_lib import GAMMA_APPLE, GAMMA_ITU709, GAMMA_LINEAR, GAMMA_NAMES, GAMMA_SRGB

GAMMAS = [GAMMA_APPLE, GAMMA_SRGB, GAMMA_LINEAR, GAMMA_ITU709]


# ---------------------------------------------------------------- XCTest vectors

def test_metal_decode_vectors(oracle, vectors):
    """28 (Y,Cb,Cr)->(R,G,B) expectations of the Metal decoder, default Apple gamma
    (EmptyiOSTests/MetalBT709DecoderTests.m:281-2239)."""
    recs = vectors["metal_decode"]
    assert len(recs) == 28
    for r in recs:
        got = oracle.decode_pixel(GAMMA_APPLE, *r["ycbcr"])
        assert list(got) == r["rgb_out"], (r["test"], r["src"], got)


def test_metal_vectors_through_nv12_frame(oracle, vectors):
    """Same vectors through the frame path the test helper uses: a 2x2 frame of one
    colour -> copyBT709ToCoreVideo -> decode (MetalBT709DecoderTests.m:189-277)."""
    for r in vectors["metal_decode"]:
        Y, Cb, Cr = r["ycbcr"]
        packed = np.full(4, (Cr << 16) | (Cb << 8) | Y, dtype=np.uint32)
        y, uv = oracle.packed_to_nv12(packed, 2, 2)
        out = oracle.decode_nv12(GAMMA_APPLE, y, uv).reshape(2, 2, 4)
        R, G, B = r["rgb_out"]
        assert (out[..., 0] == B).all() and (out[..., 1] == G).all() and (out[..., 2] == R).all()
        assert (out[..., 3] == 0xFF).all()


def test_converter_software_vectors(oracle, vectors):
    """Grey encode+decode expectations of the `Software` converter path
    (EmptyiOSTests/AppleEncodeDecodeBT709Tests.m:1464-3065).  vImage-typed tests pin
    Apple's closed Accelerate path and are not ours."""
    n = 0
    for r in vectors["converter"]:
        if r["encode_type"] == "Software":
            assert list(oracle.encode_pixel(GAMMA_APPLE, *r["rgb_in"])) == r["ycbcr"], r["test"]
        if r["decode_type"] in ("Software", "Metal"):
            assert list(oracle.decode_pixel(GAMMA_APPLE, *r["ycbcr"])) == r["rgb_out"], r["test"]
            n += 1
    assert n >= 14


def test_converter_software_frame_path(oracle, vectors):
    """unconvertSoftware semantics: packed words in, (R<<16)|(G<<8)|B with alpha 0 out."""
    for r in vectors["converter"]:
        if r["decode_type"] != "Software":
            continue
        Y, Cb, Cr = r["ycbcr"]
        packed = np.full(4, (Cr << 16) | (Cb << 8) | Y, dtype=np.uint32)
        out = oracle.unconvert_packed(GAMMA_APPLE, packed, 2, 2)
        R, G, B = r["rgb_out"]
        assert (out == ((R << 16) | (G << 8) | B)).all()


def test_direct_c_vectors(oracle, vectors):
    """Direct calls into BT709.h (EmptyiOSTests/CoreImageMetalFilterTests.m:94-416, 851-1071)."""
    seen = 0
    for r in vectors["direct_c"]:
        calls = r["calls"]
        if calls == ["BT709_convertLinearRGBToYCbCr", "BT709_convertYCbCrToRGB"]:
            g = r["applyGammaMap"]
            assert list(oracle.encode_linear_pixel(g, *r["rgb_in"])) == r["ycbcr"], r["test"]
            assert list(oracle.decode_to_linear_pixel(g, *r["ycbcr"])) == r["rgb_out"], r["test"]
        elif calls == ["BT709_from_sRGB_convertRGBToYCbCr", "BT709_to_sRGB_convertYCbCrToRGB"]:
            assert list(oracle.encode_pixel(GAMMA_ITU709, *r["rgb_in"])) == r["ycbcr"], r["test"]
            assert list(oracle.decode_pixel(GAMMA_ITU709, *r["ycbcr"])) == r["rgb_out"], r["test"]
        else:
            continue
        seen += 1
    assert seen >= 9


# ------------------------------------------------------ exhaustive reference pins

@pytest.mark.parametrize("gamma", [GAMMA_APPLE, GAMMA_SRGB, GAMMA_ITU709])
def test_roundtrip_histograms(oracle, vectors, refdata, gamma):
    """Exhaustive 2^24 encode->decode histograms asserted / recorded by the reference
    (CoreImageMetalFilterTests.m:420-553, 557-691, 696-831)."""
    name = GAMMA_NAMES[gamma]
    hist = oracle.roundtrip_histogram(gamma)
    pinned = vectors["histograms"][name]
    assert hist[0] == pinned["asserted_exact"]
    keys = ["exact"] + ["off%d" % i for i in range(1, 10)] + ["offMore9"]
    for i, k in enumerate(keys):
        if k in pinned["comment_block"]:
            assert hist[i] == pinned["comment_block"][k], (name, k)
    assert hist == refdata["histograms"][name]
    assert sum(hist) == 1 << 24


@pytest.mark.parametrize("gamma", GAMMAS)
def test_full_decode_table_hash(oracle, refdata, gamma):
    """sha256 of the whole (Y,Cb,Cr)->(R,G,B) table against the reference headers' own."""
    table = oracle.decode_table(gamma)
    assert hashlib.sha256(table.tobytes()).hexdigest() == refdata["table_sha256"][GAMMA_NAMES[gamma]]


@pytest.mark.parametrize("gamma", GAMMAS)
def test_threshold_tables_match_reference(oracle, refdata, gamma):
    thr = oracle.thresholds(gamma)
    want = np.array([int(h, 16) for h in refdata["thresholds_hex"][GAMMA_NAMES[gamma]]], dtype=np.uint32)
    assert np.array_equal(thr.view(np.uint32), want)


@pytest.mark.parametrize("gamma", GAMMAS)
def test_composite_is_a_threshold_function(oracle, gamma):
    """Every float in [0,1]: the byte map is monotone and equals the 255-entry threshold
    table -- the property the GPU kernel's exact table lookup rests on."""
    assert oracle.check_thresholds(gamma) == 0


def test_alpha_map(oracle, refdata):
    assert [oracle.decode_alpha(a) for a in range(256)] == refdata["alpha_map"]
    assert oracle.decode_alpha(16) == 0 and oracle.decode_alpha(235) == 255


def test_subsample_blocks(oracle, refdata):
    """BT709_average_pixel_values (BT709.h:1349-1509) on sampled 2x2 blocks."""
    for b in refdata["subsample_blocks"]:
        assert list(oracle.subsample_block(b["rgb"], b["in"], b["out"])) == b["y4cbcr"]


def test_xctest_average_of_4_vectors(oracle, vectors):
    """The five 2x2-averaging expectations the reference's own XCTest file asserts
    (CoreImageMetalFilterTests.m:1683-2096: BT709_average_pixel_values with sRGB input and sRGB /
    Apple / linear output): Y1..Y4 and the averaged Cb, Cr."""
    assert len(vectors["average_blocks"]) == 5
    for b in vectors["average_blocks"]:
        assert list(oracle.subsample_block(b["rgb"], b["in"], b["out"])) == b["y4cbcr"], b["test"]


def test_bundled_pattern_crops(oracle, refdata, patterns):
    """Crops of the reference's bundled test images, NV12-encoded and decoded by the
    reference headers; the oracle must reproduce every output byte in every gamma."""
    assert len(refdata["patterns"]) >= 7
    for rec in refdata["patterns"]:
        y, uv = patterns[rec["tag"] + "_y"], patterns[rec["tag"] + "_uv"]
        for gamma in GAMMAS:
            out = oracle.decode_nv12(gamma, y, uv)
            assert hashlib.sha256(out.tobytes()).hexdigest() == rec["bgra_sha256"][GAMMA_NAMES[gamma]]
        assert np.array_equal(oracle.decode_nv12(GAMMA_APPLE, y, uv), patterns[rec["tag"] + "_bgra_apple"])


def test_whole_bundled_patterns(oracle, full_patterns):
    """north_star: "bit-exact on the bundled test patterns" -- the WHOLE 1920x1080 QuickTime test pattern (both
    renditions; BASELINE config 1's frame on the CPU path), the 512x512 Image.tga and (round 4) the whole 2048x1536
    clouds photograph -- every image SURVEY 8(c) lists, none cropped -- encoded to NV12 and decoded
    by the reference headers in every gamma, and through decode + exact 2:1 pass 2: the oracle reproduces every
    byte (hashes: tests/golden/patterns_full.json)."""
    images, planes = full_patterns
    assert {tuple(r["image_size"]) for r in images} == {(1920, 1080), (512, 512), (2048, 1536)} and len(images) == 4
    for rec in images:
        y, uv = planes[rec["tag"] + "_y"], planes[rec["tag"] + "_uv"]
        assert y.shape == (rec["image_size"][1], rec["image_size"][0])
        assert hashlib.sha256(y.tobytes() + uv.tobytes()).hexdigest() == rec["nv12_sha256"]
        for gamma in GAMMAS:
            out = oracle.decode_nv12(gamma, y, uv)
            assert hashlib.sha256(out.tobytes()).hexdigest() == rec["bgra_sha256"][GAMMA_NAMES[gamma]], rec["tag"]
        half = oracle.decode_nv12_half(GAMMA_APPLE, y, uv)
        assert hashlib.sha256(half.tobytes()).hexdigest() == rec["half_sha256"]["apple"], rec["tag"]


# ------------------------------------------------------------ live reference (here)

def test_live_reference_pixels(oracle, reference):
    rng = np.random.default_rng(1)
    for gamma in GAMMAS:
        for Y, Cb, Cr in rng.integers(0, 256, (2000, 3)):
            assert oracle.decode_pixel(gamma, int(Y), int(Cb), int(Cr)) == \
                reference.decode_pixel(gamma, int(Y), int(Cb), int(Cr))
        for R, G, B in rng.integers(0, 256, (2000, 3)):
            assert oracle.encode_pixel(gamma, int(R), int(G), int(B)) == \
                reference.encode_pixel(gamma, int(R), int(G), int(B))


def test_live_reference_encode_frame(oracle, reference):
    rng = np.random.default_rng(2)
    bgra = rng.integers(0, 1 << 24, 32 * 16, dtype=np.uint32)
    for ig, og in ((1, 0), (1, 1), (0, 0), (2, 2)):
        y0, uv0 = reference.encode_nv12(bgra, 32, 16, ig, og)
        y1, uv1 = oracle.encode_nv12(bgra, 32, 16, ig, og)
        assert np.array_equal(y0, y1) and np.array_equal(uv0, uv1)


# ------------------------------------------------------------------ frame semantics

def test_odd_dimensions_rejected(oracle):
    """unconvert rejects odd width/height (BGRAToBT709Converter.m:69-74)."""
    assert oracle.unconvert_packed(GAMMA_APPLE, np.zeros(6, np.uint32), 3, 2) is None
    assert oracle.unconvert_packed(GAMMA_APPLE, np.zeros(6, np.uint32), 2, 3) is None
    assert oracle.decode_nv12(GAMMA_APPLE, np.zeros((2, 3), np.uint8), np.zeros((1, 4), np.uint8)) is None


def test_chroma_is_replicated_not_interpolated(oracle):
    """Every pixel of a 2x2 block sees the block's single CbCr sample
    (BGRAToBT709Converter.m:267-277; AAPLShaders.metal:350)."""
    y = np.full((4, 4), 120, np.uint8)
    uv = np.array([[60, 200, 200, 60], [128, 128, 90, 170]], np.uint8)
    out = oracle.decode_nv12(GAMMA_APPLE, y, uv).reshape(4, 4, 4)
    for by in range(2):
        for bx in range(2):
            blk = out[2 * by:2 * by + 2, 2 * bx:2 * bx + 2].reshape(4, 4)
            assert (blk == blk[0]).all()
            R, G, B = oracle.decode_pixel(GAMMA_APPLE, 120, int(uv[by, 2 * bx]), int(uv[by, 2 * bx + 1]))
            assert list(blk[0]) == [B, G, R, 0xFF]


def test_packed_to_nv12_keeps_odd_row_chroma(oracle):
    """copyBT709ToCoreVideo writes CbCr from every row into row/2, so the odd row wins
    (BGRAToBT709Converter.m:1063-1088)."""
    packed = np.array([(10 << 16) | (20 << 8) | 1, (11 << 16) | (21 << 8) | 2,
                       (30 << 16) | (40 << 8) | 3, (31 << 16) | (41 << 8) | 4], np.uint32)
    y, uv = oracle.packed_to_nv12(packed, 2, 2)
    assert y.tolist() == [[1, 2], [3, 4]]
    assert uv.tolist() == [[40, 30]]


def test_strided_planes(oracle):
    rng = np.random.default_rng(3)
    w, h = 18, 6
    y = rng.integers(0, 256, (h, 32), dtype=np.uint8)
    uv = rng.integers(0, 256, (h // 2, 48), dtype=np.uint8)
    a = oracle.decode_nv12(GAMMA_SRGB, (y, w), (uv, w))
    b = oracle.decode_nv12(GAMMA_SRGB, np.ascontiguousarray(y[:, :w]), np.ascontiguousarray(uv[:, :w]))
    assert np.array_equal(a, b)


def test_half_scale_definition(oracle):
    """Fused 2:1 downscale restatement (parity unpinned by the reference): a flat frame
    scales to the same colour; output is the linear-light mean of the four decoded bytes."""
    y = np.full((8, 8), 180, np.uint8)
    uv = np.full((4, 8), 128, np.uint8)
    full = oracle.decode_nv12(GAMMA_APPLE, y, uv).reshape(8, 8, 4)
    half = oracle.decode_nv12_half(GAMMA_APPLE, y, uv).reshape(4, 4, 4)
    assert (half == full[0, 0]).all()
    # black/white checker inside each 2x2 block -> linear mean 0.5 -> sRGB 188
    y2 = np.tile(np.array([[16, 235], [235, 16]], np.uint8), (4, 4))
    half2 = oracle.decode_nv12_half(GAMMA_APPLE, y2, uv).reshape(4, 4, 4)
    assert (half2[..., :3] == 188).all()


# ------------------------------------------------------------------ pass 2 pinned to the reference's own arithmetic

def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_pass2_oracle_equals_reference_composed_fixture(oracle, pass2, patterns):
    """The oracle's fused rescale (2:1 and any ratio, with and without an alpha plane) against
    tests/golden/pass2.json = outputs of oracle/ref_harness.c's composition of the REFERENCE's
    inlines (pass 1 bytes, sRGB_nonLinearNormToLinear(byteNorm(b)), the float sum, sRGB_linearNormToNonLinear,
    (int)round(v*255.0f)).  This is what pins SURVEY 8(a) row 9 to reference-executed arithmetic."""
    import pass2_cases as pc
    small = np.load(os.path.join(os.path.dirname(__file__), "golden", "pass2_small.npz"))
    for kind, gamma, src, dst, seed, with_alpha in pc.CASES:
        y, c, a = pc.seeded_frame(src, seed, with_alpha)
        got = (oracle.decode_nv12_half(gamma, y, c, alpha=a) if kind == "half"
               else oracle.decode_nv12_scaled(gamma, y, c, dst[0], dst[1], alpha=a))
        key = pc.case_key(kind, gamma, src, dst, seed, with_alpha)
        assert _sha(got) == pass2["cases"][key], key
        if key.replace("/", "_") in small:
            assert np.array_equal(got, small[key.replace("/", "_")]), key
    assert len(pass2["patterns"]) == 7
    for tag, rec in pass2["patterns"].items():
        y, c = patterns[tag + "_y"], patterns[tag + "_uv"]
        for gamma in range(4):
            assert _sha(oracle.decode_nv12_half(gamma, y, c)) == rec["half/g%d" % gamma], (tag, gamma)
            for (ow, oh) in pc.PATTERN_SIZES["scaled"]:
                assert _sha(oracle.decode_nv12_scaled(gamma, y, c, ow, oh)) == rec["scaled/g%d/%dx%d" % (gamma, ow, oh)]


def test_pass2_oracle_equals_reference_live(oracle, reference):
    """Container only: the same comparison against the reference library itself on fresh frames."""
    rng = np.random.default_rng(77)
    for case in range(12):
        w, h = 4 * int(rng.integers(1, 40)), 4 * int(rng.integers(1, 12))
        gamma = int(rng.integers(0, 4))
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        c = rng.integers(0, 256, (h // 2, w), dtype=np.uint8)
        a = rng.integers(0, 256, (h, w), dtype=np.uint8) if case % 3 == 0 else None
        assert np.array_equal(oracle.decode_nv12_half(gamma, y, c, alpha=a), reference.decode_nv12_half(gamma, y, c, alpha=a))
        ow, oh = int(rng.integers(1, 2 * w)), int(rng.integers(1, 2 * h))
        assert np.array_equal(oracle.decode_nv12_scaled(gamma, y, c, ow, oh, alpha=a),
                              reference.decode_nv12_scaled(gamma, y, c, ow, oh, alpha=a))
