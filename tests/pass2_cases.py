"""Inputs of the pass-2 golden fixture (tests/golden/pass2.json), shared by the generator
(tests/golden/make_golden.py, container only) and the tests that check the oracle and the GPU against
it.  Frames are seeded PRNG bytes (numpy PCG64: the same bytes everywhere) or crops of the bundled
test patterns already stored in tests/golden/patterns.npz."""
import numpy as np

# (kind, gamma, (W, H), (OW, OH), seed, with_alpha)
CASES = [("half", g, (w, h), (w // 2, h // 2), 100 + 7 * g + w, False)
         for g in range(4) for (w, h) in ((8, 4), (64, 32), (256, 64), (1920, 64))]
CASES += [("half", 1, (64, 32), (32, 16), 901, True), ("half", 1, (256, 64), (128, 32), 902, True)]
CASES += [("scaled", g, src, dst, 300 + 11 * g + dst[0], False)
          for g in range(4) for (src, dst) in (((64, 32), (40, 20)), ((64, 32), (17, 9)), ((30, 18), (64, 40)),
                                               ((1920, 64), (1280, 43)), ((50, 22), (1, 1)))]
CASES += [("scaled", 1, (64, 32), (40, 20), 903, True), ("scaled", 1, (30, 18), (64, 40), 904, True)]
# bundled test patterns (crops in patterns.npz, encoded by the reference's encoder): tag -> output sizes
PATTERN_SIZES = {"half": None, "scaled": [(100, 60), (37, 91)]}


def seeded_frame(size, seed, with_alpha=False):
    """(y, cbcr, alpha or None): uniform bytes, full range (exercises saturation)."""
    w, h = size
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c = rng.integers(0, 256, (h // 2, w), dtype=np.uint8)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8) if with_alpha else None
    return y, c, a


def case_key(kind, gamma, src, dst, seed, with_alpha):
    return "%s/g%d/%dx%d-%dx%d/s%d%s" % (kind, gamma, src[0], src[1], dst[0], dst[1], seed, "/alpha" if with_alpha else "")
