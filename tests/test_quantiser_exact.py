"""Exactness of the product's quantiser and alpha arithmetic, by exhaustion.

`metalbt709decoder_amd/csrc/bt709_quantise.h` is compiled twice: by hipcc into the kernels and by g++
into tests/native/quantise_sweep.cpp, which replays it against the oracle's
`(int)round(x * 255.0f)` (Renderer/BT709.h:881-883, Renderer/sRGB.h:32-36):

  * quantise_exact: every float in [0, 1] (1 065 353 217 of them) -- what alpha_word_of applies to a
    bilinear-weighted alpha and to the alpha of an RGBA16Float intermediate;
  * quantise_enumerated (the three-instruction form of the 1:1 kernels): differs at exactly ONE float,
    0x3b008080 (round 2's review), which is why it is only used on enumerated argument sets;
  * the 256 codes of an alpha sample and all 256^4 ordered tap tuples of the exact 2:1 alpha filter.

The -m gpu case drives pass 2 with a crafted RGBA16Float intermediate whose filtered alpha is that float.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FLAW_BITS = 0x3B008080  # x * 255.0f = 0x1.fffffep-2: x * 255.0f + 0.5f rounds up to 1.0f


@pytest.fixture(scope="module")
def sweep(tmp_path_factory, oracle):
    out = str(tmp_path_factory.mktemp("native") / "libquantise_sweep.so")
    odir = os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-Wall", "-Werror",
           "-I", os.path.join(ROOT, "metalbt709decoder_amd", "csrc"), "-I", odir,
           os.path.join(HERE, "native", "quantise_sweep.cpp"), "-o", out, "-L", odir, "-loracle",
           "-Wl,-rpath," + odir, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(out)
    lib.sweep_half_alpha_tuples.restype = C.c_uint64
    return lib


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def test_quantise_exact_equals_reference_rounding_for_every_float_in_unit_range(sweep):
    out = (C.c_uint64 * 3)()
    sweep.sweep_unit_floats(0, 0x3F800000, _threads(), out)
    assert out[0] == 0, "quantise_exact differs from (int)round(x * 255.0f) on %d floats" % out[0]
    # the documented flaw of the short form: one float, and it is the one the review named
    assert out[1] == 1 and out[2] == FLAW_BITS


def test_alpha_sample_codes(sweep, oracle):
    norm = (C.c_float * 256)()
    assert sweep.sweep_alpha_samples(norm) == 0
    want = np.array([oracle.lib.bt709o_decode_alpha(a) for a in range(256)], np.float32) * np.float32(1.0 / 255.0)
    assert np.array_equal(np.frombuffer(norm, np.float32), want)


def test_half_alpha_filter_all_ordered_tuples(sweep):
    """256^4 ordered tuples of tap values byte * (1/255f): the product's sum * 63.75f + 0.5f, truncated, is the
    oracle's round(255 * ((((a+b)+c)+d) * 0.25f)) on every one of them."""
    assert sweep.sweep_half_alpha_tuples(_threads()) == 0


def crafted_rgba16f_intermediate():
    """A 4 x 2 RGBA16Float intermediate, shown enlarged 2x horizontally: output column 2 samples texels 0 and 1
    with weights 0.25 / 0.75 (sx = 2 * 0.5 + 0.25 - 0.5 = 0.75), rows map 1:1 (fy = 0).  With
    alpha(texel 0) = 1026 * 2^-24 and alpha(texel 1) = 1360 * 2^-19 (both normal halves) the filtered alpha is
    0.25 a0 + 0.75 a1 = 2^-9 + 2^-17 + 2^-25 = 0x3b008080 exactly (every product and the sum are exact)."""
    a0, a1 = np.float16(1026 * 2.0 ** -24), np.float16(1360 * 2.0 ** -19)
    assert float(a0) == 1026 * 2.0 ** -24 and float(a1) == 1360 * 2.0 ** -19
    acc = np.float32(0.25) * np.float32(a0) + np.float32(0.75) * np.float32(a1)
    assert acc.view(np.uint32) == FLAW_BITS
    src = np.zeros((2, 4, 4), np.float16)
    src[..., :3] = np.float16(0.5)
    src[:, 0, 3], src[:, 1, 3], src[:, 2, 3], src[:, 3, 3] = a0, a1, np.float16(1.0), np.float16(0.25)
    return src


def test_oracle_on_the_crafted_intermediate(oracle):
    out = oracle.render_scaled(crafted_rgba16f_intermediate(), 8, 2).reshape(2, 8, 4)
    assert (out[:, 2, 3] == 0).all()  # round(0.49999997) = 0; trunc(v + 0.5f) would say 1


@pytest.mark.gpu
def test_gpu_pass2_alpha_at_the_flaw_float(oracle):
    """render_scaled's alpha goes through quantise_exact: 0, as the reference's (int)round gives."""
    import gpu_helpers as gh
    import metalbt709decoder_amd as mb
    ctx = gh.context()
    src = crafted_rgba16f_intermediate()
    scale = mb.MetalScaleRenderContext()
    assert scale.setupRenderPipelines(ctx)
    inter = ctx.makeBGRATexture((4, 2), pixels=src, pixelFormat=mb.MTLPixelFormatRGBA16Float)
    view = ctx.makeBGRATexture((8, 2))
    assert scale.renderScaled(ctx, view, 8, 2, None, None, inter, True), scale.lastStatus
    got = ctx.getBGRATexturePixels(view).view(np.uint8).reshape(2, 8, 4)
    assert np.array_equal(got, oracle.render_scaled(src, 8, 2).reshape(2, 8, 4))
    assert (got[:, 2, 3] == 0).all()


@pytest.mark.gpu
def test_gpu_scaled_alpha_sweep_of_weights(oracle):
    """The fused any-ratio kernel's filtered alpha (weights from many geometries, all 256 sample codes) against
    the oracle: the quantiser under test is the exact one, the weights are what varies."""
    import gpu_helpers as gh
    import metalbt709decoder_amd as mb
    rng = np.random.default_rng(709)
    for (w, h), (ow, oh) in (((64, 32), (51, 23)), ((48, 16), (255, 37)), ((32, 32), (33, 31)), ((256, 8), (255, 9))):
        y, c = gh.random_nv12(w, h, seed=w + ow)
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        a[:, : min(w, 256)] = np.arange(min(w, 256), dtype=np.uint8)[None, :] * (256 // min(w, 256))
        a[1::2] = 255 - a[1::2]
        got = gh.gpu_decode_scaled(y, c, (ow, oh), mb.MetalBT709GammaSRGB, alpha=a)
        assert np.array_equal(got, oracle.decode_nv12_scaled(1, y, c, ow, oh, alpha=a)), ((w, h), (ow, oh))


def test_quantiser_header_forbids_contraction_by_itself(tmp_path):
    """bt709_quantise.h carries its own `fp contract(off)` pragma: compiled WITHOUT build.py's -ffp-contract=off (hipcc's
    default is fp-contract=fast; an integrator's own build of csrc/) the quantisers still are a multiply and an add, never one
    fused multiply-add -- the rounding the sweeps above certified (round 3's advisor finding)."""
    src = tmp_path / "q.hip"
    src.write_text('#include <hip/hip_runtime.h>\n#include "bt709_quantise.h"\n'
                   "__global__ void k(const float *x, uint32_t *o) {\n"
                   "  o[threadIdx.x] = bt709::quantise_enumerated(x[threadIdx.x]) + bt709::quantise_exact(x[threadIdx.x + 64]) +\n"
                   "                   bt709::half_alpha_sum_to_byte(x[0], x[1], x[2], x[3]) + (uint32_t)bt709::alpha_norm_of_unit(x[4]);\n}\n")
    out = tmp_path / "q.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only",
                        "-I", os.path.join(ROOT, "metalbt709decoder_amd", "csrc"), str(src), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    asm = out.read_text()
    body = asm[asm.index("_Z1kPKfPj:"):asm.index(".Lfunc_end0")]
    import re
    assert not re.search(r"\bv_(fma|fmac|mad|pk_fma)_f32", body), "the quantiser was contracted into an fma"
    assert len(re.findall(r"\bv_mul_f32", body)) >= 4
    # the host twin, with FMA instructions available and GCC's default -ffp-contract=fast
    hsrc = tmp_path / "q.cpp"
    hsrc.write_text('#include "bt709_quantise.h"\nunsigned f(float x, float a, float b, float c, float d) {\n'
                    "  return bt709::quantise_enumerated(x) + bt709::half_alpha_sum_to_byte(a, b, c, d); }\n")
    hobj = tmp_path / "q.o"
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-mfma", "-Wall", "-Werror", "-c", "-I",
                        os.path.join(ROOT, "metalbt709decoder_amd", "csrc"), str(hsrc), "-o", str(hobj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dis = subprocess.run(["objdump", "-d", str(hobj)], capture_output=True, text=True).stdout
    assert "vfmadd" not in dis and "vmulss" in dis
