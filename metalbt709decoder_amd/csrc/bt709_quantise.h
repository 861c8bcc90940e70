// (int)round(x * 255.0f) -- the reference's last step (Renderer/BT709.h:881-883, `byteNorm`'s inverse of
// Renderer/sRGB.h:32-36) -- and the alpha arithmetic built on it, written ONCE for the device and for the host:
// hipcc compiles these functions into the kernels, and tests/native/quantise_sweep.cpp compiles the very same
// text with g++ (-ffp-contract=off on both sides: every operator below is one IEEE binary32 operation) to replay
// them over every input they can meet, against the oracle.  Nothing here is a table.
//
// Two quantisers, because only one of them is right for every float:
//
//   quantise_exact(x)        t = trunc(v), byte = t + (v - t >= 0.5) with v = x * 255.0f.  v - t is exact (t and v
//                            share a binade or t = 0), so this is round-half-away-from-zero for EVERY float
//                            x in [0, 1]: proven by the sweep over all 1 065 353 217 of them.  5 VALU instructions.
//                            Used wherever the argument is not enumerable: a bilinear-weighted alpha
//                            (decode_nv12_scaled), the alpha of an RGBA16Float intermediate (render_scaled).
//   quantise_enumerated(x)   trunc(v + 0.5f): 3 VALU instructions.  v + 0.5f is NOT always exact: for
//                            v = 0x1.fffffep-2 (x = 0x3b008080, the one float in [0, 1] where it matters) the sum
//                            rounds up to 1.0f and the byte comes out 1 where the reference gives 0.  It is used
//                            only where the set of possible arguments is finite and swept by a test: the 2^24
//                            (Y, Cb, Cr) triples and 256 alpha codes of the 1:1 kernels (tests/test_gpu_parity.py
//                            exhaustive sweeps), and the 256^4 ordered alpha tuples of the exact 2:1 filter
//                            (tests/test_quantiser_exact.py replays half_alpha_sum_to_byte over all of them).
#pragma once

#include <cstdint>

// Every `a * b + c` below is TWO roundings, which is what the exhaustive sweeps certified.  build.py compiles with
// -ffp-contract=off, but hipcc's own default is fp-contract=fast: an integrator's build of csrc/ without that flag would
// fuse them into one v_fma_f32 and change the rounding.  The header says it itself, for the rest of the including file.
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#if defined(__HIPCC__)
#define BT709_HD __host__ __device__ __forceinline__
#else
#define BT709_HD inline
#endif

namespace bt709 {

BT709_HD uint32_t quantise_exact(float x) {
  const float v = x * 255.0f;              // BT709.h:881: a float multiply
  const float t = __builtin_truncf(v);     // v_trunc_f32
  const float d = v - t;                   // exact
  return static_cast<uint32_t>(t) + (d >= 0.5f ? 1u : 0u);
}

BT709_HD uint32_t quantise_enumerated(float x) {
  return static_cast<uint32_t>(x * 255.0f + 0.5f);  // v_cvt_u32_f32 truncates
}

// byteNorm(decoded alpha byte) of an alpha sample's saturated luma term x: what the sampler of pass 2 reads from
// the 8-bit intermediate's alpha channel (a plain unorm; AAPLShaders.metal:411-438 writes it).  x takes 256 values.
BT709_HD float alpha_norm_of_unit(float x) {
  return __builtin_truncf(x * 255.0f + 0.5f) * (1.0f / 255.0f);
}

// the exact 2:1 filter of four such taps, (((a + b) + c) + d) * 0.25f, written back as round(255 v): the quarter
// is exact, so (s * 0.25f) * 255.0f is one rounding of s * 63.75f
BT709_HD uint32_t half_alpha_sum_to_byte(float a, float b, float c, float d) {
  const float s = ((a + b) + c) + d;
  return static_cast<uint32_t>(s * 63.75f + 0.5f);
}

}  // namespace bt709
