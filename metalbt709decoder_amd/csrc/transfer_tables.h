// Host-side builder of the exact per-gamma transfer tables the kernels look up.
//
// The reference applies, per channel, after matrix+saturate:
//     video curve -> linear      (float in, libm double pow inside)
//     linear      -> sRGB        (same)
//     (int)round(v * 255.0f)
// (Renderer/BT709.h:856-883 for the default Apple mode).  No GPU pow reproduces
// libm's correctly rounded double pow bit for bit, and none is needed: each such
// composite is a monotone step function of the float in [0,1] (checked for all
// 1 065 353 217 inputs by tests/test_oracle_golden.py), so it is fully described
// by 255 thresholds.  The kernel wants O(1) lookup, so the thresholds are laid
// out in N + 1 uniform buckets, N a power of two chosen so no bucket holds two
// thresholds:
//     byte(x) = bucket[q].base + (x >= bucket[q].edge),   q = bucket_index(x)
// The kernels saturate x to [0,1] first (as the reference does, BT709.h:444-446) and compare in
// x units; `edge` is the threshold itself.  The bucket index costs ONE float add: with
// M = 2^23 / N a float in [M, 2M) has ulp 1/N, so
//     bits(x + M) = bits(M) + round(x N)          (round to nearest even, the default mode)
// and bucket q is the set of x whose sum lands on M + q / N: buckets are 1/N wide and centred on
// q / N (q = 0 and q = N are half buckets).  Any monotone index function works as long as the
// builder below and the kernels use the same one; round-to-nearest needs no MODE register switch
// (round 1 used floor(x N) through a round-toward-zero add between two s_setreg; tools/lab_variants.py index_rtz
// rebuilds that form for A/B runs).
#pragma once

#include <cstdint>
#include <vector>

namespace bt709 {

enum Gamma : int { kGammaApple = 0, kGammaSRGB = 1, kGammaLinear = 2, kGammaITU709 = 3, kGammaCount = 4 };

// Table kinds build_transfer_table understands: the four decode composites above, plus the one
// extra composite the ENCODER needs: byte = round(255 * Apple196_linearNormToNonLinear(v))
// (BT709_from_linear(v, BT709GammaApple), Renderer/BT709.h:1150-1167).  The other two
// BT709_from_linear flavours coincide with decode composites: Srgb == kGammaLinear's,
// Linear (plain quantise) == kGammaSRGB's.
constexpr int kTableEncodeApple = 4;
constexpr int kTableKinds = 5;

// One bucket.  `edge` is the single threshold strictly inside the bucket, or +inf.
struct alignas(8) TransferBucket {
  float edge;
  uint32_t base;
};

// Bucket for the fused rescale kernels: the decoded byte is never materialised, the lookup returns
// it already linearised (sampler-side sRGB decode of an sRGB8 texel:
// sRGB_nonLinearNormToLinear(byteNorm(b)), Renderer/sRGB.h:32-57), and the choice between the two
// values of a bucket is ONE v_sub_f32 + ONE v_med3_f32 instead of compare + select:
//     d   = x - edge_pred                 edge_pred = the float just below the threshold, so
//                                         d > 0  <=>  x >= threshold (d == 0 when x == edge_pred)
//     lin = med3(lin_below, lin_above, d)
// which is exact because both linear values are stored times 2^-40: 0 <= lin_below <= lin_above
// <= 2^-40, while a positive d is at least one ulp of a threshold (>= 2^-36).  The kernels fold
// 2^40 into the power-of-two scale they apply after averaging; scaling by a power of two commutes
// with every rounding on the way (smallest product in the bilinear kernel ~2^-98: no underflow).
constexpr int kLinearScaleLog2 = -40;
struct alignas(16) TransferBucketLinear {
  float edge_pred;  // nextafter(threshold, -inf), or +inf when the bucket holds no threshold
  float lin_below;  // 2^-40 * linear value of byte `base`
  float lin_above;  // 2^-40 * linear value of byte `base + 1`
  uint32_t base;
};

struct TransferTable {
  int gamma = 0;
  uint32_t n = 0;          // bucket count N (power of two)
  float thresholds[255];   // t[k-1] = min { x in [0,1] : byte(x) >= k }
  // buckets 0..N (bucket N: x == 1.0), padded to a 16-byte multiple
  std::vector<TransferBucket> buckets_unit;
  std::vector<TransferBucketLinear> buckets_linear;  // the same N + 1 buckets, linearised outputs
  // LOG-bucket form of buckets_unit, for the 1:1 kernels, where it is at most half the size (round 5: the LINEAR mode, whose
  // thresholds crowd near zero -- sRGB's slope 12.92 -- and force N = 4096 uniform buckets, 33 KiB per workgroup):
  //     q = (bits(x + log_add) >> 16) - log_first        log_first = bits(log_add) >> 16
  // i.e. the floats x + log_add that share an exponent and 7 mantissa bits: buckets (x + log_add) / 128 wide, growing with x as
  // the thresholds' spacing does.  One more VALU instruction per channel than the uniform form (the shift); 645 buckets = 5 KiB
  // for the LINEAR mode.  log_add == 0: not built (the uniform form is small already).
  float log_add = 0.0f;
  uint32_t log_first = 0;
  std::vector<TransferBucket> buckets_log;  // padded to a 16-byte multiple
};
uint32_t bucket_index_log(float x, float log_add);  // bits(x + log_add) >> 16, the add as the kernels do it

// q of the comment above, computed exactly as the kernels do (csrc/bt709_device.h magic_index).
uint32_t bucket_index(float x, float magic);

// Scalar transfer functions, float in / float out, C semantics of the reference.
float srgb_to_linear(float v);      // Renderer/sRGB.h:43-57
float linear_to_srgb(float v);      // Renderer/sRGB.h:62-74
float itu709_to_linear(float v);    // Renderer/BT709.h:68-81
float apple196_to_linear(float v);  // Renderer/BT709.h:125-137
float linear_to_apple196(float v);  // Renderer/BT709.h:139-151
int quantize_byte(float v);         // (int)round(v * 255.0f), Renderer/BT709.h:881-883

// Per-channel composite of one gamma mode.
int transfer_to_byte(int gamma, float v);

// Encoder tables for one (input gamma, output gamma) pair of cvpbu_ycbcr_subsample /
// BT709_average_pixel_values (Renderer/CVPixelBufferUtils.h:241-399, Renderer/BT709.h:1349-1509);
// gammas use this file's ids (kGammaApple / kGammaSRGB / kGammaLinear).
struct alignas(8) EncodeByteEntry {
  float lin;       // BT709_tolinearNorm of the byte for the input gamma (BT709.h:1100-1146)
  float enc_norm;  // byteNorm(BT709_from_linear(lin, output gamma)): the kernel multiplies by Kr / Kg / Kb (BT709.h:222)
};
struct EncodeTables {
  EncodeByteEntry per_byte[256];  // one 2 KiB table for the three channels
  int from_linear_kind = 0;       // table kind whose buckets implement BT709_from_linear(., output gamma)
};
bool build_encode_tables(int in_gamma, int out_gamma, EncodeTables *out);

// Two-resolution bucket table over x in [0,1] for composites whose thresholds crowd near zero
// (the encoder's BT709_from_linear tables need N = 4096 uniformly, 33 KiB, only because of
// their first 1/16 of the range).  Used by the ENCODER (bt709_encode.hip); the rescale kernels' encode side moved to the
// log-bucket form above in round 5 (fewer instructions per lookup).  With xs = n_fine * x:
//     q = xs < split ? (uint)xs : (uint)(xs * coarse_scale) + coarse_offset
//       = min((uint)xs, ((uint)xs >> log2(ratio)) + coarse_offset)   // the two index functions cross at
//                                                                    // the split, the fine one grows faster
//     byte = buckets[q].base + (xs >= buckets[q].edge)          // edges stored times n_fine
// i.e. buckets of width 1/n_fine below split/n_fine and ratio times wider above.
struct SplitTable {
  uint32_t n_fine = 0;
  float split = 0.0f;          // in fine buckets (a multiple of ratio)
  float coarse_scale = 1.0f;   // 1 / ratio (power of two)
  uint32_t coarse_offset = 0;  // split - split / ratio
  std::vector<TransferBucket> buckets;  // padded to a 16-byte multiple
};
// Returns false if no (n_fine, split, ratio) among the candidates leaves at most one threshold
// per bucket.
bool build_split_table(int kind, SplitTable *out);

// Builds thresholds + buckets.  Returns false if gamma is unknown or the
// single-threshold-per-bucket property cannot be met with N <= 65536.
bool build_transfer_table(int gamma, TransferTable *out);

// Uniform bucket table with ANY bucket count n (not a power of two): the smallest n that still keeps
// one threshold per bucket.  The sRGB-encode composite needs n >= 3296 only because its thresholds
// are 1/(255 * 12.92) apart near zero; a power of two would cost 4096 buckets = 33 KiB, n = 33xx costs
// 26 KiB, and the index is ONE fma instead of the two-resolution table's convert + shift + add + min:
//     q = bits(fma(v, n, 2^23)) - bits(2^23) = round(v n);  byte = buckets[q].base + (v >= buckets[q].edge)
// (edge is the threshold itself, in v units: the rounding only moves bucket boundaries, which the builder
// replays exactly, never the comparison).  n + 2 buckets (q <= n, one spare), padded to 16 bytes.
struct UniformTable {
  uint32_t n = 0;
  std::vector<TransferBucket> buckets;
};
uint32_t uniform_index(float v, float n);  // q of the comment above, computed exactly as the kernels do
bool build_uniform_table(int kind, uint32_t n_min, UniformTable *out);

// ---- RGBA16Float render targets (Renderer/AAPLRenderer.m:143-170: where sRGB texture writes are
// unavailable the reference renders pass 1 into RGBA16Float, i.e. the shader's LINEAR-light float
// goes to the texture as an IEEE binary16, round to nearest even).  Per channel that is
//     H(x) = half(curve_to_linear(x))        curve = Apple 1.961 / sRGB / none / ITU (BT709.h:68-137, sRGB.h:43-57)
// a monotone step function of the saturated x with up to 15 360 steps: too many for a bucket table
// that must sit in LDS beside resident workgroups.  Below the curve's split point the reference
// multiplies by an exact constant and the hardware conversion gives H directly; above it the
// kernel takes a CANDIDATE h0 that is H or H - 1 and settles it against the threshold above it:
//     H = h0 + (x >= T[h0 + 1]),   T[h] = smallest float x with H(x) >= h
// T is indexed by the OUTPUT code, so it holds exactly one entry per step (8-9 k entries, ~34 KiB).
// The candidate: the TANGENT of the curve at the start of x's bucket (rounds 2-3 took it from v_log_f32 / v_exp_f32, two
// quarter-rate instructions per channel; round 4 introduced the tangents), {intercept, slope}:
//     p = slope * x + intercept,   ONE fma
// The curves are convex powers, so the tangent lies BELOW the curve, by at most g(g-1)/2 * 2^-14 = 1.0e-4 of the value
// (g <= 2.4) at the far end of a bucket -- a fifth of a half's spacing (2^-11 of the value) -- and intercept and slope carry a
// further factor 1 - 2^-19 and are rounded toward a smaller p, so that neither their rounding, the fma's, nor the reference's
// own float steps can lift p above the true value: half(p) is H or H - 1, never H + 1.
// Buckets (round 5, second form): the index comes from ONE v_cvt_pkrtz_f16_f32 over a PAIR of channels --
//     u = x * index_scale (binary32, round to nearest);  hb = max(half_rtz(u), kHalfCandFloor);  bucket = hb >> 3
// -- i.e. the values u sharing a binary16 exponent and 7 mantissa bits.  index_scale is chosen per curve so that the curve's
// SPLIT POINT maps exactly onto the bucket boundary u = 2^-4 (kHalfCandEdge): x >= split  <=>  bucket >= 1.  Bucket 0 --
// everything below the split, held there by the max -- has the entry {0, low_scale}: p = fma(x, low_scale, +0) IS the
// reference's exact product x * low_scale of the piece below the split.  So the kernel needs neither the comparison with the
// split, nor the select between product and tangent, nor the product itself (rounds 2-5a: three more instructions per
// channel), and two channels share the conversion, the max and -- after the fma -- the v_cvt_pk_f16_f32 and the saturating
// subtraction that indexes T.  Rounds 4-5a indexed by bits(x) >> 16 from 2^-5 (641 entries).
struct HalfTable {
  int gamma = 0;
  float split = 0.0f;      // x < split: H(x) = half(x * low_scale) (exact product); kGammaLinear: split = 2 (always)
  float low_scale = 1.0f;  // 1/16, 1/12.92f, 1, 1/4.5f (float constants of the reference)
  float pre_add = 0.0f, pre_scale = 1.0f, exponent = 1.0f;  // the curve above the split: ((x + pre_add) * pre_scale) ^ exponent
  float index_scale = 1.0f;  // s above: RN(split * s) >= 2^-4 > RN(pred(split) * s)
  uint32_t h_min = 0;      // H(split): first code the table covers
  // T[i] = smallest x with H(x) >= h_min + i (T[0] may lie below the split); one +inf entry past H(1.0); padded to 16 bytes
  std::vector<float> thresholds;
  // candidate entries {intercept, slope} of bucket i = 0 .. half_cand_index(1.0); [0] = {0, low_scale}; empty without a curve
  std::vector<float> cand;
};
constexpr uint32_t kHalfCandEdge = 0x2c00u;                // binary16 bits of 2^-4: where the split point lands
constexpr uint32_t kHalfCandFloor = kHalfCandEdge - 8u;    // the bucket below it: every x under the split is held here
// bucket of x, computed exactly as the kernel does (product, round-toward-zero conversion, max, shift)
uint32_t half_cand_index(const HalfTable &t, float x);
// the candidate exactly as the kernel forms it (the bucket's entry, one fma), for host-side replay
float half_candidate(const HalfTable &t, float x);
uint16_t float_to_half(float v);               // IEEE binary32 -> binary16, round to nearest even
float curve_to_linear(int gamma, float v);     // the per-gamma video curve alone (what pass 1 writes to a float target)
bool build_half_table(int gamma, HalfTable *out);

}  // namespace bt709
