// See transfer_tables.h.  Host code; runs once per decoder in -setupMetal's place.
//
// C++ note: every pow() below is called on explicit doubles.  The reference is C,
// where pow(float,float) IS the double function (sRGB.h:54,70; BT709.h:77,133);
// in C++ the same spelling would pick the float overload and change results.
#include "transfer_tables.h"

#include <cfenv>
#include <cmath>
#include <cstring>
#include <limits>

namespace bt709 {

float srgb_to_linear(float v) {
  if (v <= 0.04045f) return v * (1.0f / 12.92f);
  const float a = 0.055f;
  const float gamma = 2.4f;
  const float base = (v + a) * (1.0f / (1.0f + a));
  return static_cast<float>(std::pow(static_cast<double>(base), static_cast<double>(gamma)));
}

float linear_to_srgb(float v) {
  if (v <= 0.0031308f) return v * 12.92f;
  const float a = 0.055f;
  const float gamma = 1.0f / 2.4f;
  // whole expression in double, narrowed once (sRGB.h:70)
  const double r = static_cast<double>(1.0f + a) *
                       std::pow(static_cast<double>(v), static_cast<double>(gamma)) -
                   static_cast<double>(a);
  return static_cast<float>(r);
}

float itu709_to_linear(float v) {
  if (v < 0.081f) return v * (1.0f / 4.5f);
  const float a = 0.099f;
  const float gamma = 1.0f / 0.45f;
  const float base = (v + a) * (1.0f / (1.0f + a));
  return static_cast<float>(std::pow(static_cast<double>(base), static_cast<double>(gamma)));
}

float apple196_to_linear(float v) {
  const float x_intercept = 0.05583828f;
  if (v < x_intercept) return v * (1.0f / 16.0f);
  const float gamma = 1.960938f;  // APPLE_GAMMA_196, BT709.h:122
  return static_cast<float>(std::pow(static_cast<double>(v), static_cast<double>(gamma)));
}

float linear_to_apple196(float v) {
  const float y_intercept = 0.00349f;
  if (v < y_intercept) return v * 16.0f;
  const float gamma = 1.0f / 1.960938f;  // 1.0f / APPLE_GAMMA_196, BT709.h:146
  return static_cast<float>(std::pow(static_cast<double>(v), static_cast<double>(gamma)));
}

int quantize_byte(float v) {
  const float scaled = v * 255.0f;  // float multiply first
  return static_cast<int>(std::round(static_cast<double>(scaled)));
}

int transfer_to_byte(int gamma, float v) {
  switch (gamma) {
    case kGammaApple:
      return quantize_byte(linear_to_srgb(apple196_to_linear(v)));
    case kGammaSRGB:  // sRGB_to_sRGB_convertYCbCrToRGB applies no curve (BT709.h:977-983)
      return quantize_byte(v);
    case kGammaLinear:  // BT709_from_linear(v, Srgb), BT709.h:1156-1166
      return quantize_byte(linear_to_srgb(v));
    case kGammaITU709:
      return quantize_byte(linear_to_srgb(itu709_to_linear(v)));
    case kTableEncodeApple:  // BT709_from_linear(v, Apple), BT709.h:1158-1159
      return quantize_byte(linear_to_apple196(v));
    default:
      return -1;
  }
}

namespace {

float from_bits(uint32_t u) {
  float f;
  std::memcpy(&f, &u, sizeof f);
  return f;
}

// smallest float in [0,1] mapped to >= k; monotonicity makes bisection on the bit
// pattern valid (non-negative floats order like their bit patterns)
float find_threshold(int gamma, int k) {
  uint32_t lo = 0, hi = 0x3f800000u;
  if (transfer_to_byte(gamma, from_bits(hi)) < k) return std::numeric_limits<float>::infinity();
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (transfer_to_byte(gamma, from_bits(mid)) >= k) hi = mid;
    else lo = mid + 1;
  }
  return from_bits(lo);
}

uint32_t to_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, sizeof u);
  return u;
}

}  // namespace

uint32_t bucket_index(float x, float magic) {
  volatile float s = x + magic;  // binary32 add, round to nearest even (volatile: no excess precision, no folding)
  return to_bits(s) - to_bits(magic);
}

bool build_transfer_table(int gamma, TransferTable *out) {
  if (gamma < 0 || gamma >= kTableKinds || out == nullptr) return false;
  out->gamma = gamma;
  for (int k = 1; k <= 255; ++k) out->thresholds[k - 1] = find_threshold(gamma, k);

  float lin_of_byte[257];
  for (int b = 0; b < 256; ++b) lin_of_byte[b] = srgb_to_linear(b * (1.0f / 255.0f));
  lin_of_byte[256] = lin_of_byte[255];

  const float inf = std::numeric_limits<float>::infinity();
  for (uint32_t n = 256; n <= 65536; n *= 2) {
    const float magic = 8388608.0f / static_cast<float>(n);  // 2^23 / N
    std::vector<TransferBucket> b(n + 1, TransferBucket{inf, 0u});
    // bucket_index is monotone, so with every threshold filed under its own bucket
    //   base[q] = number of thresholds in buckets < q,  edge[q] = the threshold filed under q (if any)
    // gives byte(x) = base[q(x)] + (x >= edge[q(x)]) for every x in [0, 1].
    bool ok = true;
    std::vector<uint32_t> held(n + 2, 0u);
    for (int k = 0; k < 255 && ok; ++k) {
      if (out->thresholds[k] == inf) continue;  // byte k + 1 is never reached
      const uint32_t q = bucket_index(out->thresholds[k], magic);
      if (q > n || held[q] != 0) ok = false;  // a second threshold in one bucket would be lost
      else held[q] = 1, b[q].edge = out->thresholds[k];
    }
    if (!ok) continue;
    uint32_t before = 0;
    for (uint32_t q = 0; q <= n; ++q) {
      b[q].base = before;
      before += held[q];
    }
    out->n = n;
    out->buckets_unit = b;  // N + 1 buckets, edges in x units
    while ((out->buckets_unit.size() * sizeof(TransferBucket)) % 16 != 0)
      out->buckets_unit.push_back(TransferBucket{inf, 255u});
    out->buckets_linear.resize(b.size());
    for (size_t q = 0; q < b.size(); ++q) {
      TransferBucketLinear &e = out->buckets_linear[q];
      e.edge_pred = b[q].edge == inf ? inf : std::nextafter(b[q].edge, -inf);
      e.base = b[q].base;
      e.lin_below = std::ldexp(lin_of_byte[b[q].base], kLinearScaleLog2);      // exact: power of two
      e.lin_above = std::ldexp(lin_of_byte[b[q].base + 1], kLinearScaleLog2);
    }
    // the log-bucket form (transfer_tables.h): the smallest of log_add = 2^-3 .. 2^-10 that keeps one threshold per bucket,
    // taken if it is at most half the uniform table
    out->log_add = 0.0f;
    out->log_first = 0;
    out->buckets_log.clear();
    for (int e = 3; e <= 10; ++e) {
      const float add = std::ldexp(1.0f, -e);
      const uint32_t first = bucket_index_log(0.0f, add), count = bucket_index_log(1.0f, add) - first + 1;
      if (2 * count > n + 1) break;  // smaller adds only add buckets
      std::vector<TransferBucket> lb(count, TransferBucket{inf, 0u});
      std::vector<uint32_t> lheld(count, 0u);
      bool fits = true;
      for (int k = 0; k < 255 && fits; ++k) {
        if (out->thresholds[k] == inf) continue;
        const uint32_t q = bucket_index_log(out->thresholds[k], add) - first;
        if (q >= count || lheld[q] != 0) fits = false;
        else lheld[q] = 1, lb[q].edge = out->thresholds[k];
      }
      if (!fits) continue;
      uint32_t seen = 0;
      for (uint32_t q = 0; q < count; ++q) {
        lb[q].base = seen;
        seen += lheld[q];
      }
      out->log_add = add;
      out->log_first = first;
      out->buckets_log = lb;
      while ((out->buckets_log.size() * sizeof(TransferBucket)) % 16 != 0) out->buckets_log.push_back(TransferBucket{inf, 255u});
      break;
    }
    return true;
  }
  return false;
}

uint32_t bucket_index_log(float x, float log_add) {
  volatile float t = x + log_add;  // one binary32 add, round to nearest even (v_add_f32)
  return to_bits(t) >> 16;
}

uint16_t float_to_half(float v) {
  const uint32_t u = to_bits(v), sign = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7fffffffu;
  if (a >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0u));  // inf / nan
  if (a >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);  // >= 65520 rounds to infinity
  if (a < 0x33000001u) return static_cast<uint16_t>(sign);             // <= 2^-25 rounds to zero (ties to even)
  const int exp = static_cast<int>(a >> 23) - 127;
  uint32_t mant = (a & 0x7fffffu) | 0x800000u;  // 24 bits, hidden one explicit
  int shift;                                     // bits to drop
  uint32_t base;
  if (exp >= -14) {  // normal half
    shift = 13;
    base = static_cast<uint32_t>(exp + 15 - 1) << 10;  // exponent field minus the hidden one, which the mantissa adds back
  } else {           // subnormal half: value = mant * 2^(exp-23), unit 2^-24
    shift = 13 + (-14 - exp);
    base = 0;
  }
  const uint32_t kept = mant >> shift, rest = mant & ((1u << shift) - 1u), half_ulp = 1u << (shift - 1);
  uint32_t h = base + kept;
  if (rest > half_ulp || (rest == half_ulp && (kept & 1u))) ++h;  // nearest, ties to even (carry into the exponent is correct)
  return static_cast<uint16_t>(sign | h);
}

float curve_to_linear(int gamma, float v) {
  switch (gamma) {
    case kGammaApple: return apple196_to_linear(v);
    case kGammaSRGB: return srgb_to_linear(v);
    case kGammaITU709: return itu709_to_linear(v);
    default: return v;  // kGammaLinear: LinearToLinearSRGBKernel applies no curve (AAPLShaders.metal:387-407)
  }
}

bool build_half_table(int gamma, HalfTable *out) {
  if (gamma < 0 || gamma >= kGammaCount || out == nullptr) return false;
  const float inf = std::numeric_limits<float>::infinity();
  out->gamma = gamma;
  out->pre_add = 0.0f;
  out->pre_scale = 1.0f;
  out->exponent = 1.0f;
  out->low_scale = 1.0f;
  out->thresholds.clear();
  switch (gamma) {
    case kGammaApple:
      out->split = 0.05583828f;  // x < split (BT709.h:131)
      out->low_scale = 1.0f / 16.0f;
      out->exponent = 1.960938f;
      break;
    case kGammaSRGB:
      out->split = std::nextafter(0.04045f, 1.0f);  // x <= 0.04045f takes the linear piece (sRGB.h:45)
      out->low_scale = 1.0f / 12.92f;
      out->pre_add = 0.055f;
      out->pre_scale = 1.0f / (1.0f + 0.055f);
      out->exponent = 2.4f;
      break;
    case kGammaITU709:
      out->split = 0.081f;  // x < 0.081f (BT709.h:71)
      out->low_scale = 1.0f / 4.5f;
      out->pre_add = 0.099f;
      out->pre_scale = 1.0f / (1.0f + 0.099f);
      out->exponent = 1.0f / 0.45f;
      break;
    default:  // no curve: the hardware conversion alone is exact
      out->split = 2.0f;
      out->h_min = 0;
      out->thresholds.assign(4, inf);
      return true;
  }
  auto H = [gamma](float x) { return static_cast<uint32_t>(float_to_half(curve_to_linear(gamma, x))); };
  const uint32_t lo_bits = to_bits(out->split), hi_bits = 0x3f800000u;
  out->h_min = H(out->split);
  const uint32_t h_max = H(1.0f);
  {  // T[h_min] over the WHOLE curve (the piece below the split included): the kernel settles values of
     // that piece against it too, and they must stay where the exact product put them
    uint32_t lo = 0, hi = lo_bits;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (H(from_bits(mid)) >= out->h_min) hi = mid;
      else lo = mid + 1;
    }
    out->thresholds.push_back(from_bits(lo));
  }
  for (uint32_t h = out->h_min + 1; h <= h_max; ++h) {
    uint32_t lo = lo_bits, hi = hi_bits;  // H is monotone: bisect on the bit pattern
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (H(from_bits(mid)) >= h) hi = mid;
      else lo = mid + 1;
    }
    out->thresholds.push_back(from_bits(lo));
  }
  out->thresholds.push_back(inf);  // T[h_max + 1]
  while (out->thresholds.size() % 4 != 0) out->thresholds.push_back(inf);
  // The index scale (transfer_tables.h): a float s with RN(split * s) >= 2^-4 > RN(pred(split) * s), searched among the
  // neighbours of 2^-4 / split -- the interval of real s that works is about one float wide, and the product's own rounding
  // decides; the build fails if none of them does (it does for all three curves).
  {
    const float edge = 0.0625f, below = std::nextafter(out->split, 0.0f);
    const uint32_t s0 = to_bits(edge / out->split);
    bool found = false;
    for (int d = 0; d <= 16 && !found; ++d) {
      for (int sign = -1; sign <= 1 && !found; sign += 2) {
        const float s = from_bits(s0 + static_cast<uint32_t>(sign * d));
        volatile float at = out->split * s, under = below * s;
        if (at >= edge && under < edge) {
          out->index_scale = s;
          found = true;
        }
      }
    }
    if (!found) return false;
  }
  // Candidate entries, slope / intercept form.  Bucket 0 (below the split): the exact product.  Bucket i >= 1: p(x) = slope *
  // x + intercept is the tangent of the curve at x_q = the bucket's first float (bisected on the index function itself; x_q of
  // bucket 1 is the split point) scaled by `keep` = 1 - 2^-19.  Computed in double from the curve's constants; both floats are
  // then moved one step toward a SMALLER p (the slope is positive and x > 0: down; the intercept: down), so that neither their
  // rounding, nor the fma's own, nor the reference's float steps can lift p above the true value
  // (tests/native/half_candidate_sweep.cpp checks every float from 0 to 1.0).
  out->cand.clear();
  out->cand.push_back(0.0f);
  out->cand.push_back(out->low_scale);
  const double a = out->pre_add, s = out->pre_scale, g = out->exponent, keep = 1.0 - 1.0 / 524288.0;
  const float ninf = -std::numeric_limits<float>::infinity();
  const uint32_t last = half_cand_index(*out, 1.0f);
  uint32_t start = lo_bits;
  for (uint32_t i = 1; i <= last; ++i) {
    uint32_t lo = start, hi = hi_bits;  // the index is monotone in x: bisect on the bit pattern
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (half_cand_index(*out, from_bits(mid)) >= i) hi = mid;
      else lo = mid + 1;
    }
    start = lo;
    const double xq = from_bits(lo);
    const double base = (xq + a) * s;
    const double value = std::pow(base, g) * keep, slope = g * s * std::pow(base, g - 1.0) * keep;
    out->cand.push_back(std::nextafter(static_cast<float>(value - slope * xq), ninf));
    out->cand.push_back(std::nextafter(static_cast<float>(slope), ninf));
  }
  if (half_cand_index(*out, out->split) != 1 || half_cand_index(*out, std::nextafter(out->split, 0.0f)) != 0) return false;
  return true;
}

uint32_t half_cand_index(const HalfTable &t, float x) {
  volatile float u = x * t.index_scale;  // v_pk_mul_f32: one binary32 rounding
  const uint32_t ub = to_bits(u);
  // v_cvt_pkrtz_f16_f32 of a u in [2^-14, 65504): rebias the exponent, drop 13 mantissa bits.  Smaller u (binary16
  // subnormals, zero) convert to something below the floor and are held there by the max, whatever the hardware makes of them.
  uint32_t hb = ub >= (113u << 23) ? (ub - (112u << 23)) >> 13 : 0u;
  if (hb < kHalfCandFloor) hb = kHalfCandFloor;
  return (hb >> 3) - (kHalfCandFloor >> 3);
}

float half_candidate(const HalfTable &t, float x) {
  if (t.cand.empty() || !(x >= 0.0f && x <= 1.0f)) return 0.0f;
  const uint32_t i = half_cand_index(t, x);
  volatile float p = std::fmaf(x, t.cand[2 * i + 1], t.cand[2 * i]);  // the kernel's one v_fma_f32
  return p;
}

uint32_t uniform_index(float v, float n) {
  volatile float t = std::fmaf(v, n, 8388608.0f);  // one rounding of v n + 2^23, as the kernel's v_fma_f32
  return to_bits(t) - 0x4b000000u;
}

bool build_uniform_table(int kind, uint32_t n_min, UniformTable *out) {
  if (kind < 0 || kind >= kTableKinds || out == nullptr) return false;
  float thr[255];
  for (int k = 1; k <= 255; ++k) thr[k - 1] = find_threshold(kind, k);
  const float inf = std::numeric_limits<float>::infinity();
  for (uint32_t n = (n_min + 31) / 32 * 32; n <= 65536; n += 32) {
    const float nf = static_cast<float>(n);
    std::vector<TransferBucket> b(n + 2, TransferBucket{inf, 0u});
    std::vector<uint32_t> held(n + 2, 0u);
    bool ok = true;
    for (int k = 0; k < 255 && ok; ++k) {
      if (thr[k] == inf) continue;
      const uint32_t q = uniform_index(thr[k], nf);
      if (q > n + 1 || held[q] != 0) ok = false;
      else held[q] = 1, b[q].edge = thr[k];
    }
    if (!ok || uniform_index(1.0f, nf) > n + 1) continue;
    uint32_t before = 0;
    for (uint32_t q = 0; q < n + 2; ++q) {
      b[q].base = before;
      before += held[q];
    }
    while ((b.size() * sizeof(TransferBucket)) % 16 != 0) b.push_back(TransferBucket{inf, 255u});
    out->n = n;
    out->buckets = b;
    return true;
  }
  return false;
}

bool build_split_table(int kind, SplitTable *out) {
  if (kind < 0 || kind >= kTableKinds || out == nullptr) return false;
  float thr[255];
  for (int k = 1; k <= 255; ++k) thr[k - 1] = find_threshold(kind, k);
  const float inf = std::numeric_limits<float>::infinity();
  // candidates, smallest table first; the last one is a plain uniform table
  const struct { uint32_t n_fine, split, ratio; } cand[] = {
      {256, 256, 1},   {512, 512, 1},   {1024, 64, 4},  {2048, 128, 4},  {4096, 256, 8},
      {4096, 256, 4},  {4096, 512, 4},  {8192, 512, 8}, {8192, 1024, 4}, {4096, 4096, 1}, {8192, 8192, 1}};
  for (const auto &c : cand) {
    const uint32_t coarse_n = c.n_fine / c.ratio;  // coarse buckets over [0,1]
    const uint32_t offset = c.split - c.split / c.ratio;
    const uint32_t total = c.split + (coarse_n - c.split / c.ratio) + 1;  // +1: x == 1.0
    std::vector<TransferBucket> b(total);
    bool ok = true;
    int k = 0;
    for (uint32_t q = 0; q < total && ok; ++q) {
      float lo, hi;  // bucket bounds in x
      if (q < c.split) {
        lo = static_cast<float>(q) / c.n_fine;
        hi = static_cast<float>(q + 1) / c.n_fine;
      } else {
        const uint32_t j = q - offset;  // coarse bucket index
        lo = static_cast<float>(j) / coarse_n;
        hi = static_cast<float>(j + 1) / coarse_n;
      }
      while (k < 255 && thr[k] <= lo) ++k;
      b[q].base = static_cast<uint32_t>(k);
      b[q].edge = inf;
      if (k < 255 && thr[k] < hi) {
        b[q].edge = thr[k] * static_cast<float>(c.n_fine);  // exact: power of two
        if (k + 1 < 255 && thr[k + 1] < hi) ok = false;
      }
    }
    if (!ok) continue;
    while ((b.size() * sizeof(TransferBucket)) % 16 != 0) b.push_back(TransferBucket{inf, 255u});
    out->n_fine = c.n_fine;
    out->split = static_cast<float>(c.split);
    out->coarse_scale = 1.0f / static_cast<float>(c.ratio);
    out->coarse_offset = offset;
    out->buckets = b;
    return true;
  }
  return false;
}

bool build_encode_tables(int in_gamma, int out_gamma, EncodeTables *out) {
  if (out == nullptr) return false;
  auto ok = [](int g) { return g == kGammaApple || g == kGammaSRGB || g == kGammaLinear; };
  if (!ok(in_gamma) || !ok(out_gamma)) return false;
  out->from_linear_kind = out_gamma == kGammaSRGB ? kGammaLinear : (out_gamma == kGammaLinear ? kGammaSRGB : kTableEncodeApple);
  for (int b = 0; b < 256; ++b) {
    const float n = b * (1.0f / 255.0f);  // byteNorm
    float lin = n;                        // BT709_tolinearNorm, BT709.h:1125-1139
    if (in_gamma == kGammaSRGB) lin = srgb_to_linear(n);
    else if (in_gamma == kGammaApple) lin = apple196_to_linear(n);
    const int e = transfer_to_byte(out->from_linear_kind, lin);  // BT709_from_linear(lin, outputGamma)
    out->per_byte[b].lin = lin;
    out->per_byte[b].enc_norm = e * (1.0f / 255.0f);  // byteNorm inside sRGB_from_sRGB_convertRGBToYCbCr
  }
  return true;
}

}  // namespace bt709
