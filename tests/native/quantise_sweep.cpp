// TEST INFRASTRUCTURE.  Host replay of the product's alpha / quantiser arithmetic
// (metalbt709decoder_amd/csrc/bt709_quantise.h -- the SAME source text hipcc compiles into the kernels)
// against the oracle (oracle/bt709_oracle.h: bt709o_quantize = (int)round(x * 255.0f), Renderer/BT709.h:881-883),
// over every input those functions can meet.  Built by tests/test_quantiser_exact.py with
//   g++ -O2 -ffp-contract=off -fno-fast-math -shared -fPIC ... -loracle
// and called through ctypes.  Multithreaded; each sweep takes seconds on 8 cores.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "bt709_quantise.h"

extern "C" {
#include "bt709_oracle.h"
}

namespace {

float bits_to_float(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

template <class F>
void parallel(uint64_t lo, uint64_t hi, int nthreads, F body) {
  if (nthreads < 1) nthreads = 1;
  std::vector<std::thread> pool;
  const uint64_t span = (hi - lo + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    const uint64_t a = lo + span * t, b = a + span < hi ? a + span : hi;
    if (a >= b) break;
    pool.emplace_back([=] { body(a, b); });
  }
  for (auto &th : pool) th.join();
}

}  // namespace

extern "C" {

// Every float with bits in [lo_bits, hi_bits] (inclusive; [0, 0x3f800000] = all of [0, 1]).
// out[0] = floats where quantise_exact differs from the oracle (must be 0)
// out[1] = floats where quantise_enumerated differs (the documented flaw: exactly one)
// out[2] = bits of the first float where quantise_enumerated differs
void sweep_unit_floats(uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint64_t out[3]) {
  std::atomic<uint64_t> bad_exact{0}, bad_enum{0}, first{~0ull};
  parallel(lo_bits, static_cast<uint64_t>(hi_bits) + 1, nthreads, [&](uint64_t a, uint64_t b) {
    uint64_t be = 0, bn = 0, f = ~0ull;
    for (uint64_t u = a; u < b; ++u) {
      const float x = bits_to_float(static_cast<uint32_t>(u));
      const uint32_t want = static_cast<uint32_t>(bt709o_quantize(x));
      if (bt709::quantise_exact(x) != want) ++be;
      if (bt709::quantise_enumerated(x) != want) {
        ++bn;
        if (u < f) f = u;
      }
    }
    bad_exact += be;
    bad_enum += bn;
    uint64_t cur = first.load();
    while (f < cur && !first.compare_exchange_weak(cur, f)) {
    }
  });
  out[0] = bad_exact;
  out[1] = bad_enum;
  out[2] = first;
}

// The 256 codes of an alpha sample: x = its saturated luma term (the oracle's matrix step; the kernels' x is the
// same float, proven by the exhaustive GPU sweeps).  Returns the number of codes where the product's
// alpha_norm_of_unit(x) is not byteNorm(decoded alpha byte) as the oracle computes it; fills norm[256].
int sweep_alpha_samples(float norm[256]) {
  int bad = 0;
  for (int a = 0; a < 256; ++a) {
    float n[3];
    bt709o_ycbcr_to_rgbn(a, 128, 128, n);
    const float got = bt709::alpha_norm_of_unit(n[0]);
    const float want = bt709o_decode_alpha(a) * (1.0f / 255.0f);  // sRGB.h:32-36 byteNorm
    norm[a] = got;
    if (std::memcmp(&got, &want, 4) != 0) ++bad;
  }
  return bad;
}

// All 256^4 ORDERED tuples of tap values byte * (1/255f) (a superset of what four decoded samples can be):
// the product's half_alpha_sum_to_byte against the oracle's (((a+b)+c)+d) * 0.25f -> (int)round(v * 255.0f)
// (oracle/bt709_oracle.c bt709o_decode_nv12_half, alpha branch).  Returns the number of tuples that differ.
uint64_t sweep_half_alpha_tuples(int nthreads) {
  float n[256];
  for (int b = 0; b < 256; ++b) n[b] = b * (1.0f / 255.0f);
  std::atomic<uint64_t> bad{0};
  parallel(0, 256 * 256, nthreads, [&](uint64_t lo, uint64_t hi) {
    uint64_t mine = 0;
    for (uint64_t ab = lo; ab < hi; ++ab) {
      const float a = n[ab >> 8], b = n[ab & 255];
      for (int ci = 0; ci < 256; ++ci) {
        const float c = n[ci];
        for (int di = 0; di < 256; ++di) {
          const float d = n[di];
          const float s = (((a + b) + c) + d) * 0.25f;
          const uint32_t want = static_cast<uint32_t>(bt709o_quantize(s));
          if (bt709::half_alpha_sum_to_byte(a, b, c, d) != want) ++mine;
        }
      }
    }
    bad += mine;
  });
  return bad;
}

}  // extern "C"
