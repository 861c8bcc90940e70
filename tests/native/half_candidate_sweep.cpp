// TEST INFRASTRUCTURE.  Host replay of the RGBA16Float kernel's CANDIDATE (metalbt709decoder_amd/csrc/transfer_tables.cpp
// half_cand_index + half_candidate: the scaled product, its round-toward-zero binary16 bucket held at the floor, and one fma
// over the product's own table -- the operations the kernel issues) against the oracle's curve (oracle/bt709_oracle.h:
// bt709o_curve_to_linear with the reference's double pow, bt709o_float_to_half), for EVERY float x from 0 to 1.0:
//   * the candidate never exceeds the true value,
//   * its half is H(x) or H(x) - 1 -- which is all the kernel's single-threshold settlement needs -- and
//   * below the curve's split point it IS the reference's product (bucket 0, entry {0, low_scale}): its half is H(x), and
//     from the split point on the bucket is never 0.
// Built by tests/test_rgba16f.py with g++ (-ffp-contract=off) together with transfer_tables.cpp; multithreaded, seconds.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "transfer_tables.h"

extern "C" {
#include "bt709_oracle.h"
}

namespace {
float bits_to_float(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
uint32_t float_to_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
}  // namespace

// out[0] = floats swept, out[1] = candidates above the true value, out[2] = halves outside {H - 1, H} (below the split: other
// than H, or a bucket on the wrong side of the split), out[3] = halves equal to H - 1 (informative), out[4] = bits of the
// first offender (0: none), out[5] = floats from the split point on
extern "C" int sweep_half_candidate(int gamma, int nthreads, uint64_t out[6]) {
  bt709::HalfTable t;
  if (!bt709::build_half_table(gamma, &t) || t.cand.empty()) return -1;
  const uint32_t lo = 0, hi = 0x3f800000u, split_bits = float_to_bits(t.split);
  std::atomic<uint64_t> above{0}, outside{0}, below{0}, first{0};
  std::vector<std::thread> pool;
  if (nthreads < 1) nthreads = 1;
  const uint64_t span = (static_cast<uint64_t>(hi) - lo + 1 + nthreads - 1) / nthreads;
  for (int k = 0; k < nthreads; ++k) {
    const uint64_t a = lo + span * k, b = a + span < static_cast<uint64_t>(hi) + 1 ? a + span : static_cast<uint64_t>(hi) + 1;
    if (a >= b) break;
    pool.emplace_back([&, a, b] {
      uint64_t ab = 0, ou = 0, be = 0;
      for (uint64_t u = a; u < b; ++u) {
        const float x = bits_to_float(static_cast<uint32_t>(u));
        const float truth = bt709o_curve_to_linear(gamma, x), p = bt709::half_candidate(t, x);
        const uint32_t H = bt709o_float_to_half(truth), h = bt709o_float_to_half(p);
        const bool low_piece = u < split_bits;
        const bool wrong_bucket = (bt709::half_cand_index(t, x) == 0) != low_piece;
        if (p > truth) ++ab;
        if (h == H - 1 && !low_piece && !wrong_bucket) ++be;
        else if (h != H || wrong_bucket) {
          ++ou;
          uint64_t expect = 0;
          first.compare_exchange_strong(expect, u);
        }
      }
      above += ab, outside += ou, below += be;
    });
  }
  for (auto &th : pool) th.join();
  out[0] = static_cast<uint64_t>(hi) - lo + 1;
  out[1] = above, out[2] = outside, out[3] = below, out[4] = first, out[5] = static_cast<uint64_t>(hi) - split_bits + 1;
  return 0;
}
