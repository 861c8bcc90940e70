// Frame ring with placement-aware allocation (include/bt709hip_ext.h, "frame ring").
//
// A streaming application -- and bench.py -- keeps a ring of same-sized frames resident in HBM: N NV12 inputs carved from
// one slab, N BGRA outputs from another, decoded in long launches (bt709hip_decode_batch over evenly spaced frames).  On
// MI355X the rate at which such a launch streams depends on WHERE the two slabs landed (allocations made one
// after the other by one process run the same launch at 0.74-0.82 of the HBM roofline, each keeping its rate; the output
// slab carries most of it, the pairing with the input slab a further 1-2 %).  Rounds 2-3 hunted for a good pairing inside
// bench.py; this file is that hunt as product behaviour: bt709hip_ring_create allocates `tries` candidates per slab, times
// the decoder's OWN launch over the pairings and keeps the fastest.  Written on top of the public C ABI only.
#include "../../include/bt709hip_ext.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <new>
#include <utility>
#include <vector>

struct bt709hip_ring {
  bt709hip_decoder *dec = nullptr;
  bt709hip_context *ctx = nullptr;
  int width = 0, height = 0, frames = 0, half = 0, has_alpha = 0, format = BT709HIP_FORMAT_BGRA8_SRGB;
  size_t y_bytes = 0, c_bytes = 0, in_stride = 0, out_stride = 0;
  void *d_in = nullptr, *d_out = nullptr;
  std::vector<bt709hip_frame> f, a;
  std::vector<bt709hip_surface> o;
  bt709hip_ring_placement placement;
};

namespace {

constexpr size_t kHuntMinBytes = 256u << 20;  // a ring that fits the 256 MB memory-side cache has no placement to hunt for
constexpr size_t kReserveBytes = 4ull << 30;  // device memory the hunt always leaves free
constexpr int kMaxTries = 6;
// The hunt's own cost (round 6: the default hunt must fit ~1.5 s for a 12 GB ring; tools/ab_hunt.sh times variant builds of these)
#ifndef BT709_HUNT_WARM_S
#define BT709_HUNT_WARM_S 0.01  // synchronous launches over a candidate before its probe (page tables, clocks; the first candidate gets 150 ms).
                                // 30 ms until round 6: 10 ms separates the placement levels as well (profiles/r06_hunt_default.txt box 2,
                                // profiles/r06_partial_probe.txt) and takes a quarter of a second off a twelve-candidate hunt
#endif
#ifndef BT709_HUNT_CONFIRM_X
#define BT709_HUNT_CONFIRM_X 6  // the finalists' probes are this many times as long as a prescan probe
#endif
#ifndef BT709_HUNT_FRUGAL_INPUTS
#define BT709_HUNT_FRUGAL_INPUTS 1  // input candidates of the frugal hunt (the input slab moves the rate by ~1 %, the output slab by ~10 %)
#endif
constexpr int kMaxOutCandidates = 3 * kMaxTries;  // == the arrays of bt709hip_ring_placement

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int required_transfer(int gamma) {
  switch (gamma) {
    case BT709HIP_GAMMA_SRGB: return BT709HIP_TRANSFER_SRGB;
    case BT709HIP_GAMMA_LINEAR: return BT709HIP_TRANSFER_LINEAR;
    default: return BT709HIP_TRANSFER_ITU_R_709_2;
  }
}

void bind_slabs(bt709hip_ring *r, void *in, void *out) {
  r->d_in = in;
  r->d_out = out;
  const int ow = r->half ? r->width / 2 : r->width, oh = r->half ? r->height / 2 : r->height;
  const int transfer = required_transfer(bt709hip_decoder_get_gamma(r->dec));
  for (int i = 0; i < r->frames; ++i) {
    uint8_t *base = static_cast<uint8_t *>(in) + static_cast<size_t>(i) * r->in_stride;
    bt709hip_frame &f = r->f[static_cast<size_t>(i)];
    std::memset(&f, 0, sizeof f);
    f.y = base;
    f.y_stride = static_cast<size_t>(r->width);
    f.cbcr = base + r->y_bytes;
    f.cbcr_stride = static_cast<size_t>(r->width);
    f.width = r->width;
    f.height = r->height;
    f.matrix = BT709HIP_MATRIX_ITU_R_709_2;
    f.transfer = transfer;
    if (r->has_alpha) {
      bt709hip_frame &a = r->a[static_cast<size_t>(i)];
      a = f;
      a.y = base + r->y_bytes + r->c_bytes;
      a.cbcr = nullptr;
      a.transfer = BT709HIP_TRANSFER_LINEAR;
    }
    bt709hip_surface &o = r->o[static_cast<size_t>(i)];
    std::memset(&o, 0, sizeof o);
    o.bgra = static_cast<uint8_t *>(out) + static_cast<size_t>(i) * r->out_stride;
    o.stride = static_cast<size_t>(ow) * (r->format == BT709HIP_FORMAT_RGBA16F ? 8 : 4);
    o.width = ow;
    o.height = oh;
    o.format = r->format;
  }
}

int launch(bt709hip_ring *r, int first, int count, void *stream, int wait) {
  const bt709hip_frame *f = r->f.data() + first;
  const bt709hip_frame *a = r->has_alpha ? r->a.data() + first : nullptr;
  const bt709hip_surface *o = r->o.data() + first;
  return r->half ? bt709hip_decode_half_batch(r->dec, count, f, a, o, stream, wait)
                 : bt709hip_decode_batch(r->dec, count, f, a, o, stream, wait);
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Prober {
  bt709hip_ring *r;
  void *e0 = nullptr, *e1 = nullptr;
  double bytes_per_launch = 0.0;
  int rc = BT709HIP_OK;

  // GB/s of the ring's own launch over slabs (in, out): `warm_s` seconds of launches first (clocks, page tables), then n
  // launches between two events.  The input is whatever the memory holds -- any bytes decode.
  float measure(void *in, void *out, int n, double warm_s) {
    bind_slabs(r, in, out);
    const double t_end = now_s() + warm_s;
    do {
      if ((rc = launch(r, 0, r->frames, nullptr, 1)) != BT709HIP_OK) return 0.0f;
    } while (now_s() < t_end);
    if ((rc = bt709hip_event_record(r->ctx, e0, nullptr)) != BT709HIP_OK) return 0.0f;
    for (int k = 0; k < n; ++k)
      if ((rc = launch(r, 0, r->frames, nullptr, 0)) != BT709HIP_OK) return 0.0f;
    float ms = 0.0f;
    if ((rc = bt709hip_event_record(r->ctx, e1, nullptr)) != BT709HIP_OK) return 0.0f;
    if ((rc = bt709hip_event_synchronize(r->ctx, e1)) != BT709HIP_OK) return 0.0f;
    if ((rc = bt709hip_event_elapsed_ms(r->ctx, e0, e1, &ms)) != BT709HIP_OK || ms <= 0.0f) return 0.0f;
    return static_cast<float>(bytes_per_launch * n / (ms * 1e-3) / 1e9);
  }
};

// One slab of the hunt.  `rate` = its probe under the reference partner (outputs: under input 0; inputs: under the best output).
struct Slab {
  void *p = nullptr;
  float rate = 0.0f;
  bool alive = false;
};

// Device memory the call holds, against its budget (bt709hip_ring_options): every candidate is alive until the choice is made
// unless the budget says otherwise -- then the SLOWEST output seen so far goes first (the fast ones are what the pairing probes
// need), down to one.
struct Ledger {
  bt709hip_context *ctx;
  size_t budget = 0, held = 0, peak = 0;
  int evicted = 0;

  bool fits(size_t bytes) const {
    if (held + bytes > budget) return false;
    size_t free_b = 0;
    return bt709hip_mem_info(ctx, &free_b, nullptr) == BT709HIP_OK && free_b >= bytes + kReserveBytes;
  }
  void *take(size_t bytes) {
    void *p = nullptr;
    if (bt709hip_malloc(ctx, bytes, &p) != BT709HIP_OK || p == nullptr) return nullptr;
    held += bytes;
    peak = std::max(peak, held);
    return p;
  }
  void give(void *p, size_t bytes) {
    if (p == nullptr) return;
    (void)bt709hip_free(ctx, p);
    held -= bytes;
  }
};

}  // namespace

extern "C" {

int bt709hip_ring_destroy(bt709hip_ring *r) {
  if (r == nullptr) return BT709HIP_OK;
  if (r->ctx) {
    (void)bt709hip_stream_synchronize(r->ctx, nullptr);
    if (r->d_in) (void)bt709hip_free(r->ctx, r->d_in);
    if (r->d_out) (void)bt709hip_free(r->ctx, r->d_out);
  }
  delete r;
  return BT709HIP_OK;
}

int bt709hip_ring_create(bt709hip_decoder *dec, int width, int height, int frames, int half_scale, int tries, bt709hip_ring **out) {
  return bt709hip_ring_create_ex(dec, width, height, frames, half_scale, tries, nullptr, out);
}

int bt709hip_ring_create_ex(bt709hip_decoder *dec, int width, int height, int frames, int half_scale, int tries,
                            const bt709hip_ring_options *options, bt709hip_ring **out) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (dec == nullptr || width <= 0 || height <= 0 || frames <= 0 || frames > 65535 || tries < 0) return BT709HIP_ERR_INVALID_ARG;
  if ((width & 1) || (height & 1)) return BT709HIP_ERR_ODD_DIMENSIONS;
  if (half_scale && ((width & 3) || (height & 3))) return BT709HIP_ERR_ODD_DIMENSIONS;
  const int format = options != nullptr ? options->format : BT709HIP_FORMAT_BGRA8_SRGB;
  if (format != BT709HIP_FORMAT_BGRA8_SRGB && format != BT709HIP_FORMAT_RGBA16F) return BT709HIP_ERR_INVALID_ARG;
  if (format == BT709HIP_FORMAT_RGBA16F && half_scale) return BT709HIP_ERR_UNSUPPORTED;  // the fused 2:1 kernel writes BGRA8
  if (int rc = bt709hip_decoder_setup(dec)) return rc;
  if (int rc = bt709hip_decoder_prepare_format(dec, format)) return rc;  // the RGBA16F threshold table, before any probe
  bt709hip_context *ctx = bt709hip_decoder_context(dec);
  if (ctx == nullptr) return BT709HIP_ERR_NOT_SETUP;
  bt709hip_ring *r = new (std::nothrow) bt709hip_ring();
  if (r == nullptr) return BT709HIP_ERR_INVALID_ARG;
  const double t_start = now_s();
  r->dec = dec;
  r->ctx = ctx;
  r->width = width, r->height = height, r->frames = frames, r->half = half_scale ? 1 : 0, r->format = format;
  const size_t out_px_bytes = format == BT709HIP_FORMAT_RGBA16F ? 8 : 4;
  r->has_alpha = bt709hip_decoder_has_alpha(dec) > 0;
  r->y_bytes = static_cast<size_t>(width) * height;
  r->c_bytes = static_cast<size_t>(width) * (height / 2);
  const int ow = r->half ? width / 2 : width, oh = r->half ? height / 2 : height;
  // frames 256-byte aligned: the fast kernels want 16, a frame boundary on a cache-line boundary costs nothing
  r->in_stride = round_up(r->y_bytes + r->c_bytes + (r->has_alpha ? r->y_bytes : 0), 256);
  r->out_stride = round_up(static_cast<size_t>(ow) * oh * out_px_bytes, 256);
  r->f.resize(static_cast<size_t>(frames));
  r->o.resize(static_cast<size_t>(frames));
  if (r->has_alpha) r->a.resize(static_cast<size_t>(frames));
  bt709hip_ring_placement &pl = r->placement;
  std::memset(&pl, 0, sizeof pl);
  for (int &k : pl.out_kept) k = -1;

  const size_t in_bytes = r->in_stride * frames, out_bytes = r->out_stride * frames;
  if (tries == 0) tries = kMaxTries;
  tries = std::min(tries, kMaxTries);
  if (in_bytes + out_bytes < kHuntMinBytes) tries = 1;

  // THE BUDGET.  Default (round 6): FRUGAL -- the incumbent pair + one candidate pair = twice the ring -- because that is all the
  // hunt needs: free + allocate hands out other physical pages, so holding every candidate buys nothing (round 5,
  // profiles/r05_hunt_budget.txt: the 11.7 GB 4K ring hunted within 23 / 32 / 64 / 148 GB lands at 0.806-0.814 / 0.802-0.811 /
  // 0.811-0.812 / 0.811 of the roofline, in 0.9 / 1.2 / 1.9 / 3.7 s; round 6's five fresh processes on the default:
  // profiles/r06_hunt_default.txt).  A caller who wants the wide hunt back names its budget in options->max_bytes.  A budget that
  // cannot hold the ring and one more output slab leaves nothing to compare.
  const bool frugal = options == nullptr || options->frugal || options->max_bytes == 0;
  Ledger led{ctx};
  {
    size_t budget = frugal ? 2 * (in_bytes + out_bytes) : static_cast<size_t>(options->max_bytes);
    if (tries > 1 && budget < in_bytes + 2 * out_bytes) {
      tries = 1;
      pl.stopped_by = 1;
    }
    led.budget = std::max(budget, in_bytes + out_bytes);  // the ring itself is not negotiable
  }
  const double max_s = options != nullptr && options->max_ms != 0 ? options->max_ms * 1e-3 : 0.0;
  auto time_left = [&]() { return max_s == 0.0 || now_s() - t_start < max_s; };
  pl.tries = tries;
  pl.budget_bytes = led.budget;

  // Inputs first (as round 4 did: they land in different places), all alive to the end -- they are the small slabs and move the
  // rate by ~1 % -- but never more than a quarter of the budget, and always leaving room for two outputs.
  std::vector<Slab> ins, outs;
  {
    int n_in = frugal ? std::min(tries, BT709_HUNT_FRUGAL_INPUTS) : tries;
    while (n_in > 1 && ((!frugal && static_cast<size_t>(n_in) * in_bytes > led.budget / 4) || static_cast<size_t>(n_in) * in_bytes + 2 * out_bytes > led.budget)) --n_in;
    for (int i = 0; i < n_in; ++i) {
      if (i > 0 && !led.fits(in_bytes + out_bytes)) break;
      void *p = led.take(in_bytes);
      if (p == nullptr) break;
      ins.push_back(Slab{p, 0.0f, true});
    }
  }
  auto free_everything = [&]() {
    for (Slab &s : ins)
      if (s.alive) led.give(s.p, in_bytes), s.alive = false;
    for (Slab &s : outs)
      if (s.alive) led.give(s.p, out_bytes), s.alive = false;
  };
  if (!ins.empty()) {
    void *p = led.take(out_bytes);
    if (p != nullptr) outs.push_back(Slab{p, 0.0f, true});
  }
  if (ins.empty() || outs.empty()) {
    free_everything();
    delete r;
    return BT709HIP_ERR_HIP;  // out of device memory: bt709hip_last_hip_error
  }

  int bi = 0, bo = 0;
  Prober pr{r};
  // the probes time the ring's OWN launch: a coalescing decoder would queue a short ring's frames instead (count < n)
  int coalesce = 0;
  (void)bt709hip_decoder_get_option(dec, BT709HIP_OPT_COALESCE, &coalesce);
  if (tries > 1 && bt709hip_event_create(ctx, &pr.e0) == BT709HIP_OK && bt709hip_event_create(ctx, &pr.e1) == BT709HIP_OK) {
    if (coalesce > 1) (void)bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, 0);
    const double px_per_launch = static_cast<double>(width) * height * frames;
    pr.bytes_per_launch = (static_cast<double>(r->y_bytes + r->c_bytes + (r->has_alpha ? r->y_bytes : 0)) + static_cast<double>(ow) * oh * static_cast<double>(out_px_bytes)) * frames;
    // a probe = ~15 ms of the ring's own launches (3 launches resolved the top candidates to only +-1.5 %)
    const int reps = std::max(3, static_cast<int>((8.0 * 256 * 3840 * 2160 + px_per_launch - 1) / px_per_launch));
    // PRESCAN: every output candidate under input 0, one after the other.  Output slabs come in two regimes (~0.74 / ~0.80+ of
    // the roofline for the 1:1 kernel); a process whose candidates all look alike may hold `tries` slow ones (seen: 4 of 4,
    // 6 of 8), so it goes on -- always to twice `tries` (the fast regime itself spreads over 1 %), three times when they still
    // look alike.  When the budget has no room for the next candidate the slowest output alive is freed first.
    auto alive_outs = [&]() { return static_cast<int>(std::count_if(outs.begin(), outs.end(), [](const Slab &s) { return s.alive; })); };
    auto evict_slowest_out = [&]() {
      int worst = -1;
      for (size_t k = 0; k < outs.size(); ++k)
        if (outs[k].alive && (worst < 0 || outs[k].rate < outs[static_cast<size_t>(worst)].rate)) worst = static_cast<int>(k);
      if (worst < 0 || alive_outs() <= 1) return false;
      led.give(outs[static_cast<size_t>(worst)].p, out_bytes);
      outs[static_cast<size_t>(worst)].alive = false;
      ++led.evicted;
      return true;
    };
    outs[0].rate = pr.measure(ins[0].p, outs[0].p, reps, 0.15);
    int target = tries;
    while (pr.rc == BT709HIP_OK) {
      bool stuck = false;
      while (static_cast<int>(outs.size()) < target && pr.rc == BT709HIP_OK) {
        if (!time_left()) {
          pl.stopped_by = 2;
          stuck = true;
          break;
        }
        while (!led.fits(out_bytes) && evict_slowest_out()) pl.stopped_by = pl.stopped_by ? pl.stopped_by : 1;
        void *p = led.fits(out_bytes) ? led.take(out_bytes) : nullptr;
        if (p == nullptr) {
          pl.stopped_by = pl.stopped_by ? pl.stopped_by : 1;
          stuck = true;
          break;
        }
        outs.push_back(Slab{p, 0.0f, true});
        outs.back().rate = pr.measure(ins[0].p, p, reps, BT709_HUNT_WARM_S);
      }
      if (stuck || pr.rc != BT709HIP_OK) break;
      float hi = 0.0f, lo = 1e30f;
      for (const Slab &s : outs) hi = std::max(hi, s.rate), lo = std::min(lo, s.rate);
      const int n = static_cast<int>(outs.size());
      if (n >= 3 * tries || n >= kMaxOutCandidates || (n >= 2 * tries && (hi - lo) / hi >= 0.02f)) break;
      target = std::min(n + tries, kMaxOutCandidates);
    }
    pl.out_candidates = static_cast<int>(outs.size());
    for (size_t o = 0; o < outs.size() && o < static_cast<size_t>(kMaxOutCandidates); ++o) pl.out_prescan_GBps[o] = outs[o].rate;
    // the `tries` fastest outputs still alive go on to the pairing probes
    std::vector<int> order;
    for (size_t k = 0; k < outs.size(); ++k)
      if (outs[k].alive) order.push_back(static_cast<int>(k));
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return outs[static_cast<size_t>(x)].rate > outs[static_cast<size_t>(y)].rate; });
    std::vector<int> kept(order.begin(), order.begin() + std::min<size_t>(order.size(), static_cast<size_t>(tries)));
    std::sort(kept.begin(), kept.end());
    for (size_t k = 0; k < kept.size(); ++k) pl.out_kept[k] = kept[k];
    // every pairing of the inputs with the kept outputs, once
    std::map<std::pair<int, int>, float> probed;
    for (size_t i = 0; i < ins.size() && pr.rc == BT709HIP_OK; ++i)
      for (int o : kept) {
        if (pr.rc != BT709HIP_OK) break;
        if (i > 0 && !time_left()) {
          pl.stopped_by = 2;
          break;
        }
        probed[{static_cast<int>(i), o}] = i == 0 ? outs[static_cast<size_t>(o)].rate : pr.measure(ins[i].p, outs[static_cast<size_t>(o)].p, reps, BT709_HUNT_WARM_S);
      }
    pl.probes = static_cast<int>(probed.size());
    if (pr.rc == BT709HIP_OK && !probed.empty()) {
      pl.first_GBps = outs[0].rate;
      std::vector<std::pair<float, std::pair<int, int>>> ranked;
      for (const auto &kv : probed) ranked.push_back({kv.second, kv.first});
      std::sort(ranked.begin(), ranked.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
      pl.best_GBps = ranked.front().first;
      pl.worst_GBps = ranked.back().first;
      bi = ranked.front().second.first, bo = ranked.front().second.second;
      pl.chosen_GBps = ranked.front().first;
      // the three best -- and the first-allocated pairing, so that a hunt over candidates that are all alike never ends on a
      // pairing a noisy 15 ms probe preferred to the one a caller would have had anyway -- again, six times as long (~90 ms
      // each; 45 ms resolved the top three to +-0.3 %, as much as they differ): the choice is made on these
      std::vector<std::pair<int, int>> finalists;
      for (size_t k = 0; k < ranked.size() && k < 3; ++k) finalists.push_back(ranked[k].second);
      const std::pair<int, int> first_pair{0, 0};  // skipped when the prescan already ranked output 0 among the slow ones
      if (std::find(kept.begin(), kept.end(), 0) != kept.end() && std::find(finalists.begin(), finalists.end(), first_pair) == finalists.end())
        finalists.push_back(first_pair);
      float best = -1.0f;
      if (finalists.size() > 1)
        for (const auto &io : finalists) {
          if (pr.rc != BT709HIP_OK) break;
          if (!time_left()) {  // out of time: the short probes decide
            pl.stopped_by = 2;
            break;
          }
          const float v = pr.measure(ins[static_cast<size_t>(io.first)].p, outs[static_cast<size_t>(io.second)].p, BT709_HUNT_CONFIRM_X * reps, BT709_HUNT_WARM_S);
          if (v > best) best = v, bi = io.first, bo = io.second;
        }
      if (best > 0.0f) pl.chosen_GBps = best;
    }
    if (coalesce > 1) (void)bt709hip_decoder_set_option(dec, BT709HIP_OPT_COALESCE, coalesce);
  }
  if (pr.e0) (void)bt709hip_event_destroy(ctx, pr.e0);
  if (pr.e1) (void)bt709hip_event_destroy(ctx, pr.e1);
  pl.in_candidates = static_cast<int>(ins.size());
  if (pl.out_candidates == 0) pl.out_candidates = static_cast<int>(outs.size());
  pl.chosen_in = bi;
  pl.chosen_out = bo;
  pl.evicted = led.evicted;
  // After a failed probe bi / bo may still name slab 0 of each kind, and output 0 may have been evicted (freed) by the budget:
  // only a slab that is still alive is kept (and, on the error path below, freed) -- never a pointer the ledger gave back.
  void *keep_in = ins[static_cast<size_t>(bi)].alive ? ins[static_cast<size_t>(bi)].p : nullptr;
  void *keep_out = outs[static_cast<size_t>(bo)].alive ? outs[static_cast<size_t>(bo)].p : nullptr;
  ins[static_cast<size_t>(bi)].alive = false;  // not the hunt's any more
  outs[static_cast<size_t>(bo)].alive = false;
  pl.peak_bytes = led.peak;
  free_everything();
  if (pl.tries > 1) pl.hunt_ms = static_cast<float>((now_s() - t_start) * 1e3);
  if (pr.rc != BT709HIP_OK || keep_in == nullptr || keep_out == nullptr) {
    if (keep_in) (void)bt709hip_free(ctx, keep_in);  // live slabs only: a successful free leaves the probe's HIP error in place
    if (keep_out) (void)bt709hip_free(ctx, keep_out);
    const int rc = pr.rc != BT709HIP_OK ? pr.rc : BT709HIP_ERR_HIP;
    delete r;
    return rc;
  }
  bind_slabs(r, keep_in, keep_out);
  *out = r;
  return BT709HIP_OK;
}

int bt709hip_ring_frames(const bt709hip_ring *r) { return r ? r->frames : BT709HIP_ERR_INVALID_ARG; }

int bt709hip_ring_frame(const bt709hip_ring *r, int index, bt709hip_frame *frame, bt709hip_frame *alpha, bt709hip_surface *out) {
  if (r == nullptr || index < 0 || index >= r->frames) return BT709HIP_ERR_INVALID_ARG;
  if (frame) *frame = r->f[static_cast<size_t>(index)];
  if (alpha) {
    if (r->has_alpha) *alpha = r->a[static_cast<size_t>(index)];
    else std::memset(alpha, 0, sizeof *alpha);
  }
  if (out) *out = r->o[static_cast<size_t>(index)];
  return BT709HIP_OK;
}

int bt709hip_ring_placement_info(const bt709hip_ring *r, bt709hip_ring_placement *info) {
  if (r == nullptr || info == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *info = r->placement;
  return BT709HIP_OK;
}

int bt709hip_ring_decode(bt709hip_ring *r, int first, int count, void *stream, int wait_until_completed) {
  if (r == nullptr || first < 0 || count < 0 || first + count > r->frames) return BT709HIP_ERR_INVALID_ARG;
  if (count == 0) return BT709HIP_OK;
  return launch(r, first, count, stream, wait_until_completed);
}

// ----------------------------------------------------------------- ring set
// One process, several GPUs, device-resident frames (include/bt709hip_ext.h "ring set"): a context + decoder + ring per lane, one
// launch per lane per step issued from the calling thread.  Every C-ABI call binds its context's device, so the lanes need no
// thread of their own: a launch call returns as soon as the kernel is enqueued.

}  // extern "C"

struct bt709hip_ringset {
  struct Lane {
    bt709hip_context *ctx = nullptr;
    bt709hip_decoder *dec = nullptr;
    bt709hip_ring *ring = nullptr;
  };
  std::vector<Lane> lanes;
};

extern "C" {

int bt709hip_ringset_destroy(bt709hip_ringset *set) {
  if (set == nullptr) return BT709HIP_OK;
  for (auto &l : set->lanes) {
    if (l.ring) (void)bt709hip_ring_destroy(l.ring);
    if (l.dec) (void)bt709hip_decoder_destroy(l.dec);
    if (l.ctx) (void)bt709hip_context_destroy(l.ctx);
  }
  delete set;
  return BT709HIP_OK;
}

int bt709hip_ringset_create(const int *device_ordinals, int lanes, int gamma, int has_alpha, int width, int height, int frames,
                            int half_scale, int tries, const bt709hip_ring_options *options, bt709hip_ringset **out) {
  if (out == nullptr) return BT709HIP_ERR_INVALID_ARG;
  *out = nullptr;
  if (device_ordinals == nullptr || lanes <= 0 || lanes > 64) return BT709HIP_ERR_INVALID_ARG;
  bt709hip_ringset *set = new (std::nothrow) bt709hip_ringset();
  if (set == nullptr) return BT709HIP_ERR_INVALID_ARG;
  set->lanes.resize(static_cast<size_t>(lanes));
  int rc = BT709HIP_OK;
  for (int i = 0; i < lanes && rc == BT709HIP_OK; ++i) {
    bt709hip_ringset::Lane &l = set->lanes[static_cast<size_t>(i)];
    rc = bt709hip_context_create(device_ordinals[i], &l.ctx);
    if (rc == BT709HIP_OK) rc = bt709hip_decoder_create(l.ctx, gamma, has_alpha, &l.dec);
    if (rc == BT709HIP_OK) rc = bt709hip_ring_create_ex(l.dec, width, height, frames, half_scale, tries, options, &l.ring);
  }
  if (rc != BT709HIP_OK) {
    bt709hip_ringset_destroy(set);
    return rc;
  }
  *out = set;
  return BT709HIP_OK;
}

int bt709hip_ringset_lanes(const bt709hip_ringset *set) { return set ? static_cast<int>(set->lanes.size()) : BT709HIP_ERR_INVALID_ARG; }

bt709hip_context *bt709hip_ringset_lane_context(bt709hip_ringset *set, int lane) {
  return set != nullptr && lane >= 0 && static_cast<size_t>(lane) < set->lanes.size() ? set->lanes[static_cast<size_t>(lane)].ctx : nullptr;
}

bt709hip_decoder *bt709hip_ringset_lane_decoder(bt709hip_ringset *set, int lane) {
  return set != nullptr && lane >= 0 && static_cast<size_t>(lane) < set->lanes.size() ? set->lanes[static_cast<size_t>(lane)].dec : nullptr;
}

bt709hip_ring *bt709hip_ringset_lane_ring(bt709hip_ringset *set, int lane) {
  return set != nullptr && lane >= 0 && static_cast<size_t>(lane) < set->lanes.size() ? set->lanes[static_cast<size_t>(lane)].ring : nullptr;
}

int bt709hip_ringset_decode(bt709hip_ringset *set, int first, int count, int wait_until_completed) {
  if (set == nullptr) return BT709HIP_ERR_INVALID_ARG;
  // every lane's launch is enqueued before any lane is waited for: the devices run concurrently
  for (auto &l : set->lanes)
    if (int rc = bt709hip_ring_decode(l.ring, first, count, nullptr, 0)) return rc;
  return wait_until_completed ? bt709hip_ringset_synchronize(set) : BT709HIP_OK;
}

int bt709hip_ringset_synchronize(bt709hip_ringset *set) {
  if (set == nullptr) return BT709HIP_ERR_INVALID_ARG;
  int rc = BT709HIP_OK;
  for (auto &l : set->lanes)
    if (int e = bt709hip_stream_synchronize(l.ctx, nullptr)) rc = rc ? rc : e;
  return rc;
}

}  // extern "C"
