// Decode lab: times the PRODUCTION kernels (csrc/bt709_kernels.hip included verbatim) over a
// ring of 64 distinct 4K frames, several launch shapes, interleaved rounds (median / min).
// Build variants of the shipping kernels: -DBT709_MAX_BLOCK_THREADS / -DBT709_QUADS_PER_LANE (tile shape); with
// -DBT709_LAB_SRC (after `python tools/lab_variants.py --sources`) also -DBT709_INDEX_RTZ (round 1's floor index between
// two s_setreg) and -DBT709_NO_FMA_CENTRE: those gates left csrc/ in round 4.  With
// -DBT709_LAB_VARIANTS the lab copy tools/lab_quads_variants.hip is timed instead, whose own macros
// (-DBT709_LAB_NO_LDS, _LDS_CHROMA, _ADJACENT, _LDS_PAD) select round 1's rejected variants.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/decode_lab.hip \
//         metalbt709decoder_amd/csrc/transfer_tables.cpp -o tools/bin/decode_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#if defined(BT709_LAB_VARIANTS)  // round 1's kernel file with the rejected variants (see its header)
#include "lab_quads_variants.hip"
#else
#if defined(BT709_LAB_SRC)  // python tools/lab_variants.py --sources: the product kernels with the experiment gates re-inserted
#include "bin/lab_src/bt709_kernels.hip"
#else
#include "../metalbt709decoder_amd/csrc/bt709_kernels.hip"
#endif
#endif
#if defined(BT709_LAB_LDS_CHROMA)
#define LAB_EXTRA_LDS 4096
#else
#define LAB_EXTRA_LDS 0
#endif

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

using namespace bt709;

struct Ring {
  int W, H, ring, batch;
  size_t yb, cb, ob, in_stride, out_stride;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  void *d_table_unit = nullptr;
  TransferTable tt;
  std::vector<DecodeParams> params;
  hipStream_t s, s2 = nullptr;  // s2: second stream when env LAB_STREAMS=2 (launch l goes to stream l & 1)
  hipEvent_t e0, e1, e2;

  Ring(int w, int h, int ring_, int batch_, int gamma) : W(w), H(h), ring(ring_), batch(batch_) {
    // placement experiments: LAB_ROW_PAD bytes added to every output row, LAB_IN_ROW_PAD to every input row,
    // LAB_FRAME_PAD to the frame pitches
    const size_t row_pad = std::getenv("LAB_ROW_PAD") ? std::atoi(std::getenv("LAB_ROW_PAD")) : 0;
    const size_t in_row_pad = std::getenv("LAB_IN_ROW_PAD") ? std::atoi(std::getenv("LAB_IN_ROW_PAD")) : 0;
    const size_t frame_pad = std::getenv("LAB_FRAME_PAD") ? std::atoi(std::getenv("LAB_FRAME_PAD")) : 0;
    const size_t ys = W + in_row_pad, os = size_t(W) * 4 + row_pad;
    yb = ys * H;
    cb = yb / 2;
    ob = os * H;
    in_stride = (yb + cb + 255) / 256 * 256 + frame_pad;
    out_stride = ob + frame_pad;
    // LAB_ALLOC: 0 hipMalloc (default) | 1 output slab uncached | 2 both slabs uncached | 3 output fine-grained
    const int alloc = std::getenv("LAB_ALLOC") ? std::atoi(std::getenv("LAB_ALLOC")) : 0;
    //            4 one allocation for both slabs (in, then out at a 2 MiB boundary) | 5 out slab allocated before the in slab
    if (alloc == 4) {
      const size_t gap = std::getenv("LAB_GAP_KB") ? size_t(std::atoll(std::getenv("LAB_GAP_KB"))) << 10 : 0;
      const size_t in_bytes = (in_stride * ring + (2u << 20) - 1) / (2u << 20) * (2u << 20) + gap;
      CK(hipMalloc(&d_in, in_bytes + out_stride * ring));
      d_out = d_in + in_bytes;
    } else if (alloc == 5) {
      CK(hipMalloc(&d_out, out_stride * ring));
      CK(hipMalloc(&d_in, in_stride * ring));
    } else
    if (alloc == 2) CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&d_in), in_stride * ring, hipDeviceMallocUncached));
    else CK(hipMalloc(&d_in, in_stride * ring));
    if (alloc == 4 || alloc == 5) {
    } else
    if (alloc == 1 || alloc == 2) CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&d_out), out_stride * ring, hipDeviceMallocUncached));
    else if (alloc == 3) CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&d_out), out_stride * ring, hipDeviceMallocFinegrained));
    else CK(hipMalloc(&d_out, out_stride * ring));
    std::vector<uint8_t> h8(in_stride);
    uint64_t st = 0x709;
    for (int i = 0; i < ring; ++i) {
      for (size_t j = 0; j + 8 <= h8.size(); j += 8) {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        std::memcpy(&h8[j], &z, 8);
      }
      CK(hipMemcpy(d_in + i * in_stride, h8.data(), h8.size(), hipMemcpyHostToDevice));
    }
    std::printf("slabs: in %p  out %p  (out - in) = %lld KiB, in %% 1GiB = %llu KiB, out %% 1GiB = %llu KiB\n", (void *)d_in, (void *)d_out,
                (long long)((intptr_t)d_out - (intptr_t)d_in) / 1024, (unsigned long long)((uintptr_t)d_in % (1ull << 30)) / 1024,
                (unsigned long long)((uintptr_t)d_out % (1ull << 30)) / 1024);
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    if (std::getenv("LAB_STREAMS") && std::atoi(std::getenv("LAB_STREAMS")) == 2) CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    if (!build_transfer_table(gamma, &tt)) std::exit(2);
    const size_t tub = tt.buckets_unit.size() * sizeof(TransferBucket);
    CK(hipMalloc(&d_table_unit, tub));
    CK(hipMemcpy(d_table_unit, tt.buckets_unit.data(), tub, hipMemcpyHostToDevice));
    params.resize(ring / batch);
    for (int l = 0; l < ring / batch; ++l) {
      DecodeParams &p = params[l];
      std::memset(&p, 0, sizeof p);
      for (int i = 0; i < batch; ++i) {
        // LAB_SAME_INPUT=1: every frame reads ring frame 0 (the input stays in L2 / MALL): what the kernel does when its loads are short
        uint8_t *base = d_in + (std::getenv("LAB_SAME_INPUT") ? 0 : size_t(l * batch + i) * in_stride);
        // alpha: the luma plane of the ring frame half a ring away (bytes no workgroup of this launch reads): what profiles/r02_decode_lab_ceiling.txt's
        // extra-loads run read through a lab macro of that commit
        p.frames[i] = FramePlanes{base, base + yb, d_in + size_t((l * batch + i + ring / 2) % ring) * in_stride, d_out + size_t(l * batch + i) * out_stride};
      }
      p.table_unit = d_table_unit;
      p.table_unit_bytes = uint32_t(tub);
      p.unit_magic = 8388608.0f / float(tt.n);
      p.width = W;
      p.height = H;
      p.y_stride = uint32_t(ys);
      p.cbcr_stride = uint32_t(ys);
      p.out_stride = uint32_t(os);
      p.alpha_word = 0xff000000u;
    }
  }
  double bytes_per_launch() const { return double(size_t(W) * H * 11 / 2) * batch; }  // algorithmic, whatever the padding
  double once(const std::function<void(int)> &fn, int reps) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r)
      for (size_t l = 0; l < params.size(); ++l) fn(int(l));
    if (s2) {  // LAB_STREAMS=2: the odd launches went to s2; e1 must follow both streams
      CK(hipEventRecord(e2, s2));
      CK(hipStreamWaitEvent(s, e2, 0));
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / (reps * params.size());
  }
};

struct Variant {
  std::string name;
  std::function<void(int)> fn;
  std::vector<double> ms;
};

int main(int argc, char **argv) {
  const int gamma = argc > 1 ? std::atoi(argv[1]) : 0;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
  const int batch = argc > 3 ? std::atoi(argv[3]) : 32;  // frames per launch (1: the config-5 unit at 8 GPUs)
  Ring r(3840, 2160, 64, batch, gamma);
  hipStream_t s = r.s;
  std::vector<Variant> vs;
  auto add = [&](const std::string &n, std::function<void(int)> f) { vs.push_back({n, f, {}}); };
  // tile shapes that cover a 960-quad row pair exactly with kQuadsPerLane quads per lane
  for (int tiles : {1, 2, 3, 4}) {
    const int per_tile = (960 + tiles - 1) / tiles;
    int t = ((per_tile + kQuadsPerLane - 1) / kQuadsPerLane + 63) / 64 * 64;
    if (t > kMaxBlockThreads) continue;
    add("quads<nt> qpl=" + std::to_string(kQuadsPerLane) + " tiles=" + std::to_string(tiles) + " threads=" + std::to_string(t),
        [&r, s, t, tiles](int l) { launch_decode(r.params[l], r.batch, kVariantQuads, false, false, true, std::getenv("LAB_XCD_BANDS") ? std::atoi(std::getenv("LAB_XCD_BANDS")) : 0, tiles, t, (r.s2 && (l & 1)) ? r.s2 : s); });
  }

  // warm the clocks
  for (int i = 0; i < 3; ++i)
    for (auto &v : vs) r.once(v.fn, 20);
  for (int k = 0; k < rounds; ++k)
    for (auto &v : vs) v.ms.push_back(r.once(v.fn, 50));
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const double med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    std::printf("%-48s median %8.2f us %7.1f GB/s (%.3f)   best %8.2f us %7.1f GB/s\n", v.name.c_str(), med * 1e3,
                r.bytes_per_launch() / med / 1e6, r.bytes_per_launch() / med / 1e6 / 8000.0, mn * 1e3,
                r.bytes_per_launch() / mn / 1e6);
  }
  return 0;
}
