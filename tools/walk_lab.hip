// Lab: the 1:1 fast path as PERSISTENT walking workgroups, against the shipped short-lived kernel (review item 8 of round 2).
// Includes the production kernel file verbatim; the walk kernel below reuses its decode_quad / table / store helpers, so the
// two differ only in how work reaches a wave.  Result: profiles/r03_ab_walk.txt (the walk loses 15-20 % in every form).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 tools/walk_lab.hip \
//         metalbt709decoder_amd/csrc/transfer_tables.cpp -o tools/bin/walk_lab        [-DBT709_WALK_CONTIGUOUS]
//   tools/bin/walk_lab [gamma] [rounds] [workgroups per CU] [lanes per workgroup] [stagger]
// The first walk launch is compared byte for byte with the short-lived kernel's output of the same frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../metalbt709decoder_amd/csrc/bt709_kernels.hip"

namespace bt709 {

// ---------------------------------------------------------------------------
// Fast path, PERSISTENT form (BT709HIP_OPT_QUADS_KERNEL = 1; same preconditions, same bytes out).
// gridDim.x workgroups stay resident and workgroup w walks tile rows w, w + G, w + 2G, ... (a tile row =
// blockDim.x * WALK_QUADS quads of one row pair of one frame; at any moment the chip works on G neighbouring
// tile rows, so the DRAM streams stay address-ordered).  What the walk buys over one short-lived workgroup
// per tile row:
//   * the loads of step n + 1 are issued BEFORE step n's arithmetic and stores, so a wave always has
//     arithmetic of its own to run under its own load latency (in the short-lived form a wave's life is load
//     latency + arithmetic + store drain in series, and only other waves cover it);
//   * the table is staged once per workgroup of the LAUNCH, and there is no barrier in the loop;
//   * every step is straight-line: 3 * WALK_QUADS loads, then 2 * WALK_QUADS stores, so the in-order vmcnt
//     hipcc derives for "loads of step n + 1 have landed" leaves step n's stores in flight.  Lanes past a
//     row's end and steps past the launch's end re-load a valid quad and store its (identical) result again.
// The cursor (tile, row pair, frame) advances by G decomposed on the host: no division in the loop.
// ---------------------------------------------------------------------------
#ifndef BT709_WALK_QUADS
#define BT709_WALK_QUADS 2
#endif
constexpr int kWalkQuads = BT709_WALK_QUADS;

namespace {

struct WalkCursor {
  uint32_t tx, rp, f;
};

__device__ __forceinline__ void walk_advance(WalkCursor &c, const DecodeParams &p, uint32_t row_pairs) {
  c.tx += p.cursor_tx;  // < tiles_x
  c.rp += p.cursor_rp;  // < row_pairs
  c.f += p.cursor_f;
  if (c.tx >= p.tiles_x) {
    c.tx -= p.tiles_x;
    ++c.rp;
  }
  if (c.rp >= row_pairs) {
    c.rp -= row_pairs;
    ++c.f;
  }
}

struct WalkIn {
  uint32_t ya[kWalkQuads], yb[kWalkQuads], cw[kWalkQuads];
};

// The walk's loads are issued by inline asm: hipcc does not see them as vector-memory operations, so it inserts no
// s_waitcnt of its own for them -- its loop-header merge of "what is pending" made every trip wait for loads issued a
// moment earlier (and, through the in-order counter, for the stores in front of them).  The kernel waits explicitly
// (walk_landed) with the exact count.  Safe because (1) every register an asm load writes is consumed only through
// walk_landed's "+v" operands, (2) the only other vector-memory operations in the loop are the step's stores, whose
// count is fixed, and (3) a wait the compiler derives for its own operations can only be stronger than it thinks.
template <bool NT>
__device__ __forceinline__ uint32_t walk_load32(const uint8_t *row, uint32_t byte_offset) {
  uint32_t v;
  if (NT) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(v) : "v"(byte_offset), "s"(row));
  else asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_offset), "s"(row));
  return v;
}

template <bool NT>
__device__ __forceinline__ WalkIn walk_load(const DecodeParams &p, const WalkCursor &c, uint32_t quads) {
  const FramePlanes f = frame_planes(p, c.f);
  const uint8_t *y0 = f.y + static_cast<size_t>(2 * c.rp) * p.y_stride;
  const uint8_t *y1 = y0 + p.y_stride;
  const uint8_t *cc = f.cbcr + static_cast<size_t>(c.rp) * p.cbcr_stride;
  WalkIn in;
#pragma unroll
  for (int u = 0; u < kWalkQuads; ++u) {
    const uint32_t q = min((c.tx * kWalkQuads + u) * blockDim.x + threadIdx.x, quads - 1);
    in.ya[u] = walk_load32<NT>(y0, 4 * q);
    in.yb[u] = walk_load32<NT>(y1, 4 * q);
    in.cw[u] = walk_load32<NT>(cc, 4 * q);
  }
  return in;
}

// waits until at most OUTSTANDING of the wave's youngest vector-memory operations are in flight and hands the set over
template <int OUTSTANDING>
__device__ __forceinline__ void walk_landed(WalkIn &in) {
  static_assert(kWalkQuads == 2, "operand list written for two quads per lane");
  asm volatile("s_waitcnt vmcnt(%6)"
               : "+v"(in.ya[0]), "+v"(in.yb[0]), "+v"(in.cw[0]), "+v"(in.ya[1]), "+v"(in.yb[1]), "+v"(in.cw[1])
               : "n"(OUTSTANDING));
}

}  // namespace

// lab-only launch fields (they lived in the product's DecodeParams until round 4)
struct WalkParams : DecodeParams {
  uint32_t walk_stagger, walk_cus;  // start-up stagger and the CU count
};

template <bool NT>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_quads_walk(const WalkParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t row_pairs = p.height >> 1, quads = p.width >> 2;
  const uint32_t G = gridDim.x;

  // lab knob (BT709HIP_OPT_QUADS_STAGGER): workgroup w starts w * walk_stagger / 256 * 64 cycles late -- a phase ramp over
  // the resident workgroups, so that at any moment loads come from one narrow band of tile rows and stores from another
  // (what in-order dispatch of short-lived workgroups does by itself) instead of from all G rows at once
  for (uint32_t k = (blockIdx.x * p.walk_stagger) >> 8; k > 0; --k) __builtin_amdgcn_s_sleep(1);

#if defined(BT709_WALK_CONTIGUOUS)  // lab: workgroup w walks its OWN block of consecutive tile rows (cursor step 1)
  const uint32_t per_wg = (p.tile_rows + gridDim.x - 1) / gridDim.x;
  uint32_t t = blockIdx.x * per_wg;
  const uint32_t t_end = min(t + per_wg, p.tile_rows);
  if (t >= t_end) return;
#define WALK_STEP 1u
#define WALK_END t_end
#else
  uint32_t t = blockIdx.x;  // < tile_rows (the launcher never starts more workgroups than tile rows)
#define WALK_STEP G
#define WALK_END p.tile_rows
#endif
  WalkCursor pre;
  pre.tx = t % p.tiles_x;
  pre.rp = (t / p.tiles_x) % row_pairs;
  pre.f = (t / p.tiles_x) / row_pairs;
  WalkCursor cur = pre;
  const WalkCursor first = pre;  // always valid: what a load past the launch's end reads instead (result unused)

  // Loads run TWO steps ahead through three explicit register sets.  vmcnt retires loads and stores together, in issue
  // order, so a load issued behind a store waits for that store's acknowledgement -- which under this write-heavy
  // traffic takes far longer than a load.  With the order  L(n+2) | arithmetic(n) | S(n) | wait L(n+1)  the loads a step
  // waits for are OLDER than the previous step's stores: only stores issued two steps earlier sit in front of them.
  // (One step ahead -- L(n+1) | arithmetic(n) | S(n) | wait L(n+1), the loads behind S(n-1) -- ran 15 % slower than the
  // short-lived kernel; tools/ab_walk.sh.)
  WalkIn set[3];
  set[0] = walk_load<NT>(p, pre, quads);
  walk_advance(pre, p, row_pairs);
  {
    const bool more = t + WALK_STEP < WALK_END;
    const WalkCursor c = {more ? pre.tx : first.tx, more ? pre.rp : first.rp, more ? pre.f : first.f};
    set[1] = walk_load<NT>(p, c, quads);
    walk_advance(pre, p, row_pairs);
  }
  stage_table(lds_raw, p.table_unit, p.table_unit_bytes);
  __syncthreads();
  walk_landed<0>(set[0]);  // both sets: everything issued so far
  walk_landed<0>(set[1]);
  const UnitLookup ul = unit_lookup(p.unit_magic, lds_raw);

  while (true) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // step t computes set[k]; its loads go into set[(k + 2) % 3]
      {
        const bool more = t + 2 * WALK_STEP < WALK_END;
        const WalkCursor c = {more ? pre.tx : first.tx, more ? pre.rp : first.rp, more ? pre.f : first.f};
        set[(k + 2) % 3] = walk_load<NT>(p, c, quads);
        walk_advance(pre, p, row_pairs);
      }
      const FramePlanes f = frame_planes(p, cur.f);
      uint8_t *o0 = f.out + static_cast<size_t>(2 * cur.rp) * p.out_stride;
      uint8_t *o1 = o0 + p.out_stride;
#pragma unroll
      for (int u = 0; u < kWalkQuads; ++u) {
        u32x4 top, bot;
        decode_quad<false, false>(ul, set[k].ya[u], set[k].yb[u], set[k].cw[u], 0u, 0u, p.alpha_word, top, bot);
        const uint32_t q = min((cur.tx * kWalkQuads + u) * blockDim.x + threadIdx.x, quads - 1);
        store16<NT>(o0 + 16 * q, top);
        store16<NT>(o1 + 16 * q, bot);
      }
      walk_advance(cur, p, row_pairs);
      t += WALK_STEP;
      if (t >= WALK_END) return;
      // The next step's inputs were issued one step ago, IN FRONT of the previous step's stores.  Behind them in the
      // counter: those stores, this step's loads and this step's stores.
      walk_landed<2 * (2 * kWalkQuads) + 3 * kWalkQuads>(set[(k + 1) % 3]);
    }
  }
}


const char *launch_decode_walk(const DecodeParams &p_in, int frames, bool nontemporal, uint32_t workgroups,
                               uint32_t block_threads, uint32_t compute_units, uint32_t stagger, hipStream_t stream) {
  WalkParams p;
  static_cast<DecodeParams &>(p) = p_in;
  p.walk_cus = compute_units ? compute_units : 256u;
  p.walk_stagger = stagger;
  const uint32_t quads = p.width / 4, row_pairs = p.height / 2;
  const uint32_t per_tile = block_threads * kWalkQuads;
  p.tiles_x = (quads + per_tile - 1) / per_tile;
  const uint64_t total = static_cast<uint64_t>(p.tiles_x) * row_pairs * static_cast<uint32_t>(frames);
  if (total == 0 || total > 0x7fffffffu || block_threads == 0 || block_threads > static_cast<uint32_t>(kMaxBlockThreads)) return nullptr;
  p.tile_rows = static_cast<uint32_t>(total);
  if (workgroups > p.tile_rows) workgroups = p.tile_rows;
  if (workgroups == 0) workgroups = 1;
#if defined(BT709_WALK_CONTIGUOUS)
  const uint32_t cursor_step = 1;
#else
  const uint32_t cursor_step = workgroups;
#endif
  p.cursor_tx = cursor_step % p.tiles_x;
  p.cursor_rp = (cursor_step / p.tiles_x) % row_pairs;
  p.cursor_f = (cursor_step / p.tiles_x) / row_pairs;
  const size_t lds = p.table_unit_bytes;
  if (nontemporal) hipLaunchKernelGGL((decode_nv12_quads_walk<true>), dim3(workgroups), dim3(block_threads), lds, stream, p);
  else hipLaunchKernelGGL((decode_nv12_quads_walk<false>), dim3(workgroups), dim3(block_threads), lds, stream, p);
  return nontemporal ? "decode_nv12_quads_walk<nt>" : "decode_nv12_quads_walk";
}


}  // namespace bt709

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

using namespace bt709;

int main(int argc, char **argv) {
  const int gamma = argc > 1 ? std::atoi(argv[1]) : 0;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
  const uint32_t per_cu = argc > 3 ? std::atoi(argv[3]) : 3;
  const uint32_t lanes = argc > 4 ? std::atoi(argv[4]) : 512;
  const uint32_t stagger = argc > 5 ? std::atoi(argv[5]) : 0;
  const int W = 3840, H = 2160, ring = 64, batch = 32;
  const size_t yb = size_t(W) * H, in_pitch = (yb * 3 / 2 + 255) / 256 * 256, out_pitch = yb * 4;
  uint8_t *d_in, *d_out, *d_ref;
  CK(hipMalloc(&d_in, in_pitch * ring));
  CK(hipMalloc(&d_out, out_pitch * ring));
  CK(hipMalloc(&d_ref, out_pitch * batch));
  {
    std::vector<uint8_t> h(in_pitch * ring);
    uint64_t x = 0x709;
    for (auto &b : h) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      b = uint8_t(x >> 56);
    }
    CK(hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  hipDeviceProp_t props;
  CK(hipGetDeviceProperties(&props, 0));
  CK(prepare_kernels());
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_nv12_quads_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  TransferTable tt;
  if (!build_transfer_table(gamma, &tt)) return 2;
  void *d_table;
  const size_t tub = tt.buckets_unit.size() * sizeof(TransferBucket);
  CK(hipMalloc(&d_table, tub));
  CK(hipMemcpy(d_table, tt.buckets_unit.data(), tub, hipMemcpyHostToDevice));
  auto params = [&](int l, uint8_t *out_base) {
    DecodeParams p;
    std::memset(&p, 0, sizeof p);
    for (int i = 0; i < batch; ++i) {
      uint8_t *base = d_in + size_t(l * batch + i) * in_pitch;
      p.frames[i] = FramePlanes{base, base + yb, nullptr, out_base + size_t(i) * out_pitch};
    }
    p.table_unit = d_table;
    p.table_unit_bytes = uint32_t(tub);
    p.unit_magic = 8388608.0f / float(tt.n);
    p.width = W, p.height = H, p.y_stride = W, p.cbcr_stride = W, p.out_stride = W * 4, p.alpha_word = 0xff000000u;
    return p;
  };
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint32_t cus = uint32_t(props.multiProcessorCount);
  auto short_lived = [&](int l, uint8_t *out) { launch_decode(params(l, out), batch, kVariantQuads, false, false, true, 0, quads_tiles(W), quads_block_threads(W), s); };
  auto walk = [&](int l, uint8_t *out) { launch_decode_walk(params(l, out), batch, true, per_cu * cus, lanes, cus, stagger, s); };
  // parity of the walk against the shipped kernel, launch 0
  short_lived(0, d_ref);
  walk(0, d_out);
  CK(hipStreamSynchronize(s));
  {
    std::vector<uint8_t> a(out_pitch * batch), b(out_pitch * batch);
    CK(hipMemcpy(a.data(), d_ref, a.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_out, b.size(), hipMemcpyDeviceToHost));
    std::printf("walk == short-lived on %d frames: %s\n", batch, a == b ? "yes" : "NO");
    if (a != b) return 3;
  }
  auto once = [&](bool w, int reps) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r)
      for (int l = 0; l < ring / batch; ++l) w ? walk(l, d_out + size_t(l) * batch * out_pitch) : short_lived(l, d_out + size_t(l) * batch * out_pitch);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return double(ms) / (reps * (ring / batch));
  };
  for (int i = 0; i < 3; ++i) once(false, 20), once(true, 20);
  std::vector<double> a, b;
  for (int k = 0; k < rounds; ++k) a.push_back(once(false, 50)), b.push_back(once(true, 50));
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  const double bytes = double(size_t(W) * H * 11 / 2) * batch;
  std::printf("short-lived                              median %8.2f us %7.1f GB/s\n", a[a.size() / 2] * 1e3, bytes / a[a.size() / 2] / 1e6);
  std::printf("walk %u per CU x %u lanes, stagger %-4u     median %8.2f us %7.1f GB/s\n", per_cu, lanes, stagger, b[b.size() / 2] * 1e3, bytes / b[b.size() / 2] / 1e6);
  return 0;
}
