// Kernel lab: standalone timing harness for access-pattern and compute experiments on
// the GPU box.  Not part of the product; results that matter are recorded in DESIGN.md.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/kernel_lab.hip \
//         metalbt709decoder_amd/csrc/transfer_tables.cpp -o gpurun_out/kernel_lab && gpurun_out/kernel_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../metalbt709decoder_amd/csrc/bt709_kernels.hip"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                 \
    }                                                                               \
  } while (0)

using namespace bt709;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int W = 3840, H = 2160, RING = 64, BATCH = 32;
constexpr size_t YB = size_t(W) * H, CB = size_t(W) * H / 2, OB = size_t(W) * H * 4;
constexpr size_t IN_STRIDE = (YB + CB + 255) / 256 * 256, OUT_STRIDE = OB;

// ---- plain streaming kernels (ceilings) ----------------------------------------------------
__global__ void k_copy16(const u32x4 *__restrict__ a, u32x4 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    __builtin_nontemporal_store(a[i], &b[i]);
}
__global__ void k_fill16(u32x4 *__restrict__ b, size_t n, uint32_t v) {
  u32x4 x = {v, v + 1, v + 2, v + 3};
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    __builtin_nontemporal_store(x, &b[i]);
}
__global__ void k_fill16_plain(u32x4 *__restrict__ b, size_t n, uint32_t v) {
  u32x4 x = {v, v + 1, v + 2, v + 3};
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) b[i] = x;
}
__global__ void k_read16(const u32x4 *__restrict__ a, size_t n, uint32_t *sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc ^= a[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// one-shot (no loop) variants: each lane moves K consecutive-by-wave 16-byte pieces
template <int K, int MODE>  // MODE 0 plain, 1 nt
__global__ void __launch_bounds__(256) k_copy_once(const u32x4 *__restrict__ a, u32x4 *__restrict__ b, size_t n) {
  const size_t base = (size_t(blockIdx.x) * K) * 256 + threadIdx.x;
  u32x4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) if (base + k * 256 < n) v[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (base + k * 256 < n) {
      if (MODE == 1) __builtin_nontemporal_store(v[k], &b[base + k * 256]);
      else b[base + k * 256] = v[k];
    }
}
template <int K, int MODE>
__global__ void __launch_bounds__(256) k_fill_once(u32x4 *__restrict__ b, size_t n, uint32_t x) {
  const size_t base = (size_t(blockIdx.x) * K) * 256 + threadIdx.x;
  u32x4 v = {x, x + 1, x + 2, x + 3};
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (base + k * 256 < n) {
      if (MODE == 1) __builtin_nontemporal_store(v, &b[base + k * 256]);
      else b[base + k * 256] = v;
    }
}
// 4:1 write:read mix on flat arrays: read 1 piece, write 4 pieces (decode is 1.5:4)
template <int MODE>
__global__ void __launch_bounds__(256) k_mix_once(const u32x4 *__restrict__ a, u32x4 *__restrict__ b, size_t n_in) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_in) return;
  u32x4 v = a[i];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    u32x4 w = v + uint32_t(k);
    const size_t o = (size_t(blockIdx.x) * 4 + k) * 256 + threadIdx.x;
    if (MODE == 1) __builtin_nontemporal_store(w, &b[o]);
    else b[o] = w;
  }
}

// ---- decode-shaped traffic, trivial arithmetic: same loads and stores as decode_nv12_quads --
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_shape_quads(const DecodeParams p) {
  const FramePlanes f = p.frames[blockIdx.y];
  const uint32_t quads = p.width >> 2, row_pairs = p.height >> 1;
  for (uint32_t rp = blockIdx.x; rp < row_pairs; rp += gridDim.x) {
    const uint8_t *y0 = f.y + size_t(2 * rp) * p.y_stride, *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + size_t(rp) * p.cbcr_stride;
    uint8_t *o0 = f.out + size_t(2 * rp) * p.out_stride, *o1 = o0 + p.out_stride;
    for (uint32_t q0 = 0; q0 < quads; q0 += 256 * UNROLL) {
      uint32_t ya[UNROLL], yb[UNROLL], cw[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t q = q0 + u * 256 + threadIdx.x;
        if (q < quads) {
          ya[u] = *reinterpret_cast<const uint32_t *>(y0 + 4 * q);
          yb[u] = *reinterpret_cast<const uint32_t *>(y1 + 4 * q);
          cw[u] = *reinterpret_cast<const uint32_t *>(cc + 4 * q);
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t q = q0 + u * 256 + threadIdx.x;
        if (q < quads) {
          u32x4 t = {ya[u], ya[u] ^ cw[u], ya[u] + cw[u], ya[u] | 0xff000000u};
          u32x4 b = {yb[u], yb[u] ^ cw[u], yb[u] + cw[u], yb[u] | 0xff000000u};
          if (NT) {
            __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(o0 + 16 * q));
            __builtin_nontemporal_store(b, reinterpret_cast<u32x4 *>(o1 + 16 * q));
          } else {
            *reinterpret_cast<u32x4 *>(o0 + 16 * q) = t;
            *reinterpret_cast<u32x4 *>(o1 + 16 * q) = b;
          }
        }
      }
    }
  }
}

// store flavours: 0 plain, 1 nt, 2 sc0 sc1 (write-through, not kept in L2), 3 sc1
template <int MODE>
__device__ __forceinline__ void st16(uint8_t *p, u32x4 v) {
  if (MODE == 0) *reinterpret_cast<u32x4 *>(p) = v;
  else if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// generalised shape kernel: THREADS per block, each block covers CHUNKS*THREADS quads of ONE row pair;
// grid.x = row_pairs * ceil(quads / (CHUNKS*THREADS)); loads first, then stores.
template <int THREADS, int CHUNKS, int MODE>
__global__ void __launch_bounds__(THREADS) k_shape_gen(const DecodeParams p) {
  const FramePlanes f = p.frames[blockIdx.y];
  const uint32_t quads = p.width >> 2;
  const uint32_t per_rp = (quads + CHUNKS * THREADS - 1) / (CHUNKS * THREADS);
  const uint32_t rp = blockIdx.x / per_rp, part = blockIdx.x % per_rp;
  const uint8_t *y0 = f.y + size_t(2 * rp) * p.y_stride, *y1 = y0 + p.y_stride;
  const uint8_t *cc = f.cbcr + size_t(rp) * p.cbcr_stride;
  uint8_t *o0 = f.out + size_t(2 * rp) * p.out_stride, *o1 = o0 + p.out_stride;
  uint32_t ya[CHUNKS], yb[CHUNKS], cw[CHUNKS];
#pragma unroll
  for (int u = 0; u < CHUNKS; ++u) {
    const uint32_t q = (part * CHUNKS + u) * THREADS + threadIdx.x;
    if (q < quads) {
      ya[u] = *reinterpret_cast<const uint32_t *>(y0 + 4 * q);
      yb[u] = *reinterpret_cast<const uint32_t *>(y1 + 4 * q);
      cw[u] = *reinterpret_cast<const uint32_t *>(cc + 4 * q);
    }
  }
#pragma unroll
  for (int u = 0; u < CHUNKS; ++u) {
    const uint32_t q = (part * CHUNKS + u) * THREADS + threadIdx.x;
    if (q < quads) {
      u32x4 t = {ya[u], ya[u] ^ cw[u], ya[u] + cw[u], ya[u] | 0xff000000u};
      u32x4 b = {yb[u], yb[u] ^ cw[u], yb[u] + cw[u], yb[u] | 0xff000000u};
      st16<MODE>(o0 + 16 * q, t);
      st16<MODE>(o1 + 16 * q, b);
    }
  }
}

// ---- flat variant of the same traffic: one quad-pair per lane, grid covers the frame, no loop --
template <bool NT>
__global__ void __launch_bounds__(256) k_shape_flat(const DecodeParams p) {
  const FramePlanes f = p.frames[blockIdx.z];
  const uint32_t q = blockIdx.x * 256 + threadIdx.x, rp = blockIdx.y;
  if (q >= (p.width >> 2)) return;
  const uint8_t *y0 = f.y + size_t(2 * rp) * p.y_stride;
  uint32_t ya = *reinterpret_cast<const uint32_t *>(y0 + 4 * q);
  uint32_t yb = *reinterpret_cast<const uint32_t *>(y0 + p.y_stride + 4 * q);
  uint32_t cw = *reinterpret_cast<const uint32_t *>(f.cbcr + size_t(rp) * p.cbcr_stride + 4 * q);
  uint8_t *o0 = f.out + size_t(2 * rp) * p.out_stride;
  u32x4 t = {ya, ya ^ cw, ya + cw, ya | 0xff000000u};
  u32x4 b = {yb, yb ^ cw, yb + cw, yb | 0xff000000u};
  if (NT) {
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(o0 + 16 * q));
    __builtin_nontemporal_store(b, reinterpret_cast<u32x4 *>(o0 + p.out_stride + 16 * q));
  } else {
    *reinterpret_cast<u32x4 *>(o0 + 16 * q) = t;
    *reinterpret_cast<u32x4 *>(o0 + p.out_stride + 16 * q) = b;
  }
}

struct Lab {
  uint8_t *d_in = nullptr, *d_out = nullptr;
  hipEvent_t e0, e1;
  hipStream_t s;
  DecodeParams params[RING / BATCH];
  void *d_table = nullptr;
  TransferTable tt;

  void init(int gamma) {
    CK(hipMalloc(&d_in, IN_STRIDE * RING));
    CK(hipMalloc(&d_out, OUT_STRIDE * RING));
    std::vector<uint8_t> h(IN_STRIDE);
    uint64_t st = 0x709;
    for (int i = 0; i < RING; ++i) {
      for (size_t j = 0; j < h.size(); j += 8) {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        std::memcpy(&h[j], &z, 8);
      }
      CK(hipMemcpy(d_in + i * IN_STRIDE, h.data(), h.size(), hipMemcpyHostToDevice));
    }
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!build_transfer_table(gamma, &tt)) std::exit(2);
    const size_t tb = tt.buckets_unit.size() * sizeof(TransferBucket);
    CK(hipMalloc(&d_table, tb));
    CK(hipMemcpy(d_table, tt.buckets_unit.data(), tb, hipMemcpyHostToDevice));
    for (int l = 0; l < RING / BATCH; ++l) {
      DecodeParams &p = params[l];
      std::memset(&p, 0, sizeof p);
      for (int i = 0; i < BATCH; ++i) {
        uint8_t *base = d_in + size_t(l * BATCH + i) * IN_STRIDE;
        p.frames[i] = FramePlanes{base, base + YB, nullptr, d_out + size_t(l * BATCH + i) * OUT_STRIDE};
      }
      p.table_unit = d_table;
      p.table_unit_bytes = uint32_t(tb);
      p.unit_magic = 8388608.0f / float(tt.n);
      p.width = W;
      p.height = H;
      p.y_stride = W;
      p.cbcr_stride = W;
      p.out_stride = W * 4;
      p.alpha_word = 0xff000000u;
    }
  }

  // run `fn(launch_index)` for every launch of the ring, `reps` times; returns avg ms per launch
  double time(const std::function<void(int)> &fn, int reps = 20) {
    for (int l = 0; l < RING / BATCH; ++l) fn(l);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r)
      for (int l = 0; l < RING / BATCH; ++l) fn(l);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / (reps * (RING / BATCH));
  }
};

static void report(const char *name, double ms, double bytes) {
  std::printf("%-44s %9.3f us/launch  %8.1f GB/s  (%5.1f%% of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e6,
              bytes / ms / 1e6 / 80.0);
  std::fflush(stdout);
}

int main(int argc, char **argv) {
  int gamma = argc > 1 ? std::atoi(argv[1]) : 0;
  Lab lab;
  lab.init(gamma);
  const double dec_bytes = double(YB + CB + OB) * BATCH;
  hipStream_t s = lab.s;

  {  // ceilings: stream over the whole in/out slabs (half ring per "launch" to mimic the footprint)
    const size_t n_in = IN_STRIDE * BATCH / 16, n_out = OUT_STRIDE * BATCH / 16;
    for (int blocks : {2048, 8192}) {
      char nm[96];
      std::snprintf(nm, sizeof nm, "copy16 nt (in-slab -> out-slab) grid=%d", blocks);
      report(nm, lab.time([&](int l) {
               hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, s,
                                  reinterpret_cast<const u32x4 *>(lab.d_in + size_t(l) * BATCH * IN_STRIDE),
                                  reinterpret_cast<u32x4 *>(lab.d_out + size_t(l) * BATCH * OUT_STRIDE), n_in);
             }),
             2.0 * n_in * 16);
      std::snprintf(nm, sizeof nm, "fill16 nt (out-slab) grid=%d", blocks);
      report(nm, lab.time([&](int l) {
               hipLaunchKernelGGL(k_fill16, dim3(blocks), dim3(256), 0, s,
                                  reinterpret_cast<u32x4 *>(lab.d_out + size_t(l) * BATCH * OUT_STRIDE), n_out, 7u);
             }),
             1.0 * n_out * 16);
      std::snprintf(nm, sizeof nm, "fill16 plain (out-slab) grid=%d", blocks);
      report(nm, lab.time([&](int l) {
               hipLaunchKernelGGL(k_fill16_plain, dim3(blocks), dim3(256), 0, s,
                                  reinterpret_cast<u32x4 *>(lab.d_out + size_t(l) * BATCH * OUT_STRIDE), n_out, 7u);
             }),
             1.0 * n_out * 16);
      std::snprintf(nm, sizeof nm, "read16 (out-slab) grid=%d", blocks);
      report(nm, lab.time([&](int l) {
               hipLaunchKernelGGL(k_read16, dim3(blocks), dim3(256), 0, s,
                                  reinterpret_cast<const u32x4 *>(lab.d_out + size_t(l) * BATCH * OUT_STRIDE), n_out,
                                  reinterpret_cast<uint32_t *>(lab.d_table));
             }),
             1.0 * n_out * 16);
    }
  }

  {
    const size_t n_in = IN_STRIDE * BATCH / 16, n_out = OUT_STRIDE * BATCH / 16;
    auto in = [&](int l) { return reinterpret_cast<const u32x4 *>(lab.d_in + size_t(l) * BATCH * IN_STRIDE); };
    auto out = [&](int l) { return reinterpret_cast<u32x4 *>(lab.d_out + size_t(l) * BATCH * OUT_STRIDE); };
#define ONCE(K, MODE, KERN, NAME, N, BYTES, ...)                                                              \
  report(NAME, lab.time([&](int l) {                                                                          \
           hipLaunchKernelGGL((KERN<K, MODE>), dim3(unsigned(((N) + 256 * K - 1) / (256 * K))), dim3(256), 0, s, \
                              __VA_ARGS__);                                                                   \
         }),                                                                                                  \
         BYTES)
    ONCE(1, 0, k_copy_once, "copy_once K=1 plain", n_in, 2.0 * n_in * 16, in(l), out(l), n_in);
    ONCE(1, 1, k_copy_once, "copy_once K=1 nt", n_in, 2.0 * n_in * 16, in(l), out(l), n_in);
    ONCE(4, 0, k_copy_once, "copy_once K=4 plain", n_in, 2.0 * n_in * 16, in(l), out(l), n_in);
    ONCE(4, 1, k_copy_once, "copy_once K=4 nt", n_in, 2.0 * n_in * 16, in(l), out(l), n_in);
    ONCE(8, 1, k_copy_once, "copy_once K=8 nt", n_in, 2.0 * n_in * 16, in(l), out(l), n_in);
    ONCE(1, 0, k_fill_once, "fill_once K=1 plain", n_out, 1.0 * n_out * 16, out(l), n_out, 3u);
    ONCE(1, 1, k_fill_once, "fill_once K=1 nt", n_out, 1.0 * n_out * 16, out(l), n_out, 3u);
    ONCE(4, 0, k_fill_once, "fill_once K=4 plain", n_out, 1.0 * n_out * 16, out(l), n_out, 3u);
    ONCE(4, 1, k_fill_once, "fill_once K=4 nt", n_out, 1.0 * n_out * 16, out(l), n_out, 3u);
    ONCE(8, 1, k_fill_once, "fill_once K=8 nt", n_out, 1.0 * n_out * 16, out(l), n_out, 3u);
    const size_t n_mix = n_out / 4;
    report("mix_once 1r:4w plain", lab.time([&](int l) { hipLaunchKernelGGL((k_mix_once<0>), dim3(unsigned((n_mix + 255) / 256)), dim3(256), 0, s, in(l), out(l), n_mix); }), 5.0 * n_mix * 16);
    report("mix_once 1r:4w nt", lab.time([&](int l) { hipLaunchKernelGGL((k_mix_once<1>), dim3(unsigned((n_mix + 255) / 256)), dim3(256), 0, s, in(l), out(l), n_mix); }), 5.0 * n_mix * 16);
  }

#define GEN(T, C, M)                                                                                           \
  {                                                                                                            \
    const unsigned per_rp = (960 + (C) * (T)-1) / ((C) * (T));                                                  \
    report("shape_gen<T=" #T ",C=" #C ",M=" #M ">",                                                             \
           lab.time([&](int l) { hipLaunchKernelGGL((k_shape_gen<T, C, M>), dim3(1080 * per_rp, BATCH), dim3(T), 0, s, lab.params[l]); }), \
           dec_bytes);                                                                                         \
  }
  GEN(256, 4, 1) GEN(256, 4, 0) GEN(256, 4, 2) GEN(256, 4, 3)
  GEN(256, 2, 1) GEN(256, 1, 1) GEN(128, 4, 1) GEN(128, 8, 1) GEN(64, 15, 1) GEN(64, 5, 1) GEN(64, 3, 1)
  GEN(192, 5, 1) GEN(320, 3, 1) GEN(480, 2, 1) GEN(960, 1, 1) GEN(512, 2, 1) GEN(1024, 1, 1)
  GEN(320, 3, 2) GEN(960, 1, 2) GEN(192, 5, 0)

  for (int gx : {64, 128, 256, 540, 1080}) {
    char nm[96];
    std::snprintf(nm, sizeof nm, "shape_quads<4,nt>  grid.x=%d", gx);
    report(nm, lab.time([&](int l) { hipLaunchKernelGGL((k_shape_quads<4, true>), dim3(gx, BATCH), dim3(256), 0, s, lab.params[l]); }), dec_bytes);
    std::snprintf(nm, sizeof nm, "shape_quads<4,plain> grid.x=%d", gx);
    report(nm, lab.time([&](int l) { hipLaunchKernelGGL((k_shape_quads<4, false>), dim3(gx, BATCH), dim3(256), 0, s, lab.params[l]); }), dec_bytes);
  }
  report("shape_flat<nt>  (4,1080,32)", lab.time([&](int l) { hipLaunchKernelGGL((k_shape_flat<true>), dim3(4, 1080, BATCH), dim3(256), 0, s, lab.params[l]); }), dec_bytes);
  report("shape_flat<plain> (4,1080,32)", lab.time([&](int l) { hipLaunchKernelGGL((k_shape_flat<false>), dim3(4, 1080, BATCH), dim3(256), 0, s, lab.params[l]); }), dec_bytes);

  // the production kernels are timed by tools/decode_lab.hip
  return 0;
}
