// Lab: does separating reads from writes IN TIME across the whole chip buy HBM throughput for a 27 % read / 73 % write stream?
// (the 1:1 launch's stores alone run at 7.0 TB/s, its loads alone at 6.8 TB/s, mixed at 6.5-6.65.)
// A synthetic persistent kernel, no decode: every workgroup loops { read 8 x 16 B per lane; write 22 x 16 B per lane } over its
// own contiguous chunks.  mode 0: free-running.  mode 1: chip-wide windows by s_memrealtime (10 ns ticks, the same counter
// on every XCD): loads are issued only in [0, R) of every period P, stores only in [R, P).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/phase_lab tools/phase_lab.hip && tools/bin/phase_lab
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512, kLoads = 8, kStores = 22;

__global__ void __launch_bounds__(kThreads) phase_lab(const u32x4 *in, u32x4 *out, uint32_t iters, int mode, uint32_t period, uint32_t rd, int banded) {
  const uint32_t wg = blockIdx.x, nwg = gridDim.x, lane = threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (uint32_t it = 0; it < iters; ++it) {
    // XCD-aware: workgroup w runs on XCD w & 7 (round-robin dispatch); each XCD streams through its own contiguous eighth
    const size_t chunk = banded ? static_cast<size_t>(wg & 7u) * (static_cast<size_t>(iters) * (nwg >> 3)) + static_cast<size_t>(it) * (nwg >> 3) + (wg >> 3)
                                : static_cast<size_t>(it) * nwg + wg;
    const u32x4 *src = in + chunk * (kLoads * kThreads) + lane;
    u32x4 *dst = out + chunk * (kStores * kThreads) + lane;
    if (mode == 1)
      while (static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) % period >= rd) __builtin_amdgcn_s_sleep(1);
    u32x4 v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) v[k] = __builtin_nontemporal_load(src + k * kThreads);
#pragma unroll
    for (int k = 0; k < kLoads; ++k) asm volatile("" : "+v"(v[k]));
    if (mode == 1)
      while (static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) % period < rd) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int k = 0; k < kStores; ++k) {
      u32x4 w = v[k % kLoads];
      w.x ^= static_cast<uint32_t>(k);
      __builtin_nontemporal_store(w, dst + k * kThreads);
    }
    acc ^= v[0];
  }
  if (acc.x == 0x12345678u && acc.y == 0x9e3779b9u) out[0] = acc;  // keeps everything alive
}

// Wave-specialised form: a 1024-lane workgroup, waves 0-7 only LOAD (global -> LDS, chunk k + 1), waves 8-15 only STORE (LDS -> global,
// chunk k), one barrier per step, two 64 KiB LDS buffers: a wave's loads never queue behind stores.
__global__ void __launch_bounds__(1024) phase_lab_split(const u32x4 *in, u32x4 *out, uint32_t iters, int mode, uint32_t period, uint32_t rd, int banded) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  u32x4 *lds = reinterpret_cast<u32x4 *>(lds_raw);  // [2][kLoads][512]
  const uint32_t wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
  const bool loader = tid < 512;
  const uint32_t lane = tid & 511u;
  auto chunk_of = [&](uint32_t it) {
    return banded ? static_cast<size_t>(wg & 7u) * (static_cast<size_t>(iters) * (nwg >> 3)) + static_cast<size_t>(it) * (nwg >> 3) + (wg >> 3)
                  : static_cast<size_t>(it) * nwg + wg;
  };
  for (uint32_t step = 0; step <= iters; ++step) {
    if (loader) {
      if (step < iters) {
        const u32x4 *src = in + chunk_of(step) * (kLoads * 512) + lane;
        if (mode == 1)
          while (static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) % period >= rd) __builtin_amdgcn_s_sleep(1);
        u32x4 v[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) v[k] = __builtin_nontemporal_load(src + k * 512);
#pragma unroll
        for (int k = 0; k < kLoads; ++k) lds[((step & 1u) * kLoads + k) * 512 + lane] = v[k];
      }
    } else if (step > 0) {
      u32x4 *dst = out + chunk_of(step - 1) * (kStores * 512) + lane;
      u32x4 v[kLoads];
#pragma unroll
      for (int k = 0; k < kLoads; ++k) v[k] = lds[(((step - 1) & 1u) * kLoads + k) * 512 + lane];
      if (mode == 1)
        while (static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) % period < rd) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int k = 0; k < kStores; ++k) {
        u32x4 w = v[k % kLoads];
        w.x ^= static_cast<uint32_t>(k);
        __builtin_nontemporal_store(w, dst + k * 512);
      }
    }
    __syncthreads();
  }
}

int main(int argc, char **argv) {
  const int wg_per_cu = argc > 1 ? std::atoi(argv[1]) : 2;
  const int banded = argc > 2 ? std::atoi(argv[2]) : 1;
  const int split = argc > 3 ? std::atoi(argv[3]) : 0;  // 1: the wave-specialised form (one 1024-lane workgroup = 512 loading + 512 storing lanes)
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const uint32_t nwg = static_cast<uint32_t>(prop.multiProcessorCount * wg_per_cu);
  const size_t in_bytes = 3ull << 30, out_bytes = static_cast<size_t>(in_bytes) * kStores / kLoads;
  const size_t in_per_iter = static_cast<size_t>(nwg) * kLoads * kThreads * 16;
  const uint32_t iters = static_cast<uint32_t>(in_bytes / in_per_iter);
  u32x4 *in = nullptr, *out = nullptr;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, out_bytes + 4096));
  CK(hipMemset(in, 1, in_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = static_cast<double>(iters) * in_per_iter * (kLoads + kStores) / kLoads;
  if (split) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_lab_split), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  std::printf("%s%s map, %u workgroups x %d lanes, %u iterations, %.2f GB read + %.2f GB written per launch (%.0f %% reads)\n", split ? "wave-specialised, " : "", banded ? "XCD-aware" : "plain", nwg, kThreads, iters,
              iters * in_per_iter / 1e9, iters * in_per_iter * double(kStores) / kLoads / 1e9, 100.0 * kLoads / (kLoads + kStores));
  struct Cfg {
    int mode;
    uint32_t period, rd;
  };
  // one iteration of the whole grid moves in_per_iter x 3.75 bytes: at ~6.5 TB/s that is the natural period
  const double natural_us = in_per_iter * double(kLoads + kStores) / kLoads / 6.5e12 * 1e6;
  std::printf("natural period of one grid iteration at 6.5 TB/s: %.1f us\n", natural_us);
  std::vector<Cfg> cfgs = {{0, 0, 0}};
  for (double scale : {0.85, 0.92, 1.0, 1.1, 1.25})
    for (double rfrac : {0.22, 0.27, 0.33}) {
      const uint32_t p = static_cast<uint32_t>(natural_us * scale * 100.0);  // ticks of 10 ns
      cfgs.push_back({1, p, static_cast<uint32_t>(p * rfrac)});
    }
  cfgs.push_back({0, 0, 0});
  for (const Cfg &c : cfgs) {
    auto launch = [&]() {
      if (split) hipLaunchKernelGGL(phase_lab_split, dim3(nwg), dim3(1024), 2 * kLoads * 512 * 16, nullptr, in, out, iters, c.mode, c.period ? c.period : 1u, c.rd, banded);
      else hipLaunchKernelGGL(phase_lab, dim3(nwg), dim3(kThreads), 0, nullptr, in, out, iters, c.mode, c.period ? c.period : 1u, c.rd, banded);
    };
    for (int w = 0; w < 2; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double tbps = 3.0 * bytes / (ms * 1e-3) / 1e12;
    if (c.mode == 0) std::printf("free-running                          : %.3f TB/s = %.4f of 8 TB/s\n", tbps, tbps / 8.0);
    else std::printf("windows: period %6.2f us, reads %5.2f us : %.3f TB/s = %.4f of 8 TB/s\n", c.period / 100.0, c.rd / 100.0, tbps, tbps / 8.0);
    std::fflush(stdout);
  }
  return 0;
}
