// LDS gather cost on gfx950, per wave-instruction, with every CU busy (review item 3 (iii) of round 2: what would a
// 64 KiB (Y, C) -> byte table cost the persistent 2:1 kernel?).  One 1024-thread workgroup per CU (the persistent kernel's
// shape: 4 waves per SIMD), each lane issues independent reads at pseudo-random addresses (an LCG per lane, the address
// arithmetic is 2 VALU instructions per read so the LDS pipe, not the VALU, is what fills), 8 reads in flight per wait.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_gather_probe.hip -o tools/bin/lds_gather_probe
// Prints LDS cycles per wave-instruction = time x clock x CUs / wave-instructions (clock from hipDeviceProp).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const uint8_t *LdsByte;
typedef __attribute__((address_space(3))) const uint32_t *LdsWord;
typedef __attribute__((address_space(3))) const u32x4 *LdsQuad;

enum Mode : int {
  kU8Random64K = 0,   // ds_read_u8, uniformly random over 64 KiB: the (Y, C) byte table
  kU8Random256 = 1,   // ds_read_u8 inside one 256-byte row (one chroma value for the whole wave)
  kB32Table8 = 2,     // ds_read_b32 from a 256-entry table in 8 interleaved copies (lane & 7): lin[byte]
  kB32Table1 = 3,     // the same table, one copy
  kB128Copies16 = 4,  // ds_read_b128, 512 entries x 16 interleaved copies (lane & 15): the shipped decode-side lookup
  kB128Copies1 = 5,   // the same entries, one copy (8 KiB)
  kB64Table = 6,      // ds_read_b64, 3008 entries, one copy: the shipped encode-side lookup
};

template <int MODE>
__global__ void __launch_bounds__(1024) probe(uint32_t *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (uint32_t i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
  __syncthreads();
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)lds));
  // 8 pseudo-random addresses per lane, fixed for the run: the cost of a gather depends on the address pattern of the
  // instruction, not on its changing from trip to trip, and with the addresses out of the loop the loop body is 8 reads
  // + 8 v_xor (the VALU has 4x headroom: the LDS pipe is what fills)
  uint32_t state = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u, acc = 0;
  uint32_t a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    state = state * 1664525u + 1013904223u;
    const uint32_t r = state >> 8;
    if (MODE == kU8Random64K) a[k] = base + (r & 0xffffu);
    else if (MODE == kU8Random256) a[k] = base + (r & 0xffu) + 0x3300u;
    else if (MODE == kB32Table8) a[k] = base + ((r & 0xffu) << 5) + ((threadIdx.x & 7u) << 2);
    else if (MODE == kB32Table1) a[k] = base + ((r & 0xffu) << 2);
    else if (MODE == kB128Copies16) a[k] = base + ((r & 0x1ffu) << 8) + ((threadIdx.x & 15u) << 4);
    else if (MODE == kB128Copies1) a[k] = base + ((r & 0x1ffu) << 4);
    else a[k] = base + ((r % 3008u) << 3);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(a[k]));  // keep the reads inside the loop
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE <= kU8Random256) v[k] = *reinterpret_cast<LdsByte>(a[k]);
      else if (MODE <= kB32Table1) v[k] = *reinterpret_cast<LdsWord>(a[k]);
      else if (MODE <= kB128Copies1) {
        const u32x4 q = *reinterpret_cast<LdsQuad>(a[k]);
        v[k] = q.x ^ q.y ^ q.z ^ q.w;
      } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 q = *reinterpret_cast<__attribute__((address_space(3))) const u32x2 *>(a[k]);
        v[k] = q.x ^ q.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char *name, int cus, double clock_hz, uint32_t *d_out) {
  const int iters = 2000;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(1024), 160 * 1024, 0, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double wave_instr_per_cu = 16.0 * iters * 8;  // 16 waves per CU
  std::printf("%-62s %6.2f LDS cycles per wave-instruction (%.3f ms)\n", name, best * 1e-3 * clock_hz / wave_instr_per_cu, best);
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double clock_hz = p.clockRate * 1e3;
  uint32_t *d_out;
  CK(hipMalloc(&d_out, size_t(cus) * 1024 * 4));
  std::printf("%s, %d CUs, %.2f GHz nominal; one 1024-thread workgroup per CU, 8 reads in flight per wait\n", p.gcnArchName, cus, clock_hz / 1e9);
  run<kU8Random64K>("ds_read_u8, random over 64 KiB  [(Y, C) byte table]", cus, clock_hz, d_out);
  run<kU8Random256>("ds_read_u8, random inside one 256-byte row", cus, clock_hz, d_out);
  run<kB32Table8>("ds_read_b32, 256 entries x 8 copies  [lin[byte]]", cus, clock_hz, d_out);
  run<kB32Table1>("ds_read_b32, 256 entries, one copy", cus, clock_hz, d_out);
  run<kB128Copies16>("ds_read_b128, 512 entries x 16 copies  [shipped decode side]", cus, clock_hz, d_out);
  run<kB128Copies1>("ds_read_b128, 512 entries, one copy", cus, clock_hz, d_out);
  run<kB64Table>("ds_read_b64, 3008 entries, one copy  [shipped encode side]", cus, clock_hz, d_out);
  return 0;
}
