// VALU issue-rate microbenchmark: how many cycles does one wave64 VALU instruction occupy a
// gfx950 SIMD when plenty of waves are resident?  (decides whether 27 VALU/pixel is 40% or 80%
// of the SIMD's time in the decode kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){std::fprintf(stderr,"%s\n",hipGetErrorString(e_)); std::exit(1);} } while(0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float c) {
  float a[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[4];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 4; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __fadd_rn(a[i], c);          // v_add_f32
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = p[i] + f2{c, c};              // v_pk_add_f32
      } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = (u[i] << 3) + 5u;             // v_lshl_add_u32
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = static_cast<unsigned>(__uint_as_float(u[i] | 0x3f800000u) * c);  // or + mul + cvt
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + u[i];
  for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int instr_per_iter, float *d, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves = 1 wave per SIMD each)
  const int iters = 4096;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_simd = double(instr_per_iter) * iters * waves_per_simd;  // wave-instructions per SIMD
  std::printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name,
              waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
  float *d; CK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4));
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_add_f32", 64, d, w);
    run<1>("v_pk_add_f32", 32, d, w);
    run<2>("v_lshl_add_u32", 64, d, w);
  }
  return 0;
}
