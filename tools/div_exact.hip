// Is  q = fma(fma(-c, x*rc, x), rc, x*rc)  (rc = fl(1/c))  equal to the correctly rounded x / c
// for EVERY float x?  Exhaustive over all 2^32 bit patterns, for the encoder's two divisors
// (BT709.h:48-49: 1.8556f and 1.5748f).  Decides whether the encoder may replace __fdiv_rn.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k(float c, float rc, unsigned long long *bad, unsigned long long *bad_range) {
  const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
  unsigned long long nb = 0, nr = 0;
  for (uint64_t u = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
    const float x = __uint_as_float(uint32_t(u));
    if (!(x == x) || fabsf(x) > 3.0e38f) continue;
    const float want = __fdiv_rn(x, c);
    const float q0 = __fmul_rn(x, rc);
    const float e = __fmaf_rn(-c, q0, x);
    const float q = __fmaf_rn(e, rc, q0);
    if (__float_as_uint(q) != __float_as_uint(want)) {
      ++nb;
      if (fabsf(x) <= 4.0f && fabsf(x) >= 1e-30f) ++nr;
    }
  }
  atomicAdd(bad, nb);
  atomicAdd(bad_range, nr);
}
int main() {
  unsigned long long *d; hipMalloc(&d, 16);
  for (float c : {1.8556f, 1.5748f}) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, c, 1.0f / c, d, d + 1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    std::printf("c=%.4f: mismatches over all floats %llu, with 1e-30<=|x|<=4: %llu\n", c, h[0], h[1]);
  }
  return 0;
}
