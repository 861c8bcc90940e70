// What does v_cvt_pk_u8_f32 do with fractions, negatives and values past 255 on gfx950?
// (decides whether it can quantise + pack in one instruction in the encoder)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const float *in, uint32_t *out, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  uint32_t w = 0xAABBCCDDu, r;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(in[i]), "v"(w));
  out[i] = r;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 1, %2\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(r) : "v"(in[i]), "v"(w));
  out[64 + i] = r;
}
int main() {
  float h[] = {0.f, 0.25f, 0.49999f, 0.5f, 0.50001f, 0.75f, 0.99999f, 1.f, 1.5f, 2.5f, 3.5f, 254.5f, 254.99f, 255.f, 255.4f, 255.5f, 256.f, 300.f, -0.25f, -0.75f, -3.f, 127.5f, 128.5f};
  int n = sizeof h / sizeof *h;
  float *d; uint32_t *o, ho[128];
  hipMalloc(&d, sizeof h); hipMalloc(&o, 128 * 4);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(ho, o, 128 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) std::printf("%10.5f -> %08x (byte %u)   RTZ mode: byte %u\n", h[i], ho[i], (ho[i] >> 8) & 0xff, (ho[64 + i] >> 8) & 0xff);
  return 0;
}
