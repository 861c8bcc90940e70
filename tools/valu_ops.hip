// Per-opcode VALU issue cost on gfx950: cycles one wave64 instruction occupies a SIMD when w waves
// per SIMD issue independent instructions of ONE kind (inline asm, so hipcc cannot fold or pack).
// Decides how busy the decode kernels' VALU really is: their mix is converts, compares, selects
// and integer ops as much as f32 adds/multiplies.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_ops.hip -o tools/bin/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){std::fprintf(stderr,"%s\n",hipGetErrorString(e_)); std::exit(1);} } while(0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

// each OP(i) is one instruction on register set i; 8 sets, 8 repeats = 64 per iteration
#define BODY(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL(NAME, ASMSTR)                                                                       \
  __global__ void __launch_bounds__(256) NAME(float *out, int iters, long long *cyc) {             \
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 1.0001f + blockIdx.x * 1e-7f;                                                       \
    unsigned m = 0xffffff00u;                                                                     \
    long long t0 = clock64();                                                                     \
    for (int it = 0; it < iters; ++it) {                                                          \
      asm volatile(ASMSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "s"(m) : "vcc");                                                \
    }                                                                                             \
    long long t1 = clock64();                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                  \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                      \
  }

#define X8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define S8(s) s s s s s s s s

#define ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define CVTU(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %9, %8\n"
#define CMPSEL(i) "v_cmp_ge_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define CMPADDC(i) "v_cmp_ge_f32 vcc, %" #i ", %8\nv_addc_co_u32 %" #i ", vcc, 0, %" #i ", vcc\n"
#define PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define CVTI(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define MOV(i) "v_mov_b32 %" #i ", %8\n"
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define CMP(i) "v_cmp_ge_f32 vcc, %" #i ", %8\n"
#define CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"

// packed f32: operands are VGPR pairs (float2), two results per instruction
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, ASMSTR)                                                                      \
  __global__ void __launch_bounds__(256) NAME(float *out, int iters, long long *cyc) {             \
    f2 a0 = {threadIdx.x * 0.001f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    f2 b = {1.0001f + blockIdx.x * 1e-7f, 0.9999f};                                              \
    long long t0 = clock64();                                                                     \
    for (int it = 0; it < iters; ++it) {                                                          \
      asm volatile(ASMSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
    }                                                                                             \
    long long t1 = clock64();                                                                     \
    const f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                           \
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                      \
  }
#define PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define PKADDC(i) "v_pk_add_f32 %" #i ", %" #i ", %8 clamp\n"
#define MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %8\n"
#define SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define ADDCL(i) "v_add_f32_e64 %" #i ", %" #i ", %8 clamp\n"
KERNEL2(k_pkadd, S8(X8(PKADD)))
KERNEL2(k_pkmul, S8(X8(PKMUL)))
KERNEL2(k_pkfma, S8(X8(PKFMA)))
KERNEL2(k_pkaddc, S8(X8(PKADDC)))
KERNEL(k_med3, S8(X8(MED3)))
KERNEL(k_sub, S8(X8(SUBF)))
KERNEL(k_addcl, S8(X8(ADDCL)))

KERNEL(k_add, S8(X8(ADD)))
KERNEL(k_mul, S8(X8(MUL)))
KERNEL(k_fma, S8(X8(FMA)))
KERNEL(k_min, S8(X8(MINF)))
KERNEL(k_cvtu, S8(X8(CVTU)))
KERNEL(k_cvtub, S8(X8(CVTUB)))
KERNEL(k_cvti, S8(X8(CVTI)))
KERNEL(k_andor, S8(X8(ANDOR)))
KERNEL(k_cmpsel, S8(X8(CMPSEL)))
KERNEL(k_cmpaddc, S8(X8(CMPADDC)))
KERNEL(k_cmp, S8(X8(CMP)))
KERNEL(k_cnd, S8(X8(CNDS)))
KERNEL(k_perm, S8(X8(PERM)))
KERNEL(k_or3, S8(X8(OR3)))
KERNEL(k_lshladd, S8(X8(LSHLADD)))
KERNEL(k_sdwa, S8(X8(SDWA)))
KERNEL(k_mov, S8(X8(MOV)))

typedef void (*kern_t)(float *, int, long long *);

void run(const char *name, kern_t k, int instr_per_iter, float *d, long long *dc, int w) {
  const int blocks = 256 * w, iters = 2048;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, dc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, dc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long cyc = 0; CK(hipMemcpy(&cyc, dc, sizeof cyc, hipMemcpyDeviceToHost));
  const double per_simd = double(instr_per_iter) * iters * w;
  std::printf("%-22s w=%d  %8.3f ms  %6.2f ns/instr/SIMD  clock64: %6.2f ticks/instr/SIMD\n", name, w, ms,
              ms * 1e6 / per_simd, double(cyc) / per_simd);
}

int main() {
  float *d; long long *dc;
  CK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
  CK(hipMalloc(&dc, 8));
  struct { const char *n; kern_t k; int per; } ks[] = {
      {"v_add_f32", k_add, 64}, {"v_mul_f32", k_mul, 64}, {"v_fma_f32", k_fma, 64}, {"v_min_f32", k_min, 64},
      {"v_cvt_u32_f32", k_cvtu, 64}, {"v_cvt_f32_ubyte1", k_cvtub, 64}, {"v_cvt_f32_i32", k_cvti, 64},
      {"v_and_or_b32", k_andor, 64}, {"v_cmp+v_cndmask", k_cmpsel, 128}, {"v_cmp+v_addc", k_cmpaddc, 128},
      {"v_cmp_ge_f32", k_cmp, 64}, {"v_cndmask_b32", k_cnd, 64}, {"v_perm_b32", k_perm, 64},
      {"v_or3_b32", k_or3, 64}, {"v_lshl_add_u32", k_lshladd, 64}, {"v_add_u32_sdwa", k_sdwa, 64},
      {"v_mov_b32", k_mov, 64}, {"v_med3_f32", k_med3, 64}, {"v_sub_f32", k_sub, 64}, {"v_add_f32 clamp (e64)", k_addcl, 64},
      {"v_pk_add_f32 (2 results)", k_pkadd, 64}, {"v_pk_mul_f32 (2 results)", k_pkmul, 64},
      {"v_pk_fma_f32 (2 results)", k_pkfma, 64}, {"v_pk_add_f32 clamp", k_pkaddc, 64}};
  for (int w : {1, 2, 4, 8})
    for (auto &k : ks) run(k.n, k.k, k.per, d, dc, w);
  return 0;
}
