// Device-side building blocks shared by the decode (bt709_kernels.hip) and the fused
// decode + rescale kernels (bt709_rescale_half.hip, bt709_rescale_scaled.hip).  gfx950 only.
//
// Arithmetic follows the reference's CPU path (Renderer/BT709.h:466-513, 348-460, 821-908),
// which is what the 8-bit output is checked against:
//
//   Yn  = (Y  -  16) * (1/255f)          BT709.h:494
//   Cbn = (Cb - 128) * (1/255f)          BT709.h:499
//   Crn = (Cr - 128) * (1/255f)          BT709.h:500
//   R = ((Yn*My) + (Cbn*0))     + (Crn*Mcr_r)      BT709.h:424
//   G = ((Yn*My) + (Cbn*Mcb_g)) + (Crn*Mcr_g)      BT709.h:425
//   B = ((Yn*My) + (Cbn*Mcb_b)) + (Crn*0)          BT709.h:426
//   saturate, transfer curve(s), (int)round(v*255f)  BT709.h:444-446, 856-883
//
// Adding the +-0 products of the zero matrix entries never changes a sum's value (only possibly
// the sign of an exact zero, which saturates to the same bucket), so they are not computed.
// Every multiply and add of the matrix step is a separate IEEE binary32 operation: the files are
// compiled with -ffp-contract=off and the arithmetic goes through __fmul_rn/__fadd_rn so no FMA can
// form there (a CPU test greps the ISA); the one deliberate fma is centre_norm below, where it is
// provably the same rounding.
//
// VALU BUDGET (tools/valu_ops.hip, waves of 64): v_add_f32, v_mul_f32 and v_mov_b32 occupy a
// SIMD for 2 cycles; every other VALU instruction these kernels use -- converts, compares,
// selects, integer and bit operations, SDWA forms -- for 4.  The 1:1 kernel runs the VALU ~60 %
// busy at 6 TB/s and the rescale kernels are VALU-bound outright, so the helpers below are
// written by cycle count:
//   * byte -> float is v_cvt_f32_ubyteN (4) and the centring a float add (2); left alone, hipcc
//     rewrites (float)(byte) - 16.0f as an integer SDWA add plus v_cvt_f32_i32 (4 + 4);
//   * R, G, B are saturated for free by the clamp bit of the add that produces them;
//   * the bucket index of a saturated x is ONE 2-cycle add (magic_index12) instead of a 4-cycle
//     convert: with M = 2^23 / N a float in [M, 2M) has ulp 1/N, so x + M is M + round(x N) / N
//     and its bit pattern is bits(M) + round(x N); a v_lshl_add_u32 turns that into the LDS byte
//     address (its addend cancels bits(M) << shift).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "bt709_constants.h"
#include "bt709_kernels.h"
#include "bt709_quantise.h"
#include "bt709_stage.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// LDS reads through integer byte addresses (address space 3): no base-pointer add per lookup
typedef __attribute__((address_space(3))) const u32x2 *LdsPairPtr;  // ds_read_b64
typedef __attribute__((address_space(3))) const u32x4 *LdsQuadPtr;  // ds_read_b128

__device__ __forceinline__ uint32_t lds_address(const void *p) {
  return static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const unsigned char *)p));
}

__device__ __forceinline__ float byte_of(uint32_t w, int i) {
  float f = static_cast<float>((w >> (8 * i)) & 0xffu);  // v_cvt_f32_ubyte{i}
  asm("" : "+v"(f));  // opaque: keeps the centring a float add (see VALU BUDGET)
  return f;
}

__device__ __forceinline__ float byte_value(uint8_t b) {
  float f = static_cast<float>(b);
  asm("" : "+v"(f));
  return f;
}

// (v - off) * (1/255f), v an integer-valued float (a byte), off 16 or 128.  The reference subtracts
// in int (exact), converts and multiplies: ONE rounding, of the real number (v - off) / 255f.  An
// fma computes v * (1/255f) + (-off * (1/255f)) with one rounding too, and -off * (1/255f) is exact
// (off is a power of two), so the real number rounded is the same one: bit-identical, one
// instruction instead of two.  This is the only fused multiply-add in the decode kernels (a CPU
// test counts them: one per converted byte, none anywhere else).
__device__ __forceinline__ float centre_norm(float v, float off) {
  return __builtin_fmaf(v, kInv255, -off * kInv255);
}

// a + b saturated to [0, 1] (BT709.h:444-446 `saturatef`) by the add's own clamp bit
__device__ __forceinline__ float add_sat(float a, float b) {
  float r;
  asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct Chroma {  // the four Cb/Cr products of one 2x2 block (BT709.h:389-397 matrix entries)
  float cr_r, cb_g, cr_g, cb_b;
};

__device__ __forceinline__ Chroma chroma_terms(float cb, float cr) {
  const float cbn = centre_norm(cb, 128.0f);
  const float crn = centre_norm(cr, 128.0f);
  Chroma c;
  c.cr_r = __fmul_rn(crn, kMCrR);
  c.cb_g = __fmul_rn(cbn, kMCbG);
  c.cr_g = __fmul_rn(crn, kMCrG);
  c.cb_b = __fmul_rn(cbn, kMCbB);
  return c;
}

// saturated non-linear R, G, B of one pixel: exactly the reference's floats
__device__ __forceinline__ void pixel_rgb(float ybyte, const Chroma &c, float &r, float &g, float &b) {
  const float yv = __fmul_rn(centre_norm(ybyte, 16.0f), kMY);
  r = add_sat(yv, c.cr_r);
  g = add_sat(__fadd_rn(yv, c.cb_g), c.cr_g);
  b = add_sat(yv, c.cb_b);
}

// linear alpha sample: the luma term alone (AAPLShaders.metal:249-271; CPU twin BT709.h:466-513)
__device__ __forceinline__ float alpha_value(float abyte) {
  return add_sat(__fmul_rn(centre_norm(abyte, 16.0f), kMY), 0.0f);
}

// (int)round(x * 255.0f) of a saturated x (BT709.h:881-883), the whole composite of the sRGB mode ("no curve at
// all", BT709.h:977-983) and therefore of every channel of an alpha decoder, in its three-instruction form
// trunc(x * 255.0f + 0.5f).  That form is NOT exact for every float (bt709_quantise.h: x = 0x3b008080 comes out 1, the
// reference gives 0); it is used here only on the 1:1 kernels' arguments, a finite set that the exhaustive sweeps
// enumerate (2^24 triples, 256 alpha codes).  Filtered values go through quantise_exact.
__device__ __forceinline__ uint32_t quantise_byte(float x) {
  return quantise_enumerated(x);
}

// t[i] = bits(x[i] + magic) for saturated x: bits(magic) + bucket index (transfer_tables.h).  The add as it stands
// (round to nearest even: bucket q is centred on q / N); the host table builder replays it (transfer_tables.cpp
// bucket_index).  Round 1's floor(x N) form -- the adds between two writes of MODE.fp_round -- is a lab variant
// (tools/lab_variants.py index_rtz).
__device__ __forceinline__ void magic_index12(const float *x, uint32_t *t, float magic) {
#pragma unroll
  for (int i = 0; i < 12; ++i) t[i] = __float_as_uint(__fadd_rn(x[i], magic));
}

__device__ __forceinline__ void magic_index4(const float *x, uint32_t *t, float magic) {
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __float_as_uint(__fadd_rn(x[i], magic));
}

// Lookup constants of the byte-valued bucket table (TransferBucket, 8 bytes) held at `lds_table`: bucket q of x is
// (bits(x + magic) >> shift) - first (DecodeParams::unit1_*: uniform form shift 0, log-bucket form shift 16).
struct UnitLookup {
  float magic;      // M = 2^23 / N, or the log form's addend
  uint32_t offset;  // LDS address of the table - (first << 3)
  uint32_t shift;   // 0 / 16; the fast kernels take it as a template argument instead
};

template <typename Params>  // DecodeParams or the +unconvert: kernel's own
__device__ __forceinline__ UnitLookup unit_lookup(const Params &p, const void *lds_table) {
  UnitLookup u;
  u.magic = p.unit1_magic;
  u.offset = lds_address(lds_table) - (p.unit1_first << 3);
  u.shift = p.unit1_shift;
  return u;
}

// byte of the decoder's gamma for saturated x, t = bits(x + M) from magic_index*
__device__ __forceinline__ uint32_t bucket_byte(const UnitLookup &u, float x, uint32_t t) {
  const u32x2 e = *reinterpret_cast<LdsPairPtr>((t << 3) + u.offset);  // {edge bits, base}
  return e.y + (x >= __uint_as_float(e.x) ? 1u : 0u);
}

// (A<<24)|(R<<16)|(G<<8)|B in two VALU ops: v_perm_b32 places R and G (bytes 2 and 1, zeros
// elsewhere), v_or3_b32 merges B and the alpha word.  (Writing each byte straight into its lane
// with an SDWA v_addc, no pack at all, measured 1 % slower.)
__device__ __forceinline__ uint32_t pack_bgra(uint32_t R, uint32_t G, uint32_t B, uint32_t alpha_word) {
  // selector bytes, MSB first: 0x0c -> 0x00, 0x04 -> byte 0 of the first operand (R),
  // 0x00 -> byte 0 of the second operand (G), 0x0c -> 0x00
  const uint32_t rg = __builtin_amdgcn_perm(R, G, 0x0c04000cu);
  return rg | B | alpha_word;
}

__device__ __forceinline__ void stage_table(void *lds, const void *src, uint32_t bytes) {
  u32x4 *d = reinterpret_cast<u32x4 *>(lds);
  const u32x4 *s = reinterpret_cast<const u32x4 *>(src);
  const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
  const uint32_t n = bytes / 16;
  if (n <= nthreads) {  // uniform: the 4 KiB table of the Apple mode, one load and one write per lane (the loop form: its code is what was tuned)
    for (uint32_t i = tid; i < n; i += nthreads) d[i] = s[i];
    return;
  }
  // Large tables (the LINEAR mode's 4 096 buckets = 33 KiB: five rounds for 512 lanes): all of a lane's loads are issued
  // before its first write.  As a plain loop every round was its own L2 round trip inside the workgroup's lifetime
  // (SQ_WAIT_INST_LDS 442 M against 33 M cycles per launch for the 4 KiB table, +28 % wave cycles; round 3).
  constexpr int kBatch = 5;
  for (uint32_t base = tid; base < n; base += nthreads * kBatch) {
    u32x4 v[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const uint32_t i = base + static_cast<uint32_t>(k) * nthreads;
      if (i < n) v[k] = s[i];
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const uint32_t i = base + static_cast<uint32_t>(k) * nthreads;
      if (i < n) d[i] = v[k];
    }
  }
}

// Frame bytes are touched exactly once: stream them past the caches (measured +1.3 % on 4K)
template <bool NT>
__device__ __forceinline__ uint32_t load32(const uint8_t *p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
  return *reinterpret_cast<const uint32_t *>(p);
}

template <bool NT>
__device__ __forceinline__ void store16(uint8_t *p, u32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
  else *reinterpret_cast<u32x4 *>(p) = v;
}

template <bool NT>
__device__ __forceinline__ void store8(uint8_t *p, u32x2 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p));
  else *reinterpret_cast<u32x2 *>(p) = v;
}

// Frame i of the launch: from the kernarg table, or -- when the caller's frames are evenly
// spaced in memory (a ring / pool) -- frame 0 plus i times the spacing, which lifts the
// 32-frame limit of the table.
__device__ __forceinline__ FramePlanes frame_planes(const DecodeParams &p, uint32_t i) {
  if (!p.uniform) return p.frames[i];
  FramePlanes f = p.frames[0];
  f.y += static_cast<int64_t>(i) * p.step_y;
  f.cbcr += static_cast<int64_t>(i) * p.step_cbcr;
  if (f.alpha) f.alpha += static_cast<int64_t>(i) * p.step_alpha;
  f.out += static_cast<int64_t>(i) * p.step_out;
  return f;
}

}  // namespace
}  // namespace bt709
