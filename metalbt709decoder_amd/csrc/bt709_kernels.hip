// CDNA4 (gfx950) kernels of the BT.709 NV12 -> sRGB BGRA decode path.
//
// What the reference does in two Metal passes -- BT709ToLinearSRGBKernel & friends
// (Renderer/AAPLShaders.metal:336-407: read Y(gid) and CbCr(gid/2), 3x3 matrix,
// video-gamma removal, write into an sRGB8 texture whose store hardware applies
// the sRGB OETF) and, when the view is smaller, samplingShader (73-85) -- is ONE
// kernel here.  Arithmetic follows the reference's CPU path (Renderer/BT709.h:
// 466-513, 348-460, 821-908), which is what the 8-bit output is checked against:
//
//   Yn  = (Y  -  16) * (1/255f)          BT709.h:494
//   Cbn = (Cb - 128) * (1/255f)          BT709.h:499
//   Crn = (Cr - 128) * (1/255f)          BT709.h:500
//   R = ((Yn*My) + (Cbn*0))     + (Crn*Mcr_r)      BT709.h:424
//   G = ((Yn*My) + (Cbn*Mcb_g)) + (Crn*Mcr_g)      BT709.h:425
//   B = ((Yn*My) + (Cbn*Mcb_b)) + (Crn*0)          BT709.h:426
//   saturate, transfer curve(s), (int)round(v*255f)  BT709.h:444-446, 856-883
//
// Adding the +-0 products of the zero matrix entries never changes a sum's value
// (only possibly the sign of an exact zero, which maps to byte 0 either way), so
// they are not computed.  Every multiply and add is a separate IEEE binary32
// operation: this file is compiled with -ffp-contract=off and the arithmetic goes
// through __fmul_rn/__fadd_rn so no FMA can form (a CPU test greps the ISA).
//
// SCALED DOMAIN.  The five matrix constants arrive pre-multiplied by N, the
// (power-of-two) bucket count of the transfer table.  Binary floating-point
// rounding commutes with scaling by a power of two (no overflow or subnormal is
// reachable here: the smallest non-zero magnitude is ~8e-4, the largest ~2.2*N), so
// every product and sum below is exactly N times the reference's value, bit for
// bit in the mantissa.  That removes one multiply per lookup (q = (uint)xs directly)
// and the explicit saturate: v_cvt_u32_f32 clamps negatives to bucket 0 (byte 0),
// and the table simply extends to 2.25*N (any reachable R,G,B is < 2.15), where
// every bucket answers 255.  See transfer_tables.h for the table itself:
//       byte = base[q] + (xs >= edge_scaled[q]),  q = (uint)xs.
//
// Memory plan (HBM-bound: 1.5 B read + 4 B written per pixel, no reuse between
// workgroups, so no XCD-aware remap is needed):
//   * a lane owns 4-wide x 2-high pixel "quads": one dword of each luma row, one
//     dword of CbCr (two Cb,Cr pairs, each shared by a 2x2 block -- chroma is
//     REPLICATED, not interpolated: AAPLShaders.metal:350, BGRAToBT709Converter.m:
//     267-277) and two 16-byte non-temporal stores per quad;
//   * consecutive lanes own consecutive quads of the same row pair, so a wave reads
//     3 x 256 contiguous bytes and writes 2 x 1 KiB contiguous, fully coalesced;
//   * grid = (tiles per row pair, row pairs, frames): one short-lived workgroup per
//     tile, dispatched in address order (x fastest).  Measured: long-lived
//     grid-strided workgroups lose ~20 % of the bandwidth to a scattered DRAM stream,
//     and every extra instruction per wave costs about its share of run time, so the
//     kernels have no loops and no integer divisions;
//   * the tile's global loads are issued before the table is staged into LDS.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "bt709_constants.h"
#include "bt709_kernels.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// VALU budget (tools/valu_ops.hip, gfx950, waves of 64): v_add_f32 / v_mul_f32 / v_mov_b32 occupy a
// SIMD for 2 cycles, every other VALU instruction these kernels use (converts, compares, selects,
// integer and bit ops, SDWA forms) for 4.  At ~27 instructions per pixel the 1:1 kernel keeps the
// VALU ~65 % busy at 6 TB/s, so instruction count is run time here, not just memory.
#ifndef BT709_OPT_UBYTE
#define BT709_OPT_UBYTE 1
#endif
#ifndef BT709_OPT_SDWA
#define BT709_OPT_SDWA 0  // measured: -1 % (decode_lab, same call) -- kept for the lab
#endif

__device__ __forceinline__ float byte_of(uint32_t w, int i) {
  float f = static_cast<float>((w >> (8 * i)) & 0xffu);  // v_cvt_f32_ubyte{i}: 4 cycles
#if BT709_OPT_UBYTE
  // Opaque to the optimiser: otherwise (float)(byte) + (-16.0f) is rewritten as an integer SDWA
  // add plus v_cvt_f32_i32 (4 + 4 cycles) instead of this convert plus a 2-cycle v_add_f32.
  asm("" : "+v"(f));
#endif
  return f;
}

// (v - off) * (1/255f): integer-valued floats subtract exactly, so this equals the
// reference's int subtract followed by int->float conversion.
__device__ __forceinline__ float centre_norm(float v, float off) {
  return __fmul_rn(__fadd_rn(v, -off), kInv255);
}

// xs = N * x.  Returns the output byte of the decoder's gamma for pre-gamma value x.
__device__ __forceinline__ uint32_t lookup(const TransferBucket *__restrict__ tbl, float xs) {
#if defined(BT709_LAB_NO_LDS)  // tools/decode_lab only: price of the LDS lookups (wrong output)
  return static_cast<uint32_t>(xs) + (xs >= 77.0f ? 1u : 0u) + (tbl == nullptr ? 1u : 0u);
#else
  const uint32_t q = static_cast<uint32_t>(xs);  // floor for xs >= 0, 0 for xs < 0
  const TransferBucket e = tbl[q];
  return e.base + (xs >= e.edge ? 1u : 0u);
#endif
}

// (A<<24)|(R<<16)|(G<<8)|B in two VALU ops: v_perm_b32 places R and G (bytes 2 and 1, zeros
// elsewhere), v_or3_b32 merges B and the alpha word.
__device__ __forceinline__ uint32_t pack_bgra(uint32_t R, uint32_t G, uint32_t B, uint32_t alpha_word) {
  // selector bytes, MSB first: 0x0c -> 0x00, 0x04 -> byte 0 of the first operand (R),
  // 0x00 -> byte 0 of the second operand (G), 0x0c -> 0x00
  const uint32_t rg = __builtin_amdgcn_perm(R, G, 0x0c04000cu);
  return rg | B | alpha_word;
}

struct Matrix {  // BT709.h:389-397 times N (DecodeParams::m_*)
  float y, cr_r, cb_g, cr_g, cb_b;
};

struct Chroma {  // the four scaled Cb/Cr products of one 2x2 block
  float cr_r, cb_g, cr_g, cb_b;
};

__device__ __forceinline__ Chroma chroma_terms(const Matrix &m, float cb, float cr) {
  const float cbn = centre_norm(cb, 128.0f);
  const float crn = centre_norm(cr, 128.0f);
  Chroma c;
  c.cr_r = __fmul_rn(crn, m.cr_r);
  c.cb_g = __fmul_rn(cbn, m.cb_g);
  c.cr_g = __fmul_rn(crn, m.cr_g);
  c.cb_b = __fmul_rn(cbn, m.cb_b);
  return c;
}

// scaled (N x) non-linear R,G,B of one pixel, NOT saturated
__device__ __forceinline__ void pixel_rgbs(const Matrix &m, float ybyte, const Chroma &c, float &r, float &g,
                                           float &b) {
  const float yv = __fmul_rn(centre_norm(ybyte, 16.0f), m.y);
  r = __fadd_rn(yv, c.cr_r);
  g = __fadd_rn(__fadd_rn(yv, c.cb_g), c.cr_g);
  b = __fadd_rn(yv, c.cb_b);
}

// word.byte[LANE] = bucket.base + (xs >= bucket.edge): the compare's carry goes straight into byte
// LANE of the output word (SDWA destination select, other bytes preserved), so B, G, R (and a
// decoded alpha) need no pack instructions at all.
template <int LANE>
__device__ __forceinline__ void lookup_into(uint32_t &word, const TransferBucket *__restrict__ tbl, float xs,
                                            uint32_t zero) {
  const uint32_t q = static_cast<uint32_t>(xs);
  const TransferBucket e = tbl[q];
  static_assert(LANE >= 0 && LANE < 4, "byte lane");
  if (LANE == 0)
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(word) : "v"(xs), "v"(e.edge), "v"(e.base), "v"(zero) : "vcc");
  else if (LANE == 1)
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(word) : "v"(xs), "v"(e.edge), "v"(e.base), "v"(zero) : "vcc");
  else if (LANE == 2)
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(word) : "v"(xs), "v"(e.edge), "v"(e.base), "v"(zero) : "vcc");
  else
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(word) : "v"(xs), "v"(e.edge), "v"(e.base), "v"(zero) : "vcc");
}

__device__ __forceinline__ uint32_t decode_px(const TransferBucket *__restrict__ tbl, const Matrix &m, float ybyte,
                                              const Chroma &c, uint32_t alpha_word) {
  float r, g, b;
  pixel_rgbs(m, ybyte, c, r, g, b);
#if BT709_OPT_SDWA
  uint32_t word = alpha_word, zero = 0;
  asm("" : "+v"(zero));  // one VGPR of zeros (an SDWA source operand cannot be an inline constant)
  lookup_into<2>(word, tbl, r, zero);
  lookup_into<1>(word, tbl, g, zero);
  lookup_into<0>(word, tbl, b, zero);
  return word;
#else
  const uint32_t R = lookup(tbl, r);
  const uint32_t G = lookup(tbl, g);
  const uint32_t B = lookup(tbl, b);
  return pack_bgra(R, G, B, alpha_word);
#endif
}

// linear alpha sample -> byte: R channel of the matrix with Cb=Cr=128, then plain
// 8-bit quantisation (AAPLShaders.metal:249-271; CPU twin BT709.h:466-513).  An alpha
// decoder always runs the identity (sRGB-mode) table, which is exactly round(x*255).
__device__ __forceinline__ uint32_t decode_alpha(const TransferBucket *__restrict__ ident, const Matrix &m,
                                                 float abyte) {
  return lookup(ident, __fmul_rn(centre_norm(abyte, 16.0f), m.y)) << 24;
}

#ifndef BT709_OPT_UNIT
#define BT709_OPT_UNIT 1
#endif

// ---------------------------------------------------------------------------
// UNIT-DOMAIN lookups of the fast kernel.  R, G, B are formed exactly as the reference forms
// them (unscaled matrix, BT709.h:389-426) and saturated for free by the clamp bit of the add
// that produces them (BT709.h:444-446), so the table needs only its N + 1 unit entries.  The
// bucket index costs one 2-cycle add instead of a 4-cycle convert: with M = 2^23 / N a float in
// [M, 2M) has ulp 1/N, so x + M rounded TOWARD ZERO is M + floor(x N) / N and its bit pattern
// is bits(M) + floor(x N).  v_lshl_add_u32 turns that into the LDS byte address (the constant
// term cancels bits(M) << 3).  The round-toward-zero adds sit in one asm statement between two
// s_setreg of MODE.fp_round's single-precision field; everything else rounds to nearest even.
// ---------------------------------------------------------------------------
struct UnitLookup {
  float magic;      // M = 2^23 / N
  uint32_t offset;  // LDS address of the table - (bits(M) << 3)
};

__device__ __forceinline__ float add_sat(float a, float b) {
  float r;
  asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// t[i] = bits(x[i] + M), rounded toward zero, 12 at a time (an asm statement takes 30 operands)
__device__ __forceinline__ void magic_floor12(const float *x, uint32_t *t, float magic) {
  asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_add_f32 %0, %24, %12\n\tv_add_f32 %1, %24, %13\n\tv_add_f32 %2, %24, %14\n\tv_add_f32 %3, %24, %15\n\t"
      "v_add_f32 %4, %24, %16\n\tv_add_f32 %5, %24, %17\n\tv_add_f32 %6, %24, %18\n\tv_add_f32 %7, %24, %19\n\t"
      "v_add_f32 %8, %24, %20\n\tv_add_f32 %9, %24, %21\n\tv_add_f32 %10, %24, %22\n\tv_add_f32 %11, %24, %23\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
        "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]),
        "v"(x[10]), "v"(x[11]), "s"(magic));
}

typedef __attribute__((address_space(3))) const u32x2 *LdsBucketPtr;  // {edge bits, base}: one ds_read_b64, no base add

// One 4x2 quad: 8 pixels x (R, G, B) = 24 lookups.  x[3 * p + c], pixel p = 0..3 top row, 4..7 bottom.
template <bool HAS_ALPHA>
__device__ __forceinline__ void quad_unit(const UnitLookup &u, uint32_t ya, uint32_t yb, uint32_t cw, uint32_t aa,
                                          uint32_t ab, uint32_t alpha_word, u32x4 &top, u32x4 &bot) {
  const Matrix m = {kMY, kMCrR, kMCbG, kMCrG, kMCbB};
  const Chroma c0 = chroma_terms(m, byte_of(cw, 0), byte_of(cw, 1));
  const Chroma c1 = chroma_terms(m, byte_of(cw, 2), byte_of(cw, 3));
  float x[24];
#pragma unroll
  for (int px = 0; px < 8; ++px) {
    const uint32_t w = px < 4 ? ya : yb;
    const Chroma &c = (px & 2) ? c1 : c0;
    const float yv = __fmul_rn(centre_norm(byte_of(w, px & 3), 16.0f), m.y);
    x[3 * px + 0] = add_sat(yv, c.cr_r);
    x[3 * px + 1] = add_sat(__fadd_rn(yv, c.cb_g), c.cr_g);
    x[3 * px + 2] = add_sat(yv, c.cb_b);
  }
  uint32_t t[24];
  magic_floor12(x, t, u.magic);
  magic_floor12(x + 12, t + 12, u.magic);
  uint32_t byte[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    const u32x2 e = *reinterpret_cast<LdsBucketPtr>((t[i] << 3) + u.offset);
    byte[i] = e.y + (x[i] >= __uint_as_float(e.x) ? 1u : 0u);
  }
  uint32_t al[8];
#pragma unroll
  for (int px = 0; px < 8; ++px) al[px] = alpha_word;
  if (HAS_ALPHA) {
    float a[12];
    uint32_t ta[12];
#pragma unroll
    for (int px = 0; px < 8; ++px)
      a[px] = add_sat(__fmul_rn(centre_norm(byte_of(px < 4 ? aa : ab, px & 3), 16.0f), m.y), 0.0f);
#pragma unroll
    for (int i = 8; i < 12; ++i) a[i] = 0.0f;
    magic_floor12(a, ta, u.magic);
#pragma unroll
    for (int px = 0; px < 8; ++px) {
      const u32x2 e = *reinterpret_cast<LdsBucketPtr>((ta[px] << 3) + u.offset);
      al[px] = (e.y + (a[px] >= __uint_as_float(e.x) ? 1u : 0u)) << 24;
    }
  }
  top.x = pack_bgra(byte[0], byte[1], byte[2], al[0]);
  top.y = pack_bgra(byte[3], byte[4], byte[5], al[1]);
  top.z = pack_bgra(byte[6], byte[7], byte[8], al[2]);
  top.w = pack_bgra(byte[9], byte[10], byte[11], al[3]);
  bot.x = pack_bgra(byte[12], byte[13], byte[14], al[4]);
  bot.y = pack_bgra(byte[15], byte[16], byte[17], al[5]);
  bot.z = pack_bgra(byte[18], byte[19], byte[20], al[6]);
  bot.w = pack_bgra(byte[21], byte[22], byte[23], al[7]);
}

__device__ __forceinline__ void stage_table(void *lds, const void *src, uint32_t bytes) {
  u32x4 *d = reinterpret_cast<u32x4 *>(lds);
  const u32x4 *s = reinterpret_cast<const u32x4 *>(src);
  const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
  for (uint32_t i = tid; i < bytes / 16; i += nthreads) d[i] = s[i];
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

__device__ __forceinline__ Matrix matrix_of(const DecodeParams &p) {
  return Matrix{p.m_y, p.m_cr_r, p.m_cb_g, p.m_cr_g, p.m_cb_b};
}

}  // namespace

// ---------------------------------------------------------------------------
// Fast path.  Preconditions (checked by the host shim): width % 4 == 0; y, cbcr,
// alpha pointers and strides 4-byte aligned; output pointer and stride 16-byte
// aligned.  grid = (tiles, H/2, frames); a tile is blockDim * kQuadsPerLane quads.
// ---------------------------------------------------------------------------
template <bool HAS_ALPHA, bool NT>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_quads(const DecodeParams p) {
  constexpr int UNROLL = kQuadsPerLane;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucket *tbl = reinterpret_cast<TransferBucket *>(lds_raw);

  const FramePlanes f = frame_planes(p, blockIdx.z);
  const Matrix m = matrix_of(p);
  const uint32_t quads = p.width >> 2;
  const uint32_t row_pairs = p.height >> 1;
  // blockDim.y > 1 only for narrow frames: a workgroup then covers blockDim.y consecutive row
  // pairs so that it still has ~8 waves (1920-wide: 256 x 2)
  // blockDim.x is a whole number of waves, so threadIdx.y is the same in every lane of a wave:
  // taking it from the first lane makes the row pointers scalar (SGPR base + per-lane offset
  // addressing, no 64-bit VALU address arithmetic).
  const uint32_t rp_raw = blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y);
  const uint32_t rp = min(rp_raw, row_pairs - 1);
  const uint32_t q0 = blockIdx.x * (blockDim.x * UNROLL) + threadIdx.x;

  const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
  const uint8_t *y1 = y0 + p.y_stride;
  const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
  const uint8_t *a0 = HAS_ALPHA ? f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride : nullptr;
  uint8_t *o0 = f.out + static_cast<size_t>(2 * rp) * p.out_stride;
  uint8_t *o1 = o0 + p.out_stride;

  // Straight-line code: lanes past the row's end load a clamped (valid) quad and only their
  // stores are predicated.  A divergent `if (q < quads)` around the arithmetic made hipcc put
  // s_waitcnt vmcnt(0) at the join, i.e. each wave waited for the write acknowledgement of its
  // first quad's stores before touching its second quad.
  uint32_t ya[UNROLL], yb[UNROLL], cw[UNROLL], aa[UNROLL], ab[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const uint32_t q = min(q0 + u * blockDim.x, quads - 1);
    ya[u] = load32<NT>(y0 + 4 * q);
    yb[u] = load32<NT>(y1 + 4 * q);
#if !defined(BT709_LAB_LDS_CHROMA)
    cw[u] = load32<NT>(cc + 4 * q);
#endif
    if (HAS_ALPHA) {
      aa[u] = *reinterpret_cast<const uint32_t *>(a0 + 4 * q);
      ab[u] = *reinterpret_cast<const uint32_t *>(a0 + p.alpha_stride + 4 * q);
    }
  }
#if defined(BT709_LAB_LDS_CHROMA)
  // tools/decode_lab only: the north-star's "LDS-staged chroma tile" -- the CbCr row segment of
  // the tile is fetched with 16-byte loads by a quarter of the lanes, parked in LDS behind the
  // table, and every lane reads its dword(s) back after the barrier.  (Needs quads % 4 == 0.)
  uint32_t *chroma_lds = reinterpret_cast<uint32_t *>(lds_raw + p.table_bytes);
  {
    const uint32_t span = blockDim.x * UNROLL;                  // dwords of CbCr this tile needs
    const uint32_t base = blockIdx.x * span;
    for (uint32_t i = threadIdx.x; i < span / 4; i += blockDim.x) {
      const uint32_t qd = min(base + 4 * i, quads - 4);
      reinterpret_cast<u32x4 *>(chroma_lds)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(cc + 4 * qd));
    }
  }
#endif
#if BT709_OPT_UNIT
  stage_table(tbl, p.table_unit, p.table_unit_bytes);  // after the tile's loads are in flight
#else
  stage_table(tbl, p.table, p.table_bytes);  // after the tile's loads are in flight
#endif
  __syncthreads();
#if defined(BT709_LAB_LDS_CHROMA)
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) cw[u] = chroma_lds[threadIdx.x + u * blockDim.x];
#endif
  // Pin every loaded dword here: hipcc then waits for all of the tile's loads once, before any
  // store is issued, instead of emitting s_waitcnt vmcnt(0) between the first quad's stores and
  // the second quad's arithmetic (which would wait for the stores' write acknowledgements).
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    asm volatile("" : "+v"(ya[u]), "+v"(yb[u]), "+v"(cw[u]));
    if (HAS_ALPHA) asm volatile("" : "+v"(aa[u]), "+v"(ab[u]));
  }

#if BT709_OPT_UNIT
  UnitLookup ul;
  ul.magic = p.unit_magic;
  ul.offset = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)lds_raw)) -
              (__float_as_uint(ul.magic) << 3);
#endif
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const uint32_t q = q0 + u * blockDim.x;
    u32x4 top, bot;
#if BT709_OPT_UNIT
    quad_unit<HAS_ALPHA>(ul, ya[u], yb[u], cw[u], HAS_ALPHA ? aa[u] : 0u, HAS_ALPHA ? ab[u] : 0u, p.alpha_word, top, bot);
#else
    const Chroma c0 = chroma_terms(m, byte_of(cw[u], 0), byte_of(cw[u], 1));
    const Chroma c1 = chroma_terms(m, byte_of(cw[u], 2), byte_of(cw[u], 3));
    uint32_t al[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      al[i] = HAS_ALPHA ? decode_alpha(tbl, m, byte_of(aa[u], i)) : p.alpha_word;
      al[4 + i] = HAS_ALPHA ? decode_alpha(tbl, m, byte_of(ab[u], i)) : p.alpha_word;
    }
    top.x = decode_px(tbl, m, byte_of(ya[u], 0), c0, al[0]);
    top.y = decode_px(tbl, m, byte_of(ya[u], 1), c0, al[1]);
    top.z = decode_px(tbl, m, byte_of(ya[u], 2), c1, al[2]);
    top.w = decode_px(tbl, m, byte_of(ya[u], 3), c1, al[3]);
    bot.x = decode_px(tbl, m, byte_of(yb[u], 0), c0, al[4]);
    bot.y = decode_px(tbl, m, byte_of(yb[u], 1), c0, al[5]);
    bot.z = decode_px(tbl, m, byte_of(yb[u], 2), c1, al[6]);
    bot.w = decode_px(tbl, m, byte_of(yb[u], 3), c1, al[7]);
#endif
    if (q < quads && rp_raw < row_pairs) {
      store16<NT>(o0 + 16 * q, top);
      store16<NT>(o1 + 16 * q, bot);
    }
  }
}

// ---------------------------------------------------------------------------
// General path: any even width/height, any stride, byte-aligned planes, 4-byte
// aligned output.  One lane per 2x2 block, grid-strided over row pairs.  Correctness
// first; used for ragged or misaligned frames only.
// ---------------------------------------------------------------------------
template <bool HAS_ALPHA>
__global__ void __launch_bounds__(kBlockThreads)
decode_nv12_blocks(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucket *tbl = reinterpret_cast<TransferBucket *>(lds_raw);
  stage_table(tbl, p.table, p.table_bytes);
  __syncthreads();

  const FramePlanes f = frame_planes(p, blockIdx.y);
  const Matrix m = matrix_of(p);
  const uint32_t bw = p.width >> 1;
  const uint32_t row_pairs = p.height >> 1;

  for (uint32_t rp = blockIdx.x; rp < row_pairs; rp += gridDim.x) {
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    uint32_t *o0 = reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(2 * rp) * p.out_stride);
    uint32_t *o1 = reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(2 * rp + 1) * p.out_stride);
    for (uint32_t bx = threadIdx.x; bx < bw; bx += kBlockThreads) {
      const Chroma c = chroma_terms(m, static_cast<float>(cc[2 * bx]), static_cast<float>(cc[2 * bx + 1]));
      uint32_t al[4] = {p.alpha_word, p.alpha_word, p.alpha_word, p.alpha_word};
      if (HAS_ALPHA) {
        const uint8_t *a0 = f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride;
        const uint8_t *a1 = a0 + p.alpha_stride;
        al[0] = decode_alpha(tbl, m, static_cast<float>(a0[2 * bx]));
        al[1] = decode_alpha(tbl, m, static_cast<float>(a0[2 * bx + 1]));
        al[2] = decode_alpha(tbl, m, static_cast<float>(a1[2 * bx]));
        al[3] = decode_alpha(tbl, m, static_cast<float>(a1[2 * bx + 1]));
      }
      o0[2 * bx] = decode_px(tbl, m, static_cast<float>(y0[2 * bx]), c, al[0]);
      o0[2 * bx + 1] = decode_px(tbl, m, static_cast<float>(y0[2 * bx + 1]), c, al[1]);
      o1[2 * bx] = decode_px(tbl, m, static_cast<float>(y1[2 * bx]), c, al[2]);
      o1[2 * bx + 1] = decode_px(tbl, m, static_cast<float>(y1[2 * bx + 1]), c, al[3]);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused decode + exact 2:1 downscale (pass 1 + pass 2 of the reference).  A 2x2
// luma block shares one CbCr sample and becomes one output pixel.  Two-pass
// equivalent arithmetic: each decoded byte is linearised as the sRGB8 sampler
// would (the decode-side table returns the linear float directly), the four are
// averaged (((a+b)+c)+d)*0.25f, then sRGB-encoded and quantised through the
// LINEAR-mode table (second LDS table, same scaled-domain lookup).
//
// WIDE: a lane owns one quad = 4x2 source pixels = 2 output pixels (two dword luma
// loads, one dword CbCr load, one 8-byte store); grid = (tiles, H/2, frames) as in
// the 1:1 kernel.  Preconditions: width % 4 == 0, planes/strides 4-byte aligned,
// output 8-byte aligned.  !WIDE: one lane per output pixel, byte loads, any layout.
// ---------------------------------------------------------------------------
namespace {

// The decode-side table of the rescale kernel is NOT extended past N (16-byte entries: an
// extended LINEAR-mode table would not fit LDS next to the encode table), so xs is clamped.
#ifndef BT709_HALF_BATCH
#define BT709_HALF_BATCH 6
#endif
constexpr int kHalfBatch = BT709_HALF_BATCH;

__device__ __forceinline__ uint32_t linear_index(float n, float &xs) {
  xs = __builtin_fminf(xs, n);
  return static_cast<uint32_t>(xs);
}

// One 16-byte bucket {edge, lin_below, lin_above, base} with a single ds_read_b128 (the caller
// pins whole vectors, which keeps hipcc from narrowing the read to the much slower ds_read_b96).
__device__ __forceinline__ u32x4 linear_fetch(const TransferBucketLinear *__restrict__ tbl, uint32_t q) {
  return reinterpret_cast<const u32x4 *>(tbl)[q];
}

__device__ __forceinline__ float linear_select(const u32x4 &e, float xs) {
  return xs >= __uint_as_float(e.x) ? __uint_as_float(e.z) : __uint_as_float(e.y);
}

struct SplitIndex {  // transfer_tables.h SplitTable
  float split, coarse;
  uint32_t offset;
  uint32_t coarse_shift;  // log2(1 / coarse)
};

// two-resolution table: fine buckets below `split`, `1/coarse` times wider ones above
__device__ __forceinline__ uint32_t lookup_split(const TransferBucket *__restrict__ tbl, const SplitIndex &s, float xs) {
  // the fine and the coarse index functions cross exactly at the split and the fine one grows
  // faster, so the smaller of the two is the right one (4 VALU instructions instead of 6)
  const uint32_t qf = static_cast<uint32_t>(xs);
  const TransferBucket e = tbl[min(qf, (qf >> s.coarse_shift) + s.offset)];
  return e.base + (xs >= e.edge ? 1u : 0u);
}

__device__ __forceinline__ uint32_t half_px(const TransferBucketLinear *__restrict__ dec, float dn, const Matrix &m,
                                            const TransferBucket *__restrict__ enc, const SplitIndex &es, float en,
                                            float y00, float y01,
                                            float y10, float y11, const Chroma &c, uint32_t alpha_word) {
  float x[12];  // r0..r3, g0..g3, b0..b3 of the four source pixels
  pixel_rgbs(m, y00, c, x[0], x[4], x[8]);
  pixel_rgbs(m, y01, c, x[1], x[5], x[9]);
  pixel_rgbs(m, y10, c, x[2], x[6], x[10]);
  pixel_rgbs(m, y11, c, x[3], x[7], x[11]);
  uint32_t q[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) q[i] = linear_index(dn, x[i]);
  float lin[12];
  // kHalfBatch buckets in flight at a time: enough to cover the LDS latency without costing occupancy
#pragma unroll
  for (int h = 0; h < 12 / kHalfBatch; ++h) {
    u32x4 e[kHalfBatch];
#pragma unroll
    for (int i = 0; i < kHalfBatch; ++i) e[i] = linear_fetch(dec, q[kHalfBatch * h + i]);
    if (kHalfBatch == 4) asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));  // one wait for the batch
    if (kHalfBatch == 6) asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4 % kHalfBatch]), "+v"(e[5 % kHalfBatch]));
#pragma unroll
    for (int i = 0; i < kHalfBatch; ++i) lin[kHalfBatch * h + i] = linear_select(e[i], x[kHalfBatch * h + i]);
  }
  const float *lr = lin, *lg = lin + 4, *lb = lin + 8;
  // (((a+b)+c)+d) * 0.25f, then scale into the encode table's domain (both exact powers of two)
  const float k = __fmul_rn(0.25f, en);
  const float mr = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lr[0], lr[1]), lr[2]), lr[3]), k);
  const float mg = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lg[0], lg[1]), lg[2]), lg[3]), k);
  const float mb = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lb[0], lb[1]), lb[2]), lb[3]), k);
  const uint32_t R = lookup_split(enc, es, mr);
  const uint32_t G = lookup_split(enc, es, mg);
  const uint32_t B = lookup_split(enc, es, mb);
  return pack_bgra(R, G, B, alpha_word);
}

}  // namespace

// (Forcing 8 waves/SIMD with __launch_bounds__(512, 8) spills 21 VGPRs and halves the speed;
// the natural 75-79 VGPRs = 6 waves/SIMD is the better point.)
template <bool NT, bool WIDE>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_half(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucketLinear *dec = reinterpret_cast<TransferBucketLinear *>(lds_raw);
  TransferBucket *enc = reinterpret_cast<TransferBucket *>(lds_raw + p.table_bytes);

  const FramePlanes f = frame_planes(p, blockIdx.z);
  const Matrix m = matrix_of(p);
  const float en = p.table2_scale, dn = p.table_scale;
  const SplitIndex es = {p.table2_split, p.table2_coarse, p.table2_offset, 127u - (__float_as_uint(p.table2_coarse) >> 23)};
  const uint32_t out_rows = p.height >> 1;
  const uint32_t orow_raw = blockIdx.y * blockDim.y + threadIdx.y;
  const uint32_t orow = min(orow_raw, out_rows - 1);
  const uint8_t *y0 = f.y + static_cast<size_t>(2 * orow) * p.y_stride;
  const uint8_t *y1 = y0 + p.y_stride;
  const uint8_t *cc = f.cbcr + static_cast<size_t>(orow) * p.cbcr_stride;
  uint8_t *o = f.out + static_cast<size_t>(orow) * p.out_stride;

  if (WIDE) {
    constexpr int UNROLL = kQuadsPerLane;
    const uint32_t quads = p.width >> 2;
    const uint32_t q0 = blockIdx.x * (blockDim.x * UNROLL) + threadIdx.x;
    uint32_t ya[UNROLL], yb[UNROLL], cw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t q = min(q0 + u * blockDim.x, quads - 1);  // clamped load, predicated store (see 1:1 kernel)
      ya[u] = load32<NT>(y0 + 4 * q);
      yb[u] = load32<NT>(y1 + 4 * q);
      cw[u] = load32<NT>(cc + 4 * q);
    }
    stage_table(dec, p.table, p.table_bytes);
    stage_table(enc, p.table2, p.table2_bytes);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("" : "+v"(ya[u]), "+v"(yb[u]), "+v"(cw[u]));  // see 1:1 kernel
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t q = q0 + u * blockDim.x;
      const Chroma c0 = chroma_terms(m, byte_of(cw[u], 0), byte_of(cw[u], 1));
      const Chroma c1 = chroma_terms(m, byte_of(cw[u], 2), byte_of(cw[u], 3));
      u32x2 v;
      v.x = half_px(dec, dn, m, enc, es, en, byte_of(ya[u], 0), byte_of(ya[u], 1), byte_of(yb[u], 0), byte_of(yb[u], 1), c0,
                    p.alpha_word);
      v.y = half_px(dec, dn, m, enc, es, en, byte_of(ya[u], 2), byte_of(ya[u], 3), byte_of(yb[u], 2), byte_of(yb[u], 3), c1,
                    p.alpha_word);
      if (q < quads && orow_raw < out_rows) store8<NT>(o + 8 * q, v);
    }
  } else {
    stage_table(dec, p.table, p.table_bytes);
    stage_table(enc, p.table2, p.table2_bytes);
    __syncthreads();
    const uint32_t out_w = p.width >> 1;
    for (uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x; ox < out_w && orow_raw < out_rows;
         ox += gridDim.x * blockDim.x) {
      const Chroma c = chroma_terms(m, static_cast<float>(cc[2 * ox]), static_cast<float>(cc[2 * ox + 1]));
      reinterpret_cast<uint32_t *>(o)[ox] =
          half_px(dec, dn, m, enc, es, en, static_cast<float>(y0[2 * ox]), static_cast<float>(y0[2 * ox + 1]),
                  static_cast<float>(y1[2 * ox]), static_cast<float>(y1[2 * ox + 1]), c, p.alpha_word);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused decode + exact 2:1 downscale, CONFLICT-FREE form: same arithmetic and the same bytes out
// as decode_nv12_half<*, WIDE>, which is LDS-bound (12 + 3 random bucket lookups per output
// pixel; on random content ~2.4 LDS cycles per conflict-free one).  Here the decode-side table
// sits in LDS in R = 16 interleaved copies, entry q of copy c at byte (q * R + c) * 16, and lane
// l reads copy l & 15: the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...:
// MI355X_MICROARCH.md, LDS) then hit 16 different 16-byte bank groups whatever their q, so each
// lookup costs its 4 LDS cycles and no more.  The sRGB-encode table gets the copies that still
// fit (4 for the default gamma: 131 + 24 KiB of the CU's 160).  One workgroup per CU can hold
// that, so workgroups are PERSISTENT: the tables are staged once per launch, then workgroup w
// walks tile rows w, w + G, w + 2G, ... (G = gridDim.x; at any moment the CUs work on
// neighbouring row pairs, i.e. the DRAM stream stays address-ordered) with the loads of the
// next PF tile rows already in flight.  A tile row = blockDim.x quads of one row pair; the
// cursor (tile, row pair, frame) advances by G decomposed on the host: no division in the loop.
//
// Index arithmetic (this kernel is VALU-bound once the conflicts are gone, so it is counted in
// cycles, see the VALU budget note at the top): R, G, B in x units, saturated by the clamp bit of
// their last add; one round-toward-zero add of M = 2^23 / N gives bits(M) + floor(x N)
// (magic_floor12); v_lshl_add_u32 scales that to the entry's byte offset, adds the lane's copy
// offset and cancels bits(M): 2 + 4 cycles per index instead of min + convert + and-or + add
// (16).  Bucket edges are brought back to x units while staging.
// ---------------------------------------------------------------------------
namespace {

struct RepLookup {
  float magic;         // 2^23 / N: x + magic, rounded toward zero, has bit pattern bits(magic) + floor(x N)
  uint32_t dec_shift;  // log2(16 R)
  uint32_t dec_off;    // (lane & (R - 1)) * 16 - (bits(magic) << dec_shift): cancels the constant term
  uint32_t enc_shift;  // log2(8 * copies of table2)
  uint32_t enc_lane;   // LDS offset of table2 + (lane & (copies - 1)) * 8
};

struct TileCursor {
  uint32_t tx, rp, f;
};

struct QuadIn {
  uint32_t ya, yb, cw;
};

__device__ __forceinline__ void advance(TileCursor &c, const DecodeParams &p, uint32_t row_pairs) {
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

template <bool NT>
__device__ __forceinline__ QuadIn load_quad(const DecodeParams &p, const TileCursor &c, uint32_t quads) {
  const FramePlanes f = frame_planes(p, c.f);
  const uint8_t *y0 = f.y + static_cast<size_t>(2 * c.rp) * p.y_stride;
  const uint8_t *cc = f.cbcr + static_cast<size_t>(c.rp) * p.cbcr_stride;
  const uint32_t q = min(c.tx * blockDim.x + threadIdx.x, quads - 1);  // clamped load, predicated store
  QuadIn in;
  in.ya = load32<NT>(y0 + 4 * q);
  in.yb = load32<NT>(y0 + p.y_stride + 4 * q);
  in.cw = load32<NT>(cc + 4 * q);
  return in;
}

__device__ __forceinline__ uint32_t lookup_split_rep(const unsigned char *lds, const RepLookup &r, const SplitIndex &s,
                                                     float xs) {
  // fine index below the split, coarse above: the two index functions cross exactly at the split
  // (fine grows ratio times faster), so the smaller one is always the right one: 4 VALU
  // instructions instead of convert, multiply, convert, add, compare, select
  const uint32_t qf = static_cast<uint32_t>(xs);
  const uint32_t q = min(qf, (qf >> s.coarse_shift) + s.offset);
  const u32x2 e = *reinterpret_cast<LdsBucketPtr>((q << r.enc_shift) + r.enc_lane);
  return e.y + (xs >= __uint_as_float(e.x) ? 1u : 0u);
}

typedef __attribute__((address_space(3))) const u32x4 *LdsLinearPtr;

// x = saturate(yv + chroma terms) in x units, exactly the reference's value (BT709.h:424-446)
__device__ __forceinline__ void pixel_rgb_sat(const Matrix &m, float ybyte, const Chroma &c, float &r, float &g, float &b) {
  const float yv = __fmul_rn(centre_norm(ybyte, 16.0f), m.y);
  r = add_sat(yv, c.cr_r);
  g = add_sat(__fadd_rn(yv, c.cb_g), c.cr_g);
  b = add_sat(yv, c.cb_b);
}

__device__ __forceinline__ uint32_t half_px_rep(const unsigned char *lds, const RepLookup &r, const Matrix &m,
                                                const SplitIndex &es, float en, float y00, float y01, float y10,
                                                float y11, const Chroma &c, uint32_t alpha_word) {
  float x[12];  // r0..r3, g0..g3, b0..b3 of the four source pixels, saturated, in x units
  pixel_rgb_sat(m, y00, c, x[0], x[4], x[8]);
  pixel_rgb_sat(m, y01, c, x[1], x[5], x[9]);
  pixel_rgb_sat(m, y10, c, x[2], x[6], x[10]);
  pixel_rgb_sat(m, y11, c, x[3], x[7], x[11]);
  uint32_t t[12];
  magic_floor12(x, t, r.magic);
  float lin[12];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u32x4 e[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = *reinterpret_cast<LdsLinearPtr>((t[6 * h + i] << r.dec_shift) + r.dec_off);
    asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]));  // one wait per batch
#pragma unroll
    for (int i = 0; i < 6; ++i) lin[6 * h + i] = linear_select(e[i], x[6 * h + i]);
  }
  const float *lr = lin, *lg = lin + 4, *lb = lin + 8;
  const float k = __fmul_rn(0.25f, en);
  const float mr = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lr[0], lr[1]), lr[2]), lr[3]), k);
  const float mg = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lg[0], lg[1]), lg[2]), lg[3]), k);
  const float mb = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lb[0], lb[1]), lb[2]), lb[3]), k);
  const uint32_t R = lookup_split_rep(lds, r, es, mr);
  const uint32_t G = lookup_split_rep(lds, r, es, mg);
  const uint32_t B = lookup_split_rep(lds, r, es, mb);
  return pack_bgra(R, G, B, alpha_word);
}

}  // namespace

template <bool NT, int U>
__global__ void __launch_bounds__(kRepBlockThreads)
decode_nv12_half_rep(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t tid = threadIdx.x;
  const uint32_t row_pairs = p.height >> 1, quads = p.width >> 2;
  const uint32_t G = gridDim.x;

  uint32_t t = blockIdx.x;  // < tile_rows (the launcher never starts more workgroups than tile rows)
  TileCursor pre;
  pre.tx = t % p.tiles_x;
  pre.rp = (t / p.tiles_x) % row_pairs;
  pre.f = (t / p.tiles_x) / row_pairs;
  TileCursor cur = pre;

  // A step is U tile rows t, t + G, ...: their loads are issued one whole step ahead (first ones:
  // before the tables are staged).  Past the end of the launch a slot repeats the step's first tile
  // row -- same loads, same result, same store -- so every step is exactly 3U loads and U stores.
  QuadIn in[U];
  {
    const TileCursor first = pre;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool have = t + u * G < p.tile_rows;
      const TileCursor c = {have ? pre.tx : first.tx, have ? pre.rp : first.rp, have ? pre.f : first.f};
      in[u] = load_quad<NT>(p, c, quads);
      advance(pre, p, row_pairs);
    }
  }

  const uint32_t r1 = p.rep_dec_log2, r2 = p.rep_enc_log2;
  const float inv_n = __uint_as_float(0x7f000000u - __float_as_uint(p.table_scale));  // 1 / N for N = 2^k, no division
  {
    u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
    const u32x4 *src = reinterpret_cast<const u32x4 *>(p.table);
    const uint32_t n = (p.table_bytes / 16) << r1;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
      u32x4 e = src[i >> r1];
      e.x = __float_as_uint(__fmul_rn(__uint_as_float(e.x), inv_n));  // edge back into x units (inf stays inf)
      d[i] = e;
    }
    u32x2 *d2 = reinterpret_cast<u32x2 *>(lds_raw + (static_cast<size_t>(p.table_bytes) << r1));
    const u32x2 *src2 = reinterpret_cast<const u32x2 *>(p.table2);
    const uint32_t n2 = (p.table2_bytes / 8) << r2;
    for (uint32_t i = tid; i < n2; i += blockDim.x) d2[i] = src2[i >> r2];
  }
  __syncthreads();

  const uint32_t lds_base =
      static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)lds_raw));
  RepLookup r;
  r.magic = p.unit_magic;
  r.dec_shift = 4u + r1;
  r.dec_off = lds_base + (tid & ((1u << r1) - 1u)) * 16u - (__float_as_uint(r.magic) << r.dec_shift);
  r.enc_shift = 3u + r2;
  r.enc_lane = lds_base + (p.table_bytes << r1) + (tid & ((1u << r2) - 1u)) * 8u;
  const Matrix m = {kMY, kMCrR, kMCbG, kMCrG, kMCbB};
  const float en = p.table2_scale;
  const SplitIndex es = {p.table2_split, p.table2_coarse, p.table2_offset, 127u - (__float_as_uint(p.table2_coarse) >> 23)};

  for (; t < p.tile_rows; t += U * G) {
    // next step's loads first: they have this step's arithmetic (and the other waves') to arrive
    QuadIn nx[U];
    {
      const TileCursor first = pre;  // valid or not: only dereferenced when t + U * G < tile_rows
      const bool any = t + U * G < p.tile_rows;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool have = t + (U + u) * G < p.tile_rows;
        const TileCursor a = have ? pre : first;
        const TileCursor c = {any ? a.tx : cur.tx, any ? a.rp : cur.rp, any ? a.f : cur.f};
        nx[u] = load_quad<NT>(p, c, quads);
        advance(pre, p, row_pairs);
      }
    }
    const TileCursor first = cur;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool have = t + u * G < p.tile_rows;
      const TileCursor c = {have ? cur.tx : first.tx, have ? cur.rp : first.rp, have ? cur.f : first.f};
      const Chroma c0 = chroma_terms(m, byte_of(in[u].cw, 0), byte_of(in[u].cw, 1));
      const Chroma c1 = chroma_terms(m, byte_of(in[u].cw, 2), byte_of(in[u].cw, 3));
      u32x2 v;
      v.x = half_px_rep(lds_raw, r, m, es, en, byte_of(in[u].ya, 0), byte_of(in[u].ya, 1), byte_of(in[u].yb, 0),
                        byte_of(in[u].yb, 1), c0, p.alpha_word);
      v.y = half_px_rep(lds_raw, r, m, es, en, byte_of(in[u].ya, 2), byte_of(in[u].ya, 3), byte_of(in[u].yb, 2),
                        byte_of(in[u].yb, 3), c1, p.alpha_word);
      const FramePlanes f = frame_planes(p, c.f);
      uint8_t *o = f.out + static_cast<size_t>(c.rp) * p.out_stride;
      // lanes past the row's end loaded the last quad (clamp), hold its result and store it again
      const uint32_t q = min(c.tx * blockDim.x + tid, quads - 1);
      store8<NT>(o + 8 * q, v);
      advance(cur, p, row_pairs);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) in[u] = nx[u];  // hipcc waits here for the loads issued at the top (not for the stores)
  }
}

// ---------------------------------------------------------------------------
// Fused decode + bilinear rescale to ANY output size (pass 1 + MetalScaleRenderContext
// -renderScaled:, AAPLShaders.metal:73-85, for a view that is not an exact 2:1 of the frame).
// Two-pass-equivalent definition ("rescale"; parity unpinned by the reference):
//   sx = (ox + 0.5f) * (W / OW) - 0.5f,  x0 = floor(sx), fx = sx - x0, taps clamped to the edge
//   (same in y); each tap is decoded to its 8-bit sRGB value and linearised as the sRGB8 sampler
//   does; v = (((w00*l00 + w01*l01) + w10*l10) + w11*l11) with w00 = (1-fx)(1-fy), ...;
//   sRGB-encode, quantise.  For an exact 2:1 ratio every weight is 0.25 and this is bit for bit
//   the decode_nv12_half result.
// One lane per output pixel, byte gathers (cached), 4-byte coalesced stores; grid =
// (ceil(OW / blockDim), OH, frames).  Not a bandwidth kernel: 12 LDS bucket lookups per pixel.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlockThreads)
decode_nv12_scaled(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucketLinear *dec = reinterpret_cast<TransferBucketLinear *>(lds_raw);
  TransferBucket *enc = reinterpret_cast<TransferBucket *>(lds_raw + p.table_bytes);
  stage_table(dec, p.table, p.table_bytes);
  stage_table(enc, p.table2, p.table2_bytes);
  __syncthreads();

  const FramePlanes f = frame_planes(p, blockIdx.z);
  const Matrix m = matrix_of(p);
  const float en = p.table2_scale, dn = p.table_scale;
  const SplitIndex es = {p.table2_split, p.table2_coarse, p.table2_offset, 127u - (__float_as_uint(p.table2_coarse) >> 23)};
  const uint32_t oy = blockIdx.y;
  const uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x;
  if (ox >= p.out_width) return;

  const float sx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(ox), 0.5f), p.scale_x), -0.5f);
  const float sy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(oy), 0.5f), p.scale_y), -0.5f);
  const float x0f = __builtin_floorf(sx), y0f = __builtin_floorf(sy);
  const float fx = __fadd_rn(sx, -x0f), fy = __fadd_rn(sy, -y0f);
  const int wmax = static_cast<int>(p.width) - 1, hmax = static_cast<int>(p.height) - 1;
  const int xi = static_cast<int>(x0f), yi = static_cast<int>(y0f);
  const int xs[2] = {min(max(xi, 0), wmax), min(max(xi + 1, 0), wmax)};
  const int ys[2] = {min(max(yi, 0), hmax), min(max(yi + 1, 0), hmax)};
  const float gx = __fadd_rn(1.0f, -fx), gy = __fadd_rn(1.0f, -fy);
  const float w[4] = {__fmul_rn(gx, gy), __fmul_rn(fx, gy), __fmul_rn(gx, fy), __fmul_rn(fx, fy)};

  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int x = xs[t & 1], y = ys[t >> 1];
    const float yb = static_cast<float>(f.y[static_cast<size_t>(y) * p.y_stride + x]);
    const uint8_t *c = f.cbcr + static_cast<size_t>(y >> 1) * p.cbcr_stride + 2 * (x >> 1);
    const Chroma ch = chroma_terms(m, static_cast<float>(c[0]), static_cast<float>(c[1]));
    float v[3];
    pixel_rgbs(m, yb, ch, v[0], v[1], v[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t q = linear_index(dn, v[k]);
      const u32x4 e = linear_fetch(dec, q);
      const float term = __fmul_rn(w[t], linear_select(e, v[k]));
      acc[k] = t ? __fadd_rn(acc[k], term) : term;
    }
  }
  const uint32_t R = lookup_split(enc, es, __fmul_rn(acc[0], en));
  const uint32_t G = lookup_split(enc, es, __fmul_rn(acc[1], en));
  const uint32_t B = lookup_split(enc, es, __fmul_rn(acc[2], en));
  reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(oy) * p.out_stride)[ox] = pack_bgra(R, G, B, p.alpha_word);
}

// ---------------------------------------------------------------------------
// host-callable launchers (no HIP types in the signature beyond hipStream_t)
// ---------------------------------------------------------------------------
#ifndef BT709_REP_PREFETCH
#define BT709_REP_PREFETCH 2  // tile rows per step of the persistent rescale kernel (loads run one step ahead)
#endif

const char *launch_decode(const DecodeParams &p_in, int frames, int variant, bool has_alpha, bool nontemporal,
                          uint32_t grid_x, uint32_t block_threads, hipStream_t stream) {
  DecodeParams p = p_in;
  p.unit_magic = 8388608.0f / p.table_scale;  // 2^23 / N, exact: N is a power of two
#if defined(BT709_LAB_LDS_CHROMA)
  const size_t lds = p.table_bytes + 4 * block_threads * kQuadsPerLane;  // + the staged CbCr segment
#else
  size_t lds = (variant == kVariantQuads && BT709_OPT_UNIT) ? p.table_unit_bytes : p.table_bytes;
#endif
#if defined(BT709_LAB_LDS_PAD)  // tools/decode_lab only: unused LDS to cap the workgroups resident per CU
  if (const char *pad = std::getenv("BT709_LAB_LDS_PAD")) lds += static_cast<size_t>(std::atoi(pad)) * 1024;
#endif
  if (variant == kVariantQuads) {
    // grid_x = tiles per row pair; narrow frames stack row pairs in blockDim.y
    const uint32_t by = quads_rows_per_block(block_threads, grid_x);
    const dim3 grid(grid_x, (p.height / 2 + by - 1) / by, static_cast<uint32_t>(frames));
    const dim3 block(block_threads, by, 1);
    if (has_alpha) {
      hipLaunchKernelGGL((decode_nv12_quads<true, true>), grid, block, lds, stream, p);
      return "decode_nv12_quads<alpha>";
    }
    if (nontemporal) {
      hipLaunchKernelGGL((decode_nv12_quads<false, true>), grid, block, lds, stream, p);
      return "decode_nv12_quads<nt>";
    }
    hipLaunchKernelGGL((decode_nv12_quads<false, false>), grid, block, lds, stream, p);
    return "decode_nv12_quads";
  }
  // grid_x = workgroups per frame, grid-strided over row pairs
  const dim3 grid(grid_x, static_cast<uint32_t>(frames), 1);
  const dim3 block(kBlockThreads, 1, 1);
  if (has_alpha) {
    hipLaunchKernelGGL((decode_nv12_blocks<true>), grid, block, lds, stream, p);
    return "decode_nv12_blocks<alpha>";
  }
  hipLaunchKernelGGL((decode_nv12_blocks<false>), grid, block, lds, stream, p);
  return "decode_nv12_blocks";
}

hipError_t prepare_kernels() {
  const int cap = 160 * 1024;  // gfx950: 160 KiB LDS per workgroup
  const void *fns[] = {
      reinterpret_cast<const void *>(&decode_nv12_half<true, true>),
      reinterpret_cast<const void *>(&decode_nv12_half<false, true>),
      reinterpret_cast<const void *>(&decode_nv12_half<false, false>),
      reinterpret_cast<const void *>(&decode_nv12_scaled),
      reinterpret_cast<const void *>(&decode_nv12_half_rep<true, BT709_REP_PREFETCH>),
      reinterpret_cast<const void *>(&decode_nv12_half_rep<false, BT709_REP_PREFETCH>),
      reinterpret_cast<const void *>(&decode_nv12_quads<true, true>),
      reinterpret_cast<const void *>(&decode_nv12_quads<false, true>),
      reinterpret_cast<const void *>(&decode_nv12_quads<false, false>),
      reinterpret_cast<const void *>(&decode_nv12_blocks<true>),
      reinterpret_cast<const void *>(&decode_nv12_blocks<false>),
  };
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

const char *launch_decode_half(const DecodeParams &p, int frames, bool wide, bool nontemporal, uint32_t grid_x,
                               uint32_t block_threads, hipStream_t stream) {
  const uint32_t by = wide ? quads_rows_per_block(block_threads, grid_x) : 1;
  const dim3 grid(grid_x, (p.height / 2 + by - 1) / by, static_cast<uint32_t>(frames));
  const dim3 block(block_threads, by, 1);
  const size_t lds = static_cast<size_t>(p.table_bytes) + p.table2_bytes;
  if (wide) {
    if (nontemporal) hipLaunchKernelGGL((decode_nv12_half<true, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((decode_nv12_half<false, true>), grid, block, lds, stream, p);
    return "decode_nv12_half<wide>";
  }
  hipLaunchKernelGGL((decode_nv12_half<false, false>), grid, block, lds, stream, p);
  return "decode_nv12_half<narrow>";
}

const char *launch_decode_half_rep(const DecodeParams &p_in, int frames, bool nontemporal, uint32_t workgroups,
                                   hipStream_t stream) {
  DecodeParams p = p_in;
  p.unit_magic = 8388608.0f / p.table_scale;  // 2^23 / N, exact
  // copies: as many as fit the CU's LDS, decode side first (12 of the 15 lookups per output pixel)
  uint32_t r1 = 4, r2 = 0;
  while (r1 > 0 && (static_cast<uint64_t>(p.table_bytes) << r1) + p.table2_bytes > kRepLdsBytes) --r1;
  if ((static_cast<uint64_t>(p.table_bytes) << r1) + p.table2_bytes > kRepLdsBytes) return nullptr;
  while (r2 < 5 && (static_cast<uint64_t>(p.table_bytes) << r1) + (static_cast<uint64_t>(p.table2_bytes) << (r2 + 1)) <=
                       kRepLdsBytes)
    ++r2;
  p.rep_dec_log2 = r1;
  p.rep_enc_log2 = r2;
  const uint32_t quads = p.width / 4, row_pairs = p.height / 2;
  p.tiles_x = (quads + kRepBlockThreads - 1) / kRepBlockThreads;
  uint32_t threads = ((quads + p.tiles_x - 1) / p.tiles_x + 63) / 64 * 64;
  const uint64_t total = static_cast<uint64_t>(p.tiles_x) * row_pairs * static_cast<uint32_t>(frames);
  if (total == 0 || total > 0x7fffffffu) return nullptr;
  p.tile_rows = static_cast<uint32_t>(total);
  if (workgroups > p.tile_rows) workgroups = p.tile_rows;
  if (workgroups == 0) workgroups = 1;
  p.cursor_tx = workgroups % p.tiles_x;
  p.cursor_rp = (workgroups / p.tiles_x) % row_pairs;
  p.cursor_f = (workgroups / p.tiles_x) / row_pairs;
  const size_t lds = (static_cast<size_t>(p.table_bytes) << r1) + (static_cast<size_t>(p.table2_bytes) << r2);
  if (nontemporal)
    hipLaunchKernelGGL((decode_nv12_half_rep<true, BT709_REP_PREFETCH>), dim3(workgroups), dim3(threads), lds, stream, p);
  else
    hipLaunchKernelGGL((decode_nv12_half_rep<false, BT709_REP_PREFETCH>), dim3(workgroups), dim3(threads), lds, stream, p);
  return "decode_nv12_half_rep";
}

const char *launch_decode_scaled(const DecodeParams &p, int frames, hipStream_t stream) {
  const dim3 grid((p.out_width + kBlockThreads - 1) / kBlockThreads, p.out_height, static_cast<uint32_t>(frames));
  const size_t lds = static_cast<size_t>(p.table_bytes) + p.table2_bytes;
  hipLaunchKernelGGL(decode_nv12_scaled, grid, dim3(kBlockThreads), lds, stream, p);
  return "decode_nv12_scaled";
}

}  // namespace bt709
