// CDNA4 (gfx950) kernel of the step BEFORE the decode path: 8-bit BGRA -> 4:2:0 NV12
// BT.709 video range, with the reference's linear-light 2x2 chroma averaging.
//
// Restates, per 2x2 block, cvpbu_ycbcr_subsample (Renderer/CVPixelBufferUtils.h:241-399)
// -> BT709_average_pixel_values (Renderer/BT709.h:1349-1509):
//   lin[i][c]  = BT709_tolinearNorm(byte)                      BT709.h:1100-1146   (LUT, exact)
//   ave[c]     = (((l0 + l1) + l2) + l3) / 4.0f                BT709.h:1171-1190, 1404-1406
//   avgByte[c] = BT709_from_linear(ave[c], outputGamma)         BT709.h:1150-1167   (bucket table, exact)
//   (., Cb, Cr) = sRGB_from_sRGB_convertRGBToYCbCr(avgByte)     BT709.h:914-944 -> 199-268
//   Y[i]       = sRGB_from_sRGB_convertRGBToYCbCr(BT709_from_linear(lin[i][c], outputGamma))[0]
//                                                               BT709.h:1423-1487   (per-byte LUT, exact)
// with the matrix of BT709.h:222-246: Ey = (Kr*R + Kg*G) + Kb*B, Eb = (B-Ey)/1.8556f,
// Er = (R-Ey)/1.5748f, Y = round(Ey*219 + 16), C = round(E*224 + 128).
//
// Exactness notes:
//   * the per-byte LUT holds {lin, byteNorm(encoded byte)} -- one 2 KiB table for the three channels
//     (a 6 KiB version with the K_c products folded in saved three 2-cycle multiplies per pixel and
//     cost 4 KiB more staging per workgroup);
//   * x / c for the two constant divisors is computed as q0 = x*rc, q = fma(fma(-c, q0, x), rc, q0)
//     with rc = fl(1/c): this equals the correctly rounded quotient for EVERY float with
//     1e-30 <= |x| <= 4 (exhaustive check over all 2^32 patterns: tools/div_exact.hip, run by
//     tests/test_encoder.py); operands here are in [-1.1, 1.1], and a zero operand only differs in
//     the sign of zero, which the following "+ 128" erases.  These two FMAs are the division's own
//     error-correction steps, not a contracted multiply-add of the reference's expression;
//   * round() of a positive float v < 256 is trunc(v + 0.5f), and v + 0.5f is exact there (a sum
//     that moves into the next binade lands in [2^m, 2^m + 1/2), where rounding cannot reach the
//     next integer).  v_cvt_pk_u8_f32 converts AND places the byte in its lane of the output word;
//     it rounds to nearest even by default (so 209 / 193 / 283 of the 2^24 Y / Cb / Cr inputs, the
//     exact ties, would come out one low) but honours MODE.fp_round (tools/cvt_probe.hip), so the
//     twelve conversions of a quad sit in one asm statement under round-toward-zero.
//
// VALU budget (tools/valu_ops.hip): only v_add/mul/mov_f32 are 2-cycle instructions on gfx950,
// everything else this kernel uses costs 4, and at 1 Tpixel/s the VALU is ~70 % busy: the byte
// -> LUT address is one SDWA shift (byte select + << 3, table bases in the ds_read offset field)
// instead of bfe + shift-add, the split-table index is a min() of the two index functions.
//
// A lane owns a 4x2-pixel quad (two blocks): two 16-byte loads, three 4-byte stores; lanes
// of a wave are consecutive quads of one row pair; grid = (tiles, row-pair groups, frames).  LDS holds
// the 2 KiB per-byte table and the two-resolution BT709_from_linear table.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bt709_constants.h"
#include "bt709_kernels.h"
#include "bt709_stage.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr float kRcCb = 1.0f / kCbSpan;  // fl(1/1.8556f)
constexpr float kRcCr = 1.0f / kCrSpan;  // fl(1/1.5748f)

__device__ __forceinline__ float div_const(float x, float c, float rc) {
  const float q0 = __fmul_rn(x, rc);
  return __fmaf_rn(__fmaf_rn(-c, q0, x), rc, q0);
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 *LdsPairPtr;  // one ds_read_b64, base in the offset field

struct EncodeLds {
  // The per-byte table sits at LDS address 0: these kernels have no static LDS, so the dynamic
  // segment starts at 0 -- stage_encode_tables traps if that ever changes.
  uint32_t fl;     // ... of the two-resolution BT709_from_linear table (transfer_tables.h SplitTable)
  uint32_t offset, coarse_shift;
  uint32_t three;  // VGPR holding 3: SDWA operands cannot be inline constants
};

// {lin, enc_norm} of byte LANE of a BGRA word: v_lshlrev_b32_sdwa selects the byte and scales it to
// the 8-byte entry in one instruction
template <int LANE>
__device__ __forceinline__ u32x2 byte_entry(const EncodeLds &t, uint32_t word) {
  uint32_t a;
  if (LANE == 0)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(t.three), "v"(word));
  else if (LANE == 1)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(t.three), "v"(word));
  else
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(t.three), "v"(word));
  return *reinterpret_cast<LdsPairPtr>(a);
}

__device__ __forceinline__ uint32_t from_linear(const EncodeLds &t, float xs) {
  // fine index below the split, coarse above; the two index functions cross at the split and the
  // fine one grows faster, so the smaller is the right one
  const uint32_t qf = static_cast<uint32_t>(xs);
  const uint32_t q = min(qf, (qf >> t.coarse_shift) + t.offset);
  const u32x2 e = *reinterpret_cast<LdsPairPtr>((q << 3) + t.fl);
  return e.y + (xs >= __uint_as_float(e.x) ? 1u : 0u);
}

// v = e * scale + offset as the reference's float expression, plus the exact 0.5f: trunc(result)
// is (int)round(v)
__device__ __forceinline__ float quant_arg(float e, float scale, float offset) {
  return __fadd_rn(__fadd_rn(__fmul_rn(e, scale), offset), 0.5f);
}

// one 2x2 block: p = {top-left, top-right, bottom-left, bottom-right} BGRA words;
// v = {Y tl, Y tr, Y bl, Y br, Cb, Cr} ready for truncation
__device__ __forceinline__ void encode_block(const EncodeLds &t, float fl_quarter_n, const uint32_t p[4], float v[6]) {
  float sr = 0.f, sg = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x2 r = byte_entry<2>(t, p[i]);
    const u32x2 g = byte_entry<1>(t, p[i]);
    const u32x2 b = byte_entry<0>(t, p[i]);
    sr = i ? __fadd_rn(sr, __uint_as_float(r.x)) : __uint_as_float(r.x);
    sg = i ? __fadd_rn(sg, __uint_as_float(g.x)) : __uint_as_float(g.x);
    sb = i ? __fadd_rn(sb, __uint_as_float(b.x)) : __uint_as_float(b.x);
    const float ey = __fadd_rn(__fadd_rn(__fmul_rn(kKr, __uint_as_float(r.y)), __fmul_rn(kKg, __uint_as_float(g.y))),
                               __fmul_rn(kKb, __uint_as_float(b.y)));  // BT709.h:222
    v[i] = quant_arg(ey, static_cast<float>(kYMax - kYMin), 16.0f);                                            // BT709.h:233, 244
  }
  // ave = sum / 4.0f, then scaled into the table's domain: sum * (0.25 * N), both powers of two
  const float rn = __fmul_rn(static_cast<float>(from_linear(t, __fmul_rn(sr, fl_quarter_n))), kInv255);
  const float gn = __fmul_rn(static_cast<float>(from_linear(t, __fmul_rn(sg, fl_quarter_n))), kInv255);
  const float bn = __fmul_rn(static_cast<float>(from_linear(t, __fmul_rn(sb, fl_quarter_n))), kInv255);
  const float ey = __fadd_rn(__fadd_rn(__fmul_rn(kKr, rn), __fmul_rn(kKg, gn)), __fmul_rn(kKb, bn));
  const float eb = div_const(__fadd_rn(bn, -ey), kCbSpan, kRcCb);                      // BT709.h:223
  const float er = div_const(__fadd_rn(rn, -ey), kCrSpan, kRcCr);                      // BT709.h:224
  v[4] = quant_arg(eb, static_cast<float>(kCMax - kCMin), 128.0f);                     // BT709.h:234, 245
  v[5] = quant_arg(er, static_cast<float>(kCMax - kCMin), 128.0f);                     // BT709.h:235, 246
}

// Truncate the twelve values of a quad (two blocks) and place them: Y rows top / bottom, CbCr with
// Cb in the low byte (CVPixelBufferUtils.h:358-361).  One asm statement: the conversions must not
// be separated from the mode switch around them.
__device__ __forceinline__ void quantize_quad(const float a[6], const float b[6], uint32_t &ytop, uint32_t &ybot,
                                              uint32_t &cbcr) {
  ytop = ybot = cbcr = 0;
  asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 0, %0\n\tv_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %5, 0, %1\n\tv_cvt_pk_u8_f32 %1, %6, 1, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %7, 0, %2\n\tv_cvt_pk_u8_f32 %2, %8, 1, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %9, 2, %0\n\tv_cvt_pk_u8_f32 %0, %10, 3, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %11, 2, %1\n\tv_cvt_pk_u8_f32 %1, %12, 3, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\tv_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "+v"(ytop), "+v"(ybot), "+v"(cbcr)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
        "v"(b[4]), "v"(b[5]));
}

// one block (general kernel): bytes 0,1 of each word
__device__ __forceinline__ void quantize_block(const float a[6], uint32_t &ytop, uint32_t &ybot, uint32_t &cbcr) {
  ytop = ybot = cbcr = 0;
  asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 0, %0\n\tv_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %5, 0, %1\n\tv_cvt_pk_u8_f32 %1, %6, 1, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %7, 0, %2\n\tv_cvt_pk_u8_f32 %2, %8, 1, %2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "+v"(ytop), "+v"(ybot), "+v"(cbcr)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]));
}

__device__ __forceinline__ EncodeLds stage_encode_tables(unsigned char *lds_raw, const EncodeParams &p) {
  u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
  const u32x4 *sb = reinterpret_cast<const u32x4 *>(p.per_byte);
  const u32x4 *sf = reinterpret_cast<const u32x4 *>(p.from_linear);
  const uint32_t nb = 256 * sizeof(EncodeByteEntry) / 16, nf = p.from_linear_bytes / 16;
  // (The first version staged a uniform 33 KiB BT709_from_linear table here -- ten dependent L2
  // round trips per workgroup -- and a version that left it in global memory was bound by the
  // 64-line gathers; the two-resolution table is 6-10 KiB.)
  stage_batched(d, nb + nf, threadIdx.x, blockDim.x, [&](uint32_t i) { return i < nb ? sb[i] : sf[i - nb]; });
  EncodeLds t;
  if (static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)lds_raw)) != 0u)
    __builtin_trap();
  t.fl = 256u * static_cast<uint32_t>(sizeof(EncodeByteEntry));
  t.offset = p.from_linear_offset;
  t.coarse_shift = 127u - (__float_as_uint(p.from_linear_coarse) >> 23);  // log2(1 / coarse), coarse = 2^-k
  t.three = 3u;
  asm("" : "+v"(t.three));
  return t;
}

__device__ __forceinline__ EncodeFrame encode_frame(const EncodeParams &p, uint32_t i) {
  if (!p.uniform) return p.frames[i];
  EncodeFrame f = p.frames[0];
  f.bgra += static_cast<int64_t>(i) * p.step_bgra;
  f.y += static_cast<int64_t>(i) * p.step_y;
  f.cbcr += static_cast<int64_t>(i) * p.step_cbcr;
  return f;
}

}  // namespace

// grid = (tiles, row-pair groups, frames); a workgroup walks row_pairs_per_block consecutive row
// pairs of one frame, prefetching the next pair while it encodes the current one.
__global__ void __launch_bounds__(kMaxBlockThreads)
encode_bgra_nv12(const EncodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // XCD-aware work map (p.xcd_bands; launches of a multiple of 8 pictures): grid.x = 8 x tiles, x & 7 = the workgroup's place in
  // the round-robin over the XCDs, which owns a contiguous band of the launch's pictures (bt709_kernels.hip decode_nv12_quads)
  const uint32_t tile = p.xcd_bands ? blockIdx.x >> 3 : blockIdx.x;
  const EncodeFrame f = encode_frame(p, p.xcd_bands ? (blockIdx.x & 7u) * p.frames_per_band + blockIdx.z : blockIdx.z);
  const uint32_t quads = p.width >> 2;
  const uint32_t row_pairs = p.height >> 1;
  const uint32_t q_raw = tile * blockDim.x + threadIdx.x;
  const uint32_t q = min(q_raw, quads - 1);
  const uint32_t rp0 = blockIdx.y * p.row_pairs_per_block;
  const uint32_t rp_end = min(rp0 + p.row_pairs_per_block, row_pairs);

  // row pointers are uniform (SGPR base), the lane adds a 32-bit offset: no 64-bit VALU address arithmetic
  const uint8_t *s0 = f.bgra + static_cast<size_t>(2 * rp0) * p.bgra_stride;
  u32x4 top = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s0 + 16u * q));
  u32x4 bot = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s0 + p.bgra_stride + 16u * q));

  const EncodeLds t = stage_encode_tables(lds_raw, p);  // after the first loads are in flight
  __syncthreads();
  const float quarter_n = __fmul_rn(0.25f, p.from_linear_scale);

  for (uint32_t rp = rp0; rp < rp_end; ++rp) {
    // prefetch the next row pair before the arithmetic; nothing on the workgroup's last pair (a
    // clamped re-read of the own rows cost 8 % extra HBM reads at 9 row pairs per workgroup:
    // non-temporal loads do not stay in L2)
    u32x4 ntop = top, nbot = bot;
    if (rp + 1 < rp_end) {
      const uint8_t *s1 = f.bgra + static_cast<size_t>(2 * (rp + 1)) * p.bgra_stride;
      ntop = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s1 + 16u * q));
      nbot = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s1 + p.bgra_stride + 16u * q));
    }

    float va[6], vb[6];
    {
      const uint32_t blk[4] = {top.x, top.y, bot.x, bot.y};
      encode_block(t, quarter_n, blk, va);
    }
    {
      const uint32_t blk[4] = {top.z, top.w, bot.z, bot.w};
      encode_block(t, quarter_n, blk, vb);
    }
    uint32_t ytop, ybot, cbcr;
    quantize_quad(va, vb, ytop, ybot, cbcr);
    if (q_raw < quads) {
      uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
      __builtin_nontemporal_store(ytop, reinterpret_cast<uint32_t *>(y0 + 4u * q));
      __builtin_nontemporal_store(ybot, reinterpret_cast<uint32_t *>(y0 + p.y_stride + 4u * q));
      __builtin_nontemporal_store(
          cbcr, reinterpret_cast<uint32_t *>(f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride + 4u * q));
    }
    top = ntop;
    bot = nbot;
  }
}

// General layout: one lane per 2x2 block, scalar loads/stores, any alignment.
__global__ void __launch_bounds__(kBlockThreads)
encode_bgra_nv12_blocks(const EncodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const EncodeLds t = stage_encode_tables(lds_raw, p);
  __syncthreads();
  const EncodeFrame f = encode_frame(p, blockIdx.z);
  const float quarter_n = __fmul_rn(0.25f, p.from_linear_scale);
  const uint32_t bw = p.width >> 1;
  const uint32_t rp = blockIdx.y;
  for (uint32_t bx = blockIdx.x * blockDim.x + threadIdx.x; bx < bw; bx += gridDim.x * blockDim.x) {
    const uint32_t *r0 = reinterpret_cast<const uint32_t *>(f.bgra + static_cast<size_t>(2 * rp) * p.bgra_stride);
    const uint32_t *r1 = reinterpret_cast<const uint32_t *>(f.bgra + static_cast<size_t>(2 * rp + 1) * p.bgra_stride);
    const uint32_t blk[4] = {r0[2 * bx], r0[2 * bx + 1], r1[2 * bx], r1[2 * bx + 1]};
    float v[6];
    encode_block(t, quarter_n, blk, v);
    uint32_t ytop, ybot, cbcr;
    quantize_block(v, ytop, ybot, cbcr);
    uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    uint8_t *y1 = y0 + p.y_stride;
    uint8_t *c = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    y0[2 * bx] = static_cast<uint8_t>(ytop);
    y0[2 * bx + 1] = static_cast<uint8_t>(ytop >> 8);
    y1[2 * bx] = static_cast<uint8_t>(ybot);
    y1[2 * bx + 1] = static_cast<uint8_t>(ybot >> 8);
    c[2 * bx] = static_cast<uint8_t>(cbcr);
    c[2 * bx + 1] = static_cast<uint8_t>(cbcr >> 8);
  }
}

const char *launch_encode(const EncodeParams &params, int frames, bool fast, bool xcd_bands, hipStream_t stream) {
  if (fast && xcd_bands && params.uniform && frames > kXcdBandMinFrames && frames % 8 != 0) {
    // any count of 64 pictures or more: the XCD-aware map over the multiple of 8, the plain map over the rest (launch_decode)
    const int head = frames - frames % 8;
    launch_encode(params, head, fast, xcd_bands, stream);
    EncodeParams tail = params;
    tail.frames[0].bgra += static_cast<int64_t>(head) * tail.step_bgra;
    tail.frames[0].y += static_cast<int64_t>(head) * tail.step_y;
    tail.frames[0].cbcr += static_cast<int64_t>(head) * tail.step_cbcr;
    return launch_encode(tail, frames - head, fast, false, stream);
  }
  EncodeParams p = params;
  if (p.row_pairs_per_block == 0) p.row_pairs_per_block = encode_row_pairs_per_block(p.width, p.height, frames);
  const size_t lds = 256 * sizeof(EncodeByteEntry) + p.from_linear_bytes;
  if (fast) {
    const uint32_t quads = p.width / 4;
    uint32_t threads = p.block_threads ? p.block_threads : encode_block_threads(p.width);
    // One picture per launch (what a caller of +convertIntoCoreVideoBuffer: issues) is one under-filled generation of
    // workgroups: tiles of up to 512 lanes (3840 -> 2 x 512 instead of 3 x 320) ran 7-8 % faster there
    // (tools/encode_single_shapes.sh, round 3: 665 -> 718-722 Gpixel/s); batched launches keep the 320-lane tiles.
    if (p.block_threads == 0 && frames == 1) {
      const uint32_t tiles = (quads + 511) / 512;
      threads = ((quads + tiles - 1) / tiles + 63) / 64 * 64;
      if (threads < 64) threads = 64;
    }
    if (threads > static_cast<uint32_t>(kMaxBlockThreads)) threads = kMaxBlockThreads;
    dim3 grid((quads + threads - 1) / threads,
              (p.height / 2 + p.row_pairs_per_block - 1) / p.row_pairs_per_block, frames);
    if (xcd_bands && frames >= kXcdBandMinFrames && frames % 8 == 0) {
      p.xcd_bands = 1;
      p.frames_per_band = static_cast<uint32_t>(frames) / 8u;
      grid = dim3(grid.x * 8u, grid.y, p.frames_per_band);
    }
    hipLaunchKernelGGL(encode_bgra_nv12, grid, dim3(threads), lds, stream, p);
    return "encode_bgra_nv12";
  }
  const dim3 grid((p.width / 2 + kBlockThreads - 1) / kBlockThreads, p.height / 2, frames);
  hipLaunchKernelGGL(encode_bgra_nv12_blocks, grid, dim3(kBlockThreads), lds, stream, p);
  return "encode_bgra_nv12_blocks";
}

hipError_t prepare_encode_kernels() {
  const int cap = 160 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_bgra_nv12),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_bgra_nv12_blocks),
                             hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}

}  // namespace bt709
