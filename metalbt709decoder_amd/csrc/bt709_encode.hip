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
//   * the per-byte LUT already holds the float product K_c * byteNorm(encoded byte), so a pixel's
//     Ey is two adds;
//   * x / c for the two constant divisors is computed as q0 = x*rc, q = fma(fma(-c, q0, x), rc, q0)
//     with rc = fl(1/c): this equals the correctly rounded quotient for EVERY float with
//     1e-30 <= |x| <= 4 (exhaustive check over all 2^32 patterns: tools/div_exact.hip, run by
//     tests/test_encoder.py); operands here are in [-1.1, 1.1], and a zero operand only differs in
//     the sign of zero, which the following "+ 128" erases.  These two FMAs are the division's own
//     error-correction steps, not a contracted multiply-add of the reference's expression;
//   * round() of a positive float v is ((uint)(2v) + 1) >> 1 (floor(v + 1/2) in integers); the
//     doubling is folded into the constants (power of two: exact).
//
// A lane owns a 4x2-pixel quad (two blocks): two 16-byte loads, three 4-byte stores; lanes
// of a wave are consecutive quads of one row pair; grid = (tiles, row-pair groups, frames).  LDS holds
// the three 2 KiB per-byte tables and the two-resolution BT709_from_linear table.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bt709_constants.h"
#include "bt709_kernels.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr float kRcCb = 1.0f / kCbSpan;  // fl(1/1.8556f)
constexpr float kRcCr = 1.0f / kCrSpan;  // fl(1/1.5748f)

__device__ __forceinline__ float div_const(float x, float c, float rc) {
  const float q0 = __fmul_rn(x, rc);
  return __fmaf_rn(__fmaf_rn(-c, q0, x), rc, q0);
}

// (int)round((double)(e * scale + offset)) for a positive result, via the doubled value
__device__ __forceinline__ uint32_t quant2(float e, float scale2, float offset2) {
  const uint32_t t2 = static_cast<uint32_t>(__fadd_rn(__fmul_rn(e, scale2), offset2));  // floor(2v)
  return (t2 + 1u) >> 1;
}

struct EncodeLds {
  const EncodeByteEntry *r, *g, *b;
  const TransferBucket *fl;  // two-resolution table (transfer_tables.h SplitTable)
  float split, coarse;
  uint32_t offset;
};

__device__ __forceinline__ uint32_t from_linear(const EncodeLds &t, float xs) {
  const uint32_t qf = static_cast<uint32_t>(xs);
  const uint32_t qc = static_cast<uint32_t>(__fmul_rn(xs, t.coarse)) + t.offset;  // exact: power of two
  const TransferBucket e = t.fl[xs < t.split ? qf : qc];
  return e.base + (xs >= e.edge ? 1u : 0u);
}

// one 2x2 block: p = {top-left, top-right, bottom-left, bottom-right} BGRA words
__device__ __forceinline__ void encode_block(const EncodeLds &t, float fl_quarter_n, const uint32_t p[4], uint32_t y[4],
                                             uint32_t &cb, uint32_t &cr) {
  float sr = 0.f, sg = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const EncodeByteEntry r = t.r[(p[i] >> 16) & 0xff];
    const EncodeByteEntry g = t.g[(p[i] >> 8) & 0xff];
    const EncodeByteEntry b = t.b[p[i] & 0xff];
    sr = i ? __fadd_rn(sr, r.lin) : r.lin;
    sg = i ? __fadd_rn(sg, g.lin) : g.lin;
    sb = i ? __fadd_rn(sb, b.lin) : b.lin;
    const float ey = __fadd_rn(__fadd_rn(r.k_enc, g.k_enc), b.k_enc);                 // BT709.h:222
    y[i] = quant2(ey, 2.0f * static_cast<float>(kYMax - kYMin), 32.0f);               // BT709.h:233, 244
  }
  // ave = sum / 4.0f, then scaled into the table's domain: sum * (0.25 * N), both powers of two
  const float rn = __fmul_rn(static_cast<float>(from_linear(t, __fmul_rn(sr, fl_quarter_n))), kInv255);
  const float gn = __fmul_rn(static_cast<float>(from_linear(t, __fmul_rn(sg, fl_quarter_n))), kInv255);
  const float bn = __fmul_rn(static_cast<float>(from_linear(t, __fmul_rn(sb, fl_quarter_n))), kInv255);
  const float ey = __fadd_rn(__fadd_rn(__fmul_rn(kKr, rn), __fmul_rn(kKg, gn)), __fmul_rn(kKb, bn));
  const float eb = div_const(__fadd_rn(bn, -ey), kCbSpan, kRcCb);                      // BT709.h:223
  const float er = div_const(__fadd_rn(rn, -ey), kCrSpan, kRcCr);                      // BT709.h:224
  cb = quant2(eb, 2.0f * static_cast<float>(kCMax - kCMin), 256.0f);                   // BT709.h:234, 245
  cr = quant2(er, 2.0f * static_cast<float>(kCMax - kCMin), 256.0f);                   // BT709.h:235, 246
}

__device__ __forceinline__ EncodeLds stage_encode_tables(unsigned char *lds_raw, const EncodeParams &p) {
  u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
  const u32x4 *sb = reinterpret_cast<const u32x4 *>(p.per_byte);
  const u32x4 *sf = reinterpret_cast<const u32x4 *>(p.from_linear);
  const uint32_t nb = 3 * 256 * sizeof(EncodeByteEntry) / 16, nf = p.from_linear_bytes / 16;
  // (The first version staged a uniform 33 KiB BT709_from_linear table here -- ten dependent L2
  // round trips per workgroup -- and a version that left it in global memory was bound by the
  // 64-line gathers; the two-resolution table is 6-10 KiB.)
  for (uint32_t i = threadIdx.x; i < nb + nf; i += blockDim.x) d[i] = i < nb ? sb[i] : sf[i - nb];
  EncodeLds t;
  t.r = reinterpret_cast<const EncodeByteEntry *>(lds_raw);
  t.g = t.r + 256;
  t.b = t.r + 512;
  t.fl = reinterpret_cast<const TransferBucket *>(t.r + 768);
  t.split = p.from_linear_split;
  t.coarse = p.from_linear_coarse;
  t.offset = p.from_linear_offset;
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
  const EncodeFrame f = encode_frame(p, blockIdx.z);
  const uint32_t quads = p.width >> 2;
  const uint32_t row_pairs = p.height >> 1;
  const uint32_t q_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = min(q_raw, quads - 1);
  const uint32_t rp0 = blockIdx.y * p.row_pairs_per_block;
  const uint32_t rp_end = min(rp0 + p.row_pairs_per_block, row_pairs);

  const uint8_t *s0 = f.bgra + static_cast<size_t>(2 * rp0) * p.bgra_stride + 16 * static_cast<size_t>(q);
  u32x4 top = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s0));
  u32x4 bot = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s0 + p.bgra_stride));

  const EncodeLds t = stage_encode_tables(lds_raw, p);  // after the first loads are in flight
  __syncthreads();
  const float quarter_n = __fmul_rn(0.25f, p.from_linear_scale);

  for (uint32_t rp = rp0; rp < rp_end; ++rp) {
    // prefetch the next row pair (clamped: the last iteration re-reads its own rows) before the arithmetic
    const uint32_t rn = min(rp + 1, rp_end - 1);
    const uint8_t *s1 = f.bgra + static_cast<size_t>(2 * rn) * p.bgra_stride + 16 * static_cast<size_t>(q);
    const u32x4 ntop = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s1));
    const u32x4 nbot = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s1 + p.bgra_stride));

    uint32_t ya[4], yb[4], cb0, cr0, cb1, cr1;
    {
      const uint32_t blk[4] = {top.x, top.y, bot.x, bot.y};
      uint32_t y4[4];
      encode_block(t, quarter_n, blk, y4, cb0, cr0);
      ya[0] = y4[0], ya[1] = y4[1], yb[0] = y4[2], yb[1] = y4[3];
    }
    {
      const uint32_t blk[4] = {top.z, top.w, bot.z, bot.w};
      uint32_t y4[4];
      encode_block(t, quarter_n, blk, y4, cb1, cr1);
      ya[2] = y4[0], ya[3] = y4[1], yb[2] = y4[2], yb[3] = y4[3];
    }
    if (q_raw < quads) {
      uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride + 4 * static_cast<size_t>(q);
      __builtin_nontemporal_store(ya[0] | (ya[1] << 8) | (ya[2] << 16) | (ya[3] << 24),
                                  reinterpret_cast<uint32_t *>(y0));
      __builtin_nontemporal_store(yb[0] | (yb[1] << 8) | (yb[2] << 16) | (yb[3] << 24),
                                  reinterpret_cast<uint32_t *>(y0 + p.y_stride));
      // Cb low byte, Cr high (CVPixelBufferUtils.h:358-361)
      __builtin_nontemporal_store(
          cb0 | (cr0 << 8) | (cb1 << 16) | (cr1 << 24),
          reinterpret_cast<uint32_t *>(f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride + 4 * static_cast<size_t>(q)));
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
    uint32_t y4[4], cb, cr;
    encode_block(t, quarter_n, blk, y4, cb, cr);
    uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    uint8_t *y1 = y0 + p.y_stride;
    uint8_t *c = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    y0[2 * bx] = static_cast<uint8_t>(y4[0]);
    y0[2 * bx + 1] = static_cast<uint8_t>(y4[1]);
    y1[2 * bx] = static_cast<uint8_t>(y4[2]);
    y1[2 * bx + 1] = static_cast<uint8_t>(y4[3]);
    c[2 * bx] = static_cast<uint8_t>(cb);
    c[2 * bx + 1] = static_cast<uint8_t>(cr);
  }
}

const char *launch_encode(const EncodeParams &params, int frames, bool fast, hipStream_t stream) {
  EncodeParams p = params;
  if (p.row_pairs_per_block == 0) p.row_pairs_per_block = encode_row_pairs_per_block(p.width, p.height, frames);
  const size_t lds = 3 * 256 * sizeof(EncodeByteEntry) + p.from_linear_bytes;
  if (fast) {
    const uint32_t quads = p.width / 4;
    uint32_t threads = p.block_threads ? p.block_threads : encode_block_threads(p.width);
    if (threads > static_cast<uint32_t>(kMaxBlockThreads)) threads = kMaxBlockThreads;
    const dim3 grid((quads + threads - 1) / threads,
                    (p.height / 2 + p.row_pairs_per_block - 1) / p.row_pairs_per_block, frames);
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
