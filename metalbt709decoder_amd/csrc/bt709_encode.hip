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
// with the matrix of BT709.h:222-244: Ey = (Kr*R + Kg*G) + Kb*B, Eb = (B-Ey)/1.8556f,
// Er = (R-Ey)/1.5748f, Y = round(Ey*219 + 16), C = round(E*224 + 128).  The two divisions
// are IEEE correctly rounded (__fdiv_rn); round() is C's half-away-from-zero on a positive
// float.  No multiply-add is contracted (-ffp-contract=off); the division's own internal
// FMAs are part of a correctly rounded quotient.
//
// A lane owns a 4x2-pixel quad (two blocks): two 16-byte loads, three 4-byte stores; lanes
// of a wave are consecutive quads of one row pair; grid = (tiles, row pairs, 1).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bt709_constants.h"
#include "bt709_kernels.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t round_pos(float v) {
  // (int)round((double)v) for v >= 0: ties away from zero
  const float t = __builtin_truncf(v);
  return static_cast<uint32_t>(t) + ((v - t) >= 0.5f ? 1u : 0u);
}

struct Ycc {
  float ey, eb, er;
};

__device__ __forceinline__ Ycc rgbn_to_e(float rn, float gn, float bn) {
  Ycc o;
  o.ey = __fadd_rn(__fadd_rn(__fmul_rn(kKr, rn), __fmul_rn(kKg, gn)), __fmul_rn(kKb, bn));  // BT709.h:222
  o.eb = __fdiv_rn(__fadd_rn(bn, -o.ey), kCbSpan);                                            // BT709.h:223
  o.er = __fdiv_rn(__fadd_rn(rn, -o.ey), kCrSpan);                                            // BT709.h:224
  return o;
}

__device__ __forceinline__ uint32_t quant_y(float ey) {  // BT709.h:233, 244
  return round_pos(__fadd_rn(__fmul_rn(ey, static_cast<float>(kYMax - kYMin)), 16.0f));
}
__device__ __forceinline__ uint32_t quant_c(float e) {  // BT709.h:234-235, 245-246
  return round_pos(__fadd_rn(__fmul_rn(e, static_cast<float>(kCMax - kCMin)), 128.0f));
}

__device__ __forceinline__ uint32_t from_linear(const TransferBucket *__restrict__ tbl, float n, float v) {
  const float xs = __fmul_rn(v, n);  // exact: n is a power of two
  const uint32_t q = static_cast<uint32_t>(xs);
  const TransferBucket e = tbl[q];
  return e.base + (xs >= e.edge ? 1u : 0u);
}

// one 2x2 block: p = {top-left, top-right, bottom-left, bottom-right} BGRA words
__device__ __forceinline__ void encode_block(const EncodeByteEntry *__restrict__ bytes,
                                             const TransferBucket *__restrict__ fl, float fl_n, const uint32_t p[4],
                                             uint32_t y[4], uint32_t &cb, uint32_t &cr) {
  float sum[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const EncodeByteEntry r = bytes[(p[i] >> 16) & 0xff];
    const EncodeByteEntry g = bytes[(p[i] >> 8) & 0xff];
    const EncodeByteEntry b = bytes[p[i] & 0xff];
    sum[0] = i ? __fadd_rn(sum[0], r.lin) : r.lin;
    sum[1] = i ? __fadd_rn(sum[1], g.lin) : g.lin;
    sum[2] = i ? __fadd_rn(sum[2], b.lin) : b.lin;
    y[i] = quant_y(rgbn_to_e(r.enc_norm, g.enc_norm, b.enc_norm).ey);
  }
  float an[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    an[c] = __fmul_rn(static_cast<float>(from_linear(fl, fl_n, __fmul_rn(sum[c], 0.25f))), kInv255);  // /4.0f, byteNorm
  const Ycc a = rgbn_to_e(an[0], an[1], an[2]);
  cb = quant_c(a.eb);
  cr = quant_c(a.er);
}

}  // namespace

__global__ void __launch_bounds__(kBlockThreads)
encode_bgra_nv12(const EncodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  EncodeByteEntry *bytes = reinterpret_cast<EncodeByteEntry *>(lds_raw);
  TransferBucket *fl = reinterpret_cast<TransferBucket *>(lds_raw + 256 * sizeof(EncodeByteEntry));

  const uint32_t quads = p.width >> 2;
  const uint32_t rp = blockIdx.y;
  const uint32_t q_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = min(q_raw, quads - 1);
  const uint8_t *s0 = p.bgra + static_cast<size_t>(2 * rp) * p.bgra_stride + 16 * static_cast<size_t>(q);
  const u32x4 top = *reinterpret_cast<const u32x4 *>(s0);
  const u32x4 bot = *reinterpret_cast<const u32x4 *>(s0 + p.bgra_stride);

  {  // stage both tables after the loads are in flight
    u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
    const u32x4 *sb = reinterpret_cast<const u32x4 *>(p.per_byte);
    const u32x4 *sf = reinterpret_cast<const u32x4 *>(p.from_linear);
    const uint32_t nb = 256 * sizeof(EncodeByteEntry) / 16, nf = p.from_linear_bytes / 16;
    for (uint32_t i = threadIdx.x; i < nb + nf; i += blockDim.x) d[i] = i < nb ? sb[i] : sf[i - nb];
  }
  __syncthreads();

  uint32_t ya[4], yb[4], cb0, cr0, cb1, cr1;
  {
    const uint32_t blk[4] = {top.x, top.y, bot.x, bot.y};
    uint32_t y4[4];
    encode_block(bytes, fl, p.from_linear_scale, blk, y4, cb0, cr0);
    ya[0] = y4[0], ya[1] = y4[1], yb[0] = y4[2], yb[1] = y4[3];
  }
  {
    const uint32_t blk[4] = {top.z, top.w, bot.z, bot.w};
    uint32_t y4[4];
    encode_block(bytes, fl, p.from_linear_scale, blk, y4, cb1, cr1);
    ya[2] = y4[0], ya[3] = y4[1], yb[2] = y4[2], yb[3] = y4[3];
  }
  if (q_raw < quads) {
    uint8_t *y0 = p.y + static_cast<size_t>(2 * rp) * p.y_stride + 4 * static_cast<size_t>(q);
    *reinterpret_cast<uint32_t *>(y0) = ya[0] | (ya[1] << 8) | (ya[2] << 16) | (ya[3] << 24);
    *reinterpret_cast<uint32_t *>(y0 + p.y_stride) = yb[0] | (yb[1] << 8) | (yb[2] << 16) | (yb[3] << 24);
    *reinterpret_cast<uint32_t *>(p.cbcr + static_cast<size_t>(rp) * p.cbcr_stride + 4 * static_cast<size_t>(q)) =
        cb0 | (cr0 << 8) | (cb1 << 16) | (cr1 << 24);  // Cb low byte, Cr high (CVPixelBufferUtils.h:358-361)
  }
}

// General layout: one lane per 2x2 block, scalar loads/stores, any alignment.
__global__ void __launch_bounds__(kBlockThreads)
encode_bgra_nv12_blocks(const EncodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  EncodeByteEntry *bytes = reinterpret_cast<EncodeByteEntry *>(lds_raw);
  TransferBucket *fl = reinterpret_cast<TransferBucket *>(lds_raw + 256 * sizeof(EncodeByteEntry));
  {
    u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
    const u32x4 *sb = reinterpret_cast<const u32x4 *>(p.per_byte);
    const u32x4 *sf = reinterpret_cast<const u32x4 *>(p.from_linear);
    const uint32_t nb = 256 * sizeof(EncodeByteEntry) / 16, nf = p.from_linear_bytes / 16;
    for (uint32_t i = threadIdx.x; i < nb + nf; i += blockDim.x) d[i] = i < nb ? sb[i] : sf[i - nb];
  }
  __syncthreads();
  const uint32_t bw = p.width >> 1;
  const uint32_t rp = blockIdx.y;
  for (uint32_t bx = blockIdx.x * blockDim.x + threadIdx.x; bx < bw; bx += gridDim.x * blockDim.x) {
    const uint32_t *r0 = reinterpret_cast<const uint32_t *>(p.bgra + static_cast<size_t>(2 * rp) * p.bgra_stride);
    const uint32_t *r1 = reinterpret_cast<const uint32_t *>(p.bgra + static_cast<size_t>(2 * rp + 1) * p.bgra_stride);
    const uint32_t blk[4] = {r0[2 * bx], r0[2 * bx + 1], r1[2 * bx], r1[2 * bx + 1]};
    uint32_t y4[4], cb, cr;
    encode_block(bytes, fl, p.from_linear_scale, blk, y4, cb, cr);
    uint8_t *y0 = p.y + static_cast<size_t>(2 * rp) * p.y_stride;
    uint8_t *y1 = y0 + p.y_stride;
    uint8_t *c = p.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    y0[2 * bx] = static_cast<uint8_t>(y4[0]);
    y0[2 * bx + 1] = static_cast<uint8_t>(y4[1]);
    y1[2 * bx] = static_cast<uint8_t>(y4[2]);
    y1[2 * bx + 1] = static_cast<uint8_t>(y4[3]);
    c[2 * bx] = static_cast<uint8_t>(cb);
    c[2 * bx + 1] = static_cast<uint8_t>(cr);
  }
}

const char *launch_encode(const EncodeParams &p, bool fast, hipStream_t stream) {
  const size_t lds = 256 * sizeof(EncodeByteEntry) + p.from_linear_bytes;
  if (fast) {
    const uint32_t quads = p.width / 4;
    const dim3 grid((quads + kBlockThreads - 1) / kBlockThreads, p.height / 2, 1);
    hipLaunchKernelGGL(encode_bgra_nv12, grid, dim3(kBlockThreads), lds, stream, p);
    return "encode_bgra_nv12";
  }
  const dim3 grid((p.width / 2 + kBlockThreads - 1) / kBlockThreads, p.height / 2, 1);
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
