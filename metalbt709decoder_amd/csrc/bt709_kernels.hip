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
// The transfer step is an exact bucketed threshold table (transfer_tables.h)
// staged in LDS: q = (uint)(x*N); byte = base[q] + (x >= edge[q]).
//
// Memory plan (HBM-bound: 1.5 B read + 4 B written per pixel, no reuse between
// workgroups, so no XCD-aware remap is needed):
//   * a lane owns a 4-wide x 2-high pixel quad-pair: one dword of each luma row,
//     one dword of CbCr (two Cb,Cr pairs, each shared by a 2x2 block -- chroma is
//     REPLICATED, not interpolated: AAPLShaders.metal:350, BGRAToBT709Converter.m:
//     267-277) and two 16-byte stores;
//   * consecutive lanes own consecutive quads of the same row pair, so a wave reads
//     3 x 256 contiguous bytes and writes 2 x 1 KiB contiguous, fully coalesced;
//   * one workgroup per row pair (grid.x = H/2), blockDim sized so a lane owns two
//     quads; all loads of the row pair are issued before any arithmetic;
//   * grid.y = frame: a batch of independent frames is one launch.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bt709_constants.h"
#include "bt709_kernels.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float byte_of(uint32_t w, int i) {
  return static_cast<float>((w >> (8 * i)) & 0xffu);  // -> v_cvt_f32_ubyte{i}
}

// (v - off) * (1/255f): integer-valued floats subtract exactly, so this equals the
// reference's int subtract followed by int->float conversion.
__device__ __forceinline__ float centre_norm(float v, float off) {
  return __fmul_rn(__fadd_rn(v, -off), kInv255);
}

__device__ __forceinline__ float sat(float v) {
  // saturatef (Renderer/sRGB.h:18-27); v is never NaN here
  return __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f);
}

__device__ __forceinline__ uint32_t lookup(const TransferBucket *__restrict__ tbl, float n, float x) {
  const uint32_t q = static_cast<uint32_t>(__fmul_rn(x, n));  // exact: n is a power of two
  const TransferBucket e = tbl[q];
  return e.base + (x >= e.edge ? 1u : 0u);
}

struct Chroma {  // the four Cb/Cr products of one 2x2 block
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

// saturated non-linear R,G,B of one pixel
__device__ __forceinline__ void pixel_rgbn(float ybyte, const Chroma &c, float &r, float &g, float &b) {
  const float yv = __fmul_rn(centre_norm(ybyte, 16.0f), kMY);
  r = sat(__fadd_rn(yv, c.cr_r));
  g = sat(__fadd_rn(__fadd_rn(yv, c.cb_g), c.cr_g));
  b = sat(__fadd_rn(yv, c.cb_b));
}

__device__ __forceinline__ uint32_t decode_px(const TransferBucket *__restrict__ tbl, float n, float ybyte,
                                              const Chroma &c, uint32_t alpha_word) {
  float r, g, b;
  pixel_rgbn(ybyte, c, r, g, b);
  const uint32_t R = lookup(tbl, n, r);
  const uint32_t G = lookup(tbl, n, g);
  const uint32_t B = lookup(tbl, n, b);
  return alpha_word | (R << 16) | (G << 8) | B;
}

// linear alpha sample -> byte: R channel of the matrix with Cb=Cr=128, then plain
// 8-bit quantisation (AAPLShaders.metal:249-271; CPU twin BT709.h:466-513).  `ident`
// is the identity-gamma table, which is exactly round(x*255).
__device__ __forceinline__ uint32_t decode_alpha(const TransferBucket *__restrict__ ident, float n, float abyte) {
  const float yv = __fmul_rn(centre_norm(abyte, 16.0f), kMY);
  return lookup(ident, n, sat(yv)) << 24;
}

__device__ __forceinline__ void stage_table(void *lds, const void *src, uint32_t bytes) {
  u32x4 *d = reinterpret_cast<u32x4 *>(lds);
  const u32x4 *s = reinterpret_cast<const u32x4 *>(src);
  for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

template <bool NT>
__device__ __forceinline__ void store16(uint8_t *p, u32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
  else *reinterpret_cast<u32x4 *>(p) = v;
}

}  // namespace

// ---------------------------------------------------------------------------
// Fast path.  Preconditions (checked by the host shim): width % 4 == 0; y, cbcr,
// alpha pointers and strides 4-byte aligned; output pointer and stride 16-byte
// aligned.
//
// Launch shape (measured, tools/kernel_lab.hip, DESIGN.md "launch geometry"): ONE
// workgroup per row pair and per frame -- grid = (H/2, frames) -- with
// blockDim = ceil(W/8) rounded up to a wave (480 threads for 3840), so that every
// lane owns UNROLL = 2 quads.  Workgroups are dispatched in address order and live
// for one burst of loads and one burst of stores; long-lived grid-stride workgroups
// scatter the DRAM access stream and lose ~20 % of the bandwidth.  The loops below
// still stride by gridDim/blockDim, so any launch shape is correct.
//
// The frame loads of the first tile are issued BEFORE the transfer table is staged
// into LDS, so the table's L2 round trip hides under the HBM latency of the tile.
// ---------------------------------------------------------------------------
template <bool HAS_ALPHA, bool NT>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_quads(const DecodeParams p) {
  constexpr int UNROLL = kQuadsPerLane;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucket *tbl = reinterpret_cast<TransferBucket *>(lds_raw);

  const FramePlanes f = p.frames[blockIdx.y];
  const float n = p.table_scale;
  const uint32_t quads = p.width >> 2;
  const uint32_t row_pairs = p.height >> 1;
  const uint32_t threads = blockDim.x;
  bool table_ready = false;

  for (uint32_t rp = blockIdx.x; rp < row_pairs; rp += gridDim.x) {
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    const uint8_t *a0 = HAS_ALPHA ? f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride : nullptr;
    uint8_t *o0 = f.out + static_cast<size_t>(2 * rp) * p.out_stride;
    uint8_t *o1 = o0 + p.out_stride;

    for (uint32_t q0 = 0; q0 < quads; q0 += threads * UNROLL) {
      uint32_t ya[UNROLL], yb[UNROLL], cw[UNROLL], aa[UNROLL], ab[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t q = q0 + u * threads + threadIdx.x;
        if (q < quads) {
          ya[u] = *reinterpret_cast<const uint32_t *>(y0 + 4 * q);
          yb[u] = *reinterpret_cast<const uint32_t *>(y1 + 4 * q);
          cw[u] = *reinterpret_cast<const uint32_t *>(cc + 4 * q);
          if (HAS_ALPHA) {
            aa[u] = *reinterpret_cast<const uint32_t *>(a0 + 4 * q);
            ab[u] = *reinterpret_cast<const uint32_t *>(a0 + p.alpha_stride + 4 * q);
          }
        }
      }
      if (!table_ready) {  // uniform across the workgroup
        stage_table(tbl, p.table, p.table_bytes);
        __syncthreads();
        table_ready = true;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t q = q0 + u * threads + threadIdx.x;
        if (q < quads) {
          const Chroma c0 = chroma_terms(byte_of(cw[u], 0), byte_of(cw[u], 1));
          const Chroma c1 = chroma_terms(byte_of(cw[u], 2), byte_of(cw[u], 3));
          u32x4 top, bot;
          uint32_t al[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            al[i] = HAS_ALPHA ? decode_alpha(tbl, n, byte_of(aa[u], i)) : p.alpha_word;
            al[4 + i] = HAS_ALPHA ? decode_alpha(tbl, n, byte_of(ab[u], i)) : p.alpha_word;
          }
          top.x = decode_px(tbl, n, byte_of(ya[u], 0), c0, al[0]);
          top.y = decode_px(tbl, n, byte_of(ya[u], 1), c0, al[1]);
          top.z = decode_px(tbl, n, byte_of(ya[u], 2), c1, al[2]);
          top.w = decode_px(tbl, n, byte_of(ya[u], 3), c1, al[3]);
          bot.x = decode_px(tbl, n, byte_of(yb[u], 0), c0, al[4]);
          bot.y = decode_px(tbl, n, byte_of(yb[u], 1), c0, al[5]);
          bot.z = decode_px(tbl, n, byte_of(yb[u], 2), c1, al[6]);
          bot.w = decode_px(tbl, n, byte_of(yb[u], 3), c1, al[7]);
          store16<NT>(o0 + 16 * q, top);
          store16<NT>(o1 + 16 * q, bot);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// General path: any even width/height, any stride, byte-aligned planes, 4-byte
// aligned output.  One lane per 2x2 block.  Correctness first; used for ragged or
// misaligned frames only.
// ---------------------------------------------------------------------------
template <bool HAS_ALPHA>
__global__ void __launch_bounds__(kBlockThreads)
decode_nv12_blocks(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucket *tbl = reinterpret_cast<TransferBucket *>(lds_raw);
  stage_table(tbl, p.table, p.table_bytes);
  __syncthreads();

  const FramePlanes f = p.frames[blockIdx.y];
  const float n = p.table_scale;
  const uint32_t bw = p.width >> 1;
  const uint32_t row_pairs = p.height >> 1;

  for (uint32_t rp = blockIdx.x; rp < row_pairs; rp += gridDim.x) {
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    uint32_t *o0 = reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(2 * rp) * p.out_stride);
    uint32_t *o1 = reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(2 * rp + 1) * p.out_stride);
    for (uint32_t bx = threadIdx.x; bx < bw; bx += kBlockThreads) {
      const Chroma c = chroma_terms(static_cast<float>(cc[2 * bx]), static_cast<float>(cc[2 * bx + 1]));
      uint32_t al[4] = {p.alpha_word, p.alpha_word, p.alpha_word, p.alpha_word};
      if (HAS_ALPHA) {
        const uint8_t *a0 = f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride;
        const uint8_t *a1 = a0 + p.alpha_stride;
        al[0] = decode_alpha(tbl, n, static_cast<float>(a0[2 * bx]));
        al[1] = decode_alpha(tbl, n, static_cast<float>(a0[2 * bx + 1]));
        al[2] = decode_alpha(tbl, n, static_cast<float>(a1[2 * bx]));
        al[3] = decode_alpha(tbl, n, static_cast<float>(a1[2 * bx + 1]));
      }
      o0[2 * bx] = decode_px(tbl, n, static_cast<float>(y0[2 * bx]), c, al[0]);
      o0[2 * bx + 1] = decode_px(tbl, n, static_cast<float>(y0[2 * bx + 1]), c, al[1]);
      o1[2 * bx] = decode_px(tbl, n, static_cast<float>(y1[2 * bx]), c, al[2]);
      o1[2 * bx + 1] = decode_px(tbl, n, static_cast<float>(y1[2 * bx + 1]), c, al[3]);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused decode + exact 2:1 downscale (pass 1 + pass 2 of the reference).  A 2x2
// luma block shares one CbCr sample and becomes one output pixel.  Two-pass
// equivalent arithmetic: each decoded byte is linearised as the sRGB8 sampler
// would (table returns the linear float directly), the four are averaged
// (((a+b)+c)+d)*0.25f, then sRGB-encoded and quantised through the LINEAR-mode
// table (second LDS table).
// A lane owns 4 output pixels (8 luma columns x 2 rows): two dwordx2 luma loads,
// one dwordx2 chroma load, one 16-byte store.  Preconditions: width % 8 == 0,
// planes/strides 8-byte aligned, output 16-byte aligned; otherwise the shim uses
// decode_half_blocks.
// ---------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float lookup_linear(const TransferBucketLinear *__restrict__ tbl, float n, float x) {
  const uint32_t q = static_cast<uint32_t>(__fmul_rn(x, n));
  const TransferBucketLinear e = tbl[q];
  return x >= e.edge ? e.lin_above : e.lin_below;
}

__device__ __forceinline__ uint32_t half_px(const TransferBucketLinear *__restrict__ dec, float dn,
                                            const TransferBucket *__restrict__ enc, float en,
                                            float y00, float y01, float y10, float y11, const Chroma &c,
                                            uint32_t alpha_word) {
  float r[4], g[4], b[4];
  pixel_rgbn(y00, c, r[0], g[0], b[0]);
  pixel_rgbn(y01, c, r[1], g[1], b[1]);
  pixel_rgbn(y10, c, r[2], g[2], b[2]);
  pixel_rgbn(y11, c, r[3], g[3], b[3]);
  float lr[4], lg[4], lb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lr[i] = lookup_linear(dec, dn, r[i]);
    lg[i] = lookup_linear(dec, dn, g[i]);
    lb[i] = lookup_linear(dec, dn, b[i]);
  }
  const float mr = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lr[0], lr[1]), lr[2]), lr[3]), 0.25f);
  const float mg = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lg[0], lg[1]), lg[2]), lg[3]), 0.25f);
  const float mb = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(lb[0], lb[1]), lb[2]), lb[3]), 0.25f);
  const uint32_t R = lookup(enc, en, mr);
  const uint32_t G = lookup(enc, en, mg);
  const uint32_t B = lookup(enc, en, mb);
  return alpha_word | (R << 16) | (G << 8) | B;
}

}  // namespace

template <bool NT, bool WIDE>
__global__ void __launch_bounds__(kBlockThreads)
decode_nv12_half(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TransferBucketLinear *dec = reinterpret_cast<TransferBucketLinear *>(lds_raw);
  TransferBucket *enc = reinterpret_cast<TransferBucket *>(lds_raw + p.table_bytes);
  stage_table(dec, p.table, p.table_bytes);
  stage_table(enc, p.table2, p.table2_bytes);
  __syncthreads();

  const FramePlanes f = p.frames[blockIdx.y];
  const float dn = p.table_scale, en = p.table2_scale;
  const uint32_t out_w = p.width >> 1;
  const uint32_t out_rows = p.height >> 1;

  for (uint32_t orow = blockIdx.x; orow < out_rows; orow += gridDim.x) {
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * orow) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(orow) * p.cbcr_stride;
    uint8_t *o = f.out + static_cast<size_t>(orow) * p.out_stride;
    if (WIDE) {
      const uint32_t groups = out_w >> 2;  // 4 output pixels per lane
      for (uint32_t gq = threadIdx.x; gq < groups; gq += kBlockThreads) {
        const uint2 ya = *reinterpret_cast<const uint2 *>(y0 + 8 * gq);
        const uint2 yb = *reinterpret_cast<const uint2 *>(y1 + 8 * gq);
        const uint2 cw = *reinterpret_cast<const uint2 *>(cc + 8 * gq);
        const Chroma c0 = chroma_terms(byte_of(cw.x, 0), byte_of(cw.x, 1));
        const Chroma c1 = chroma_terms(byte_of(cw.x, 2), byte_of(cw.x, 3));
        const Chroma c2 = chroma_terms(byte_of(cw.y, 0), byte_of(cw.y, 1));
        const Chroma c3 = chroma_terms(byte_of(cw.y, 2), byte_of(cw.y, 3));
        u32x4 v;
        v.x = half_px(dec, dn, enc, en, byte_of(ya.x, 0), byte_of(ya.x, 1), byte_of(yb.x, 0), byte_of(yb.x, 1), c0, p.alpha_word);
        v.y = half_px(dec, dn, enc, en, byte_of(ya.x, 2), byte_of(ya.x, 3), byte_of(yb.x, 2), byte_of(yb.x, 3), c1, p.alpha_word);
        v.z = half_px(dec, dn, enc, en, byte_of(ya.y, 0), byte_of(ya.y, 1), byte_of(yb.y, 0), byte_of(yb.y, 1), c2, p.alpha_word);
        v.w = half_px(dec, dn, enc, en, byte_of(ya.y, 2), byte_of(ya.y, 3), byte_of(yb.y, 2), byte_of(yb.y, 3), c3, p.alpha_word);
        store16<NT>(o + 16 * gq, v);
      }
    } else {
      for (uint32_t ox = threadIdx.x; ox < out_w; ox += kBlockThreads) {
        const Chroma c = chroma_terms(static_cast<float>(cc[2 * ox]), static_cast<float>(cc[2 * ox + 1]));
        reinterpret_cast<uint32_t *>(o)[ox] =
            half_px(dec, dn, enc, en, static_cast<float>(y0[2 * ox]), static_cast<float>(y0[2 * ox + 1]),
                    static_cast<float>(y1[2 * ox]), static_cast<float>(y1[2 * ox + 1]), c, p.alpha_word);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host-callable launchers (no HIP types in the signature beyond hipStream_t)
// ---------------------------------------------------------------------------
const char *launch_decode(const DecodeParams &p, int frames, int variant, bool has_alpha, bool nontemporal,
                          uint32_t grid_x, uint32_t block_threads, hipStream_t stream) {
  const dim3 grid(grid_x, static_cast<uint32_t>(frames), 1);
  const size_t lds = p.table_bytes;
  if (variant == kVariantQuads) {
    const dim3 block(block_threads, 1, 1);
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
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_nv12_half<true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_nv12_half<false, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_nv12_half<false, false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}

const char *launch_decode_half(const DecodeParams &p, int frames, bool wide, bool nontemporal, uint32_t grid_x,
                               hipStream_t stream) {
  const dim3 grid(grid_x, static_cast<uint32_t>(frames), 1);
  const dim3 block(kBlockThreads, 1, 1);
  const size_t lds = static_cast<size_t>(p.table_bytes) + p.table2_bytes;
  if (wide) {
    if (nontemporal) hipLaunchKernelGGL((decode_nv12_half<true, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((decode_nv12_half<false, true>), grid, block, lds, stream, p);
    return "decode_nv12_half<wide>";
  }
  hipLaunchKernelGGL((decode_nv12_half<false, false>), grid, block, lds, stream, p);
  return "decode_nv12_half<narrow>";
}

}  // namespace bt709
