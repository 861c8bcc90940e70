// CDNA4 (gfx950) kernels of the BT.709 NV12 -> sRGB BGRA decode path.
//
// What the reference does in its first Metal pass -- BT709ToLinearSRGBKernel & friends
// (Renderer/AAPLShaders.metal:336-407: read Y(gid) and CbCr(gid/2), 3x3 matrix, video-gamma
// removal, write into an sRGB8 texture whose store hardware applies the sRGB OETF) -- with the
// arithmetic of the reference's CPU path (bt709_device.h).  The transfer step is a table:
//       byte = bucket[q].base + (x >= bucket[q].edge),  q = floor(x N)
// (transfer_tables.h), x the saturated R, G or B.
//
// Memory plan (HBM-bound: 1.5 B read + 4 B written per pixel, no reuse between workgroups -- nothing to keep in an L2 --
// but the ORDER in which the eight XCDs walk a long launch decides how the DRAM streams interleave: batched launches of 64
// frames or more give each XCD a contiguous band of the frames, see decode_nv12_quads below):
//   * a lane owns 4-wide x 2-high pixel "quads": one dword of each luma row, one dword of CbCr
//     (two Cb,Cr pairs, each shared by a 2x2 block -- chroma is REPLICATED, not interpolated:
//     AAPLShaders.metal:350, BGRAToBT709Converter.m:267-277) and two 16-byte non-temporal
//     stores per quad;
//   * consecutive lanes own consecutive quads of the same row pair, so a wave reads 3 x 256
//     contiguous bytes and writes 2 x 1 KiB contiguous, fully coalesced;
//   * grid = (tiles per row pair, row pairs, frames): one short-lived workgroup per tile,
//     dispatched in address order (x fastest).  Measured: long-lived grid-strided workgroups
//     lose ~20 % of the bandwidth, fewer than 4 resident workgroups per CU lose 1-40 %, and
//     every VALU instruction shows up in run time (bt709_device.h, VALU BUDGET), so the kernels
//     have no loops and no integer divisions;
//   * the tile's global loads are issued before the 4 KiB table is staged into LDS.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "bt709_device.h"

namespace bt709 {
namespace {

// One 4x2 quad: 8 pixels x (R, G, B) = 24 lookups.  Pixel p = 0..3 top row, 4..7 bottom row.
// QUANT: the decoder's mode is sRGB, whose composite is the plain quantiser ("no curve at all", BT709.h:977-983):
// every channel is quantise_byte of its saturated value -- no table, no LDS.  Always set for alpha decoders
// (hasAlphaChannel forces the sRGB mode, MetalBT709Decoder.m:165-169).
template <bool HAS_ALPHA, bool QUANT, bool LOGIDX>
__device__ __forceinline__ void decode_quad(const UnitLookup &u, uint32_t ya, uint32_t yb, uint32_t cw, uint32_t aa,
                                            uint32_t ab, uint32_t alpha_word, u32x4 &top, u32x4 &bot) {
  const Chroma c0 = chroma_terms(byte_of(cw, 0), byte_of(cw, 1));
  const Chroma c1 = chroma_terms(byte_of(cw, 2), byte_of(cw, 3));
  float x[24];
#pragma unroll
  for (int px = 0; px < 8; ++px)
    pixel_rgb(byte_of(px < 4 ? ya : yb, px & 3), (px & 2) ? c1 : c0, x[3 * px], x[3 * px + 1], x[3 * px + 2]);
  static_assert(QUANT || !HAS_ALPHA, "an alpha decoder runs the sRGB mode");
  uint32_t byte[24], al[8];
  if (QUANT) {
#pragma unroll
    for (int i = 0; i < 24; ++i) byte[i] = quantise_byte(x[i]);
  } else {
    uint32_t t[24];
    magic_index12(x, t, u.magic);
    magic_index12(x + 12, t + 12, u.magic);
    if (LOGIDX) {  // log-bucket table (the LINEAR mode): the sum's exponent and top 7 mantissa bits
#pragma unroll
      for (int i = 0; i < 24; ++i) t[i] >>= 16;
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) byte[i] = bucket_byte(u, x[i], t[i]);
  }
#pragma unroll
  for (int px = 0; px < 8; ++px)
    al[px] = HAS_ALPHA ? quantise_byte(alpha_value(byte_of(px < 4 ? aa : ab, px & 3))) << 24 : alpha_word;
  top.x = pack_bgra(byte[0], byte[1], byte[2], al[0]);
  top.y = pack_bgra(byte[3], byte[4], byte[5], al[1]);
  top.z = pack_bgra(byte[6], byte[7], byte[8], al[2]);
  top.w = pack_bgra(byte[9], byte[10], byte[11], al[3]);
  bot.x = pack_bgra(byte[12], byte[13], byte[14], al[4]);
  bot.y = pack_bgra(byte[15], byte[16], byte[17], al[5]);
  bot.z = pack_bgra(byte[18], byte[19], byte[20], al[6]);
  bot.w = pack_bgra(byte[21], byte[22], byte[23], al[7]);
}

// One 2x2 block (general path): 4 pixels x (R, G, B); y = {tl, tr, bl, br}
template <bool HAS_ALPHA, bool QUANT>
__device__ __forceinline__ void decode_block(const UnitLookup &u, const float y[4], float cb, float cr, const float a[4],
                                             uint32_t alpha_word, uint32_t out[4]) {
  const Chroma c = chroma_terms(cb, cr);
  float x[12];
#pragma unroll
  for (int px = 0; px < 4; ++px) pixel_rgb(y[px], c, x[3 * px], x[3 * px + 1], x[3 * px + 2]);
  if (QUANT) {  // sRGB mode: the plain quantiser for every channel (see decode_quad)
#pragma unroll
    for (int px = 0; px < 4; ++px)
      out[px] = pack_bgra(quantise_byte(x[3 * px]), quantise_byte(x[3 * px + 1]), quantise_byte(x[3 * px + 2]),
                          HAS_ALPHA ? quantise_byte(alpha_value(a[px])) << 24 : alpha_word);
    return;
  }
  uint32_t t[12];
  magic_index12(x, t, u.magic);
#pragma unroll
  for (int i = 0; i < 12; ++i) t[i] >>= u.shift;  // general path: the table's form is a run-time value
#pragma unroll
  for (int px = 0; px < 4; ++px)
    out[px] = pack_bgra(bucket_byte(u, x[3 * px], t[3 * px]), bucket_byte(u, x[3 * px + 1], t[3 * px + 1]),
                        bucket_byte(u, x[3 * px + 2], t[3 * px + 2]), alpha_word);
}

}  // namespace

// ---------------------------------------------------------------------------
// Fast path.  Preconditions (checked by the host shim): width % 4 == 0; y, cbcr,
// alpha pointers and strides 4-byte aligned; output pointer and stride 16-byte
// aligned.  grid = (tiles, H/2, frames); a tile is blockDim * kQuadsPerLane quads.
// ---------------------------------------------------------------------------
// RP: consecutive row pairs a workgroup covers, all of them loaded before the table is staged (1 in every shipped kernel; 2 and 4
// served the LINEAR mode's 33 KiB uniform table in the first half of round 5, until its log-bucket table made it 5 KiB:
// decode_nv12_quads_log below).  LOGIDX: the table is in log-bucket form (one shift more per channel).
template <bool HAS_ALPHA, bool NT, bool QUANT, int RP, bool LOGIDX = false>
__device__ __forceinline__ void quads_body(const DecodeParams &p, unsigned char *lds_raw) {
  constexpr int UNROLL = kQuadsPerLane;

  // XCD-AWARE WORK MAP (p.xcd_bands; launches of 64 frames or more, in multiples of 8: launch_decode).  Workgroups are dealt round-robin
  // over the 8 XCDs in dispatch order, so with the plain map (tile, row pair, frame) XCD k owns the tile rows = k mod 8 of
  // ONE address stream -- and an XCD that runs a few percent ahead of another (they sit at different distances from
  // the HBM stacks) widens the band of rows in flight for the whole launch: measured, the longer a launch, the slower
  // (64 / 128 / 256 frames per launch: 0.75 / 0.71 / 0.70 of the roofline against 0.77 for 32).  Here grid.x = 8 x tiles, so
  // x & 7 IS the workgroup's position in the round-robin, and XCD-class b gets a contiguous band of the launch's frames
  // [b F/8, (b + 1) F/8): eight sequential streams that cannot drift into each other.  Long launches then GAIN (no tail,
  // no boundary): 256 frames per launch 0.80-0.81.  Speed only: nothing depends on which XCD a workgroup really lands on.
  const uint32_t tile = p.xcd_bands ? blockIdx.x >> 3 : blockIdx.x;
  const uint32_t frame = p.xcd_bands == 1 ? (blockIdx.x & 7u) * p.frames_per_band + blockIdx.z
                       : (p.xcd_bands == 2 ? blockIdx.z * 8u + (blockIdx.x & 7u) : blockIdx.z);
  const FramePlanes f = frame_planes(p, frame);
  const uint32_t quads = p.width >> 2;
  const uint32_t row_pairs = p.height >> 1;
  // blockDim.y > 1 only for narrow frames: a workgroup then covers blockDim.y consecutive row
  // pairs so that it still has ~8 waves (1920-wide: 256 x 2).  blockDim.x is a whole number of
  // waves, so threadIdx.y is the same in every lane of a wave: taking it from the first lane
  // makes the row pointers scalar (SGPR base + per-lane offset addressing, no 64-bit VALU
  // address arithmetic).
  const uint32_t rp_first = (blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y)) * RP;
  // quad u of this lane: consecutive lanes own consecutive quads (a store instruction must fill whole
  // lines: a lane owning ADJACENT quads measured 3x slower, tools/lab_quads_variants.hip)
  const uint32_t q0 = tile * (blockDim.x * UNROLL) + threadIdx.x;

  // Straight-line code: lanes past the row's end load a clamped (valid) quad and only their
  // stores are predicated.  A divergent `if (q < quads)` around the arithmetic made hipcc put
  // s_waitcnt vmcnt(0) at the join, i.e. each wave waited for the write acknowledgement of its
  // first quad's stores before touching its second quad.
  uint32_t ya[RP][UNROLL], yb[RP][UNROLL], cw[RP][UNROLL], aa[RP][UNROLL], ab[RP][UNROLL];
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const uint32_t rp = min(rp_first + r, row_pairs - 1);
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    const uint8_t *a0 = HAS_ALPHA ? f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride : nullptr;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t q = min((q0 + u * blockDim.x), quads - 1);
      ya[r][u] = load32<NT>(y0 + 4 * q);
      yb[r][u] = load32<NT>(y1 + 4 * q);
      cw[r][u] = load32<NT>(cc + 4 * q);
      if (HAS_ALPHA) {
        aa[r][u] = load32<NT>(a0 + 4 * q);
        ab[r][u] = load32<NT>(a0 + p.alpha_stride + 4 * q);
      }
    }
  }
  if (!QUANT) {  // the sRGB mode needs no table (decode_quad)
    stage_table(lds_raw, p.table_unit, p.table_unit_bytes);  // after the tile's loads are in flight
    __syncthreads();
  }
  // Pin every loaded dword here: hipcc then waits for all of the tile's loads once, before any
  // store is issued, instead of emitting s_waitcnt vmcnt(0) between the first quad's stores and
  // the second quad's arithmetic (which would wait for the stores' write acknowledgements).
#pragma unroll
  for (int r = 0; r < RP; ++r)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      asm volatile("" : "+v"(ya[r][u]), "+v"(yb[r][u]), "+v"(cw[r][u]));
      if (HAS_ALPHA) asm volatile("" : "+v"(aa[r][u]), "+v"(ab[r][u]));
    }

  const UnitLookup ul = unit_lookup(p, lds_raw);
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const uint32_t rp_raw = rp_first + r;
    uint8_t *o0 = f.out + static_cast<size_t>(2 * min(rp_raw, row_pairs - 1)) * p.out_stride;
    uint8_t *o1 = o0 + p.out_stride;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t q = (q0 + u * blockDim.x);
      u32x4 top, bot;
      decode_quad<HAS_ALPHA, QUANT, LOGIDX>(ul, ya[r][u], yb[r][u], cw[r][u], HAS_ALPHA ? aa[r][u] : 0u, HAS_ALPHA ? ab[r][u] : 0u, p.alpha_word, top,
                             bot);
      if (q < quads && rp_raw < row_pairs) {
        store16<NT>(o0 + 16 * q, top);
        store16<NT>(o1 + 16 * q, bot);
      }
    }
  }
}

template <bool HAS_ALPHA, bool NT, bool QUANT>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_quads(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  quads_body<HAS_ALPHA, NT, QUANT, 1>(p, lds_raw);
}

// The plain kernel over a LOG-bucket table (DecodeParams::unit1_shift == 16; transfer_tables.h TransferTable::buckets_log):
// the LINEAR mode's 4 096 uniform buckets become 645 -- 5 KiB staged per workgroup instead of 33 -- for one shift per channel.
// One process, one ring, 4K, LINEAR mode (profiles/r05_ab_linear_log.txt), against the first half of round 5's answer to the
// 33 KiB table (the same body over 2 / 4 row pairs per workgroup, `decode_nv12_quads_rows`, itself 0.706 -> 0.777 over the plain
// kernel): 256 / 64 / 32 / 8 / 1 frames per launch 0.788 / 0.763 / 0.767 / 0.721 / 0.472 against 0.778 / 0.706 / 0.708 /
// 0.648 / 0.451.  The rows kernel is gone.
template <bool NT>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_quads_log(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  quads_body<false, NT, false, 1, true>(p, lds_raw);
}

// ---------------------------------------------------------------------------
// General path: any even width/height, any stride, byte-aligned planes, 4-byte
// aligned output.  One lane per 2x2 block, grid-strided over row pairs.  Correctness
// first; used for ragged or misaligned frames only.
// ---------------------------------------------------------------------------
template <bool HAS_ALPHA, bool QUANT>
__global__ void __launch_bounds__(kBlockThreads)
decode_nv12_blocks(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  if (!QUANT) {
    stage_table(lds_raw, p.table_unit, p.table_unit_bytes);
    __syncthreads();
  }

  const UnitLookup ul = unit_lookup(p, lds_raw);
  const FramePlanes f = frame_planes(p, blockIdx.y);
  const uint32_t bw = p.width >> 1;
  const uint32_t row_pairs = p.height >> 1;

  for (uint32_t rp = blockIdx.x; rp < row_pairs; rp += gridDim.x) {
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    uint32_t *o0 = reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(2 * rp) * p.out_stride);
    uint32_t *o1 = reinterpret_cast<uint32_t *>(f.out + static_cast<size_t>(2 * rp + 1) * p.out_stride);
    for (uint32_t bx = threadIdx.x; bx < bw; bx += kBlockThreads) {
      const float y[4] = {byte_value(y0[2 * bx]), byte_value(y0[2 * bx + 1]), byte_value(y1[2 * bx]),
                          byte_value(y1[2 * bx + 1])};
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      if (HAS_ALPHA) {
        const uint8_t *a0 = f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride;
        const uint8_t *a1 = a0 + p.alpha_stride;
        a[0] = byte_value(a0[2 * bx]);
        a[1] = byte_value(a0[2 * bx + 1]);
        a[2] = byte_value(a1[2 * bx]);
        a[3] = byte_value(a1[2 * bx + 1]);
      }
      uint32_t out[4];
      decode_block<HAS_ALPHA, QUANT>(ul, y, byte_value(cc[2 * bx]), byte_value(cc[2 * bx + 1]), a, p.alpha_word, out);
      o0[2 * bx] = out[0];
      o0[2 * bx + 1] = out[1];
      o1[2 * bx] = out[2];
      o1[2 * bx + 1] = out[3];
    }
  }
}

// ---------------------------------------------------------------------------
// +[BGRAToBT709Converter unconvert:outBGRAPixels:width:height:type:] on its actual input
// (Renderer/BGRAToBT709Converter.h:34-46; .m:146-198 unconvertSoftware): PACKED 4:4:4 words Y | Cb << 8 | Cr << 16, one
// per pixel, every pixel with its own chroma -> BGRA words.  Same per-pixel function as the NV12 kernels (the decoder's
// gamma; the reference hard-selects Apple196 at .m:165-173), alpha byte = the decoder's alpha fill (0 reproduces
// unconvertSoftware's words, .m:187-193).  4 B read + 4 B written per pixel.  VEC: a lane owns 4 consecutive pixels
// (16-byte load and store); otherwise one pixel per lane.  grid = (tiles, rows).
// ---------------------------------------------------------------------------
struct UnconvertParams {
  const uint8_t *in;   // packed words (frame blockIdx.z: in + z * in_step, or ins[z] when the table is used)
  uint8_t *out;        // BGRA words
  int64_t in_step, out_step;  // evenly spaced frames (bt709hip_unconvert_batch)
  const uint8_t *ins[kMaxBatch];
  uint8_t *outs[kMaxBatch];
  uint32_t use_table;
  uint32_t in_stride, out_stride, width, height;
  const void *table_unit;
  uint32_t table_unit_bytes;
  float unit1_magic;  // DecodeParams::unit1_*
  uint32_t unit1_first, unit1_shift;
  uint32_t alpha_word;
};

template <bool VEC, bool QUANT>
__global__ void __launch_bounds__(kBlockThreads)
unconvert_packed444(const UnconvertParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // VEC: a lane owns 4 consecutive pixels of TWO consecutive rows (height is even: BGRAToBT709Converter.m:69-74), both 16-byte
  // loads issued before the table is staged -- the table's 4 KiB are then shared by 2 048 pixels instead of 1 024 and the
  // loads overlap the staging, as in the 1:1 kernel (round 4: 12.7 -> 11.x us per 4K frame, tools/bench_unconvert.py).
  // Frame words are touched once: non-temporal.  !VEC: one pixel per lane, one row per workgroup row, any alignment.
  constexpr uint32_t N = VEC ? 4 : 1, ROWS = VEC ? 2 : 1;
  const uint32_t row0 = blockIdx.y * ROWS;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // group of N pixels
  const uint8_t *frame_in = p.use_table ? p.ins[blockIdx.z] : p.in + static_cast<int64_t>(blockIdx.z) * p.in_step;
  uint8_t *frame_out = p.use_table ? p.outs[blockIdx.z] : p.out + static_cast<int64_t>(blockIdx.z) * p.out_step;
  const bool live = i * N < p.width;
  const uint32_t ic = live ? i : 0u;  // lanes past the row's end load a valid group and do not store (every lane stages the table)
  uint32_t w[ROWS][N];
#pragma unroll
  for (uint32_t r = 0; r < ROWS; ++r) {
    const uint8_t *in = frame_in + static_cast<size_t>(row0 + r) * p.in_stride;
    if (VEC) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + 16 * ic));
      w[r][0] = v.x, w[r][N > 1 ? 1 : 0] = v.y, w[r][N > 2 ? 2 : 0] = v.z, w[r][N > 3 ? 3 : 0] = v.w;
    } else {
      w[r][0] = *reinterpret_cast<const uint32_t *>(in + 4 * ic);
    }
  }
  if (!QUANT) {
    stage_table(lds_raw, p.table_unit, p.table_unit_bytes);  // after the loads are in flight
    __syncthreads();
  }
  const UnitLookup ul = unit_lookup(p, lds_raw);
  if (!live) return;
#pragma unroll
  for (uint32_t r = 0; r < ROWS; ++r) {
    float x[3 * N];
#pragma unroll
    for (uint32_t k = 0; k < N; ++k) {
      const Chroma c = chroma_terms(byte_of(w[r][k], 1), byte_of(w[r][k], 2));
      pixel_rgb(byte_of(w[r][k], 0), c, x[3 * k], x[3 * k + 1], x[3 * k + 2]);
    }
    uint32_t o[N];
#pragma unroll
    for (uint32_t k = 0; k < N; ++k) {
      uint32_t b[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        b[ch] = QUANT ? quantise_byte(x[3 * k + ch]) : bucket_byte(ul, x[3 * k + ch], __float_as_uint(__fadd_rn(x[3 * k + ch], ul.magic)) >> ul.shift);
      o[k] = pack_bgra(b[0], b[1], b[2], p.alpha_word);
    }
    uint8_t *out = frame_out + static_cast<size_t>(row0 + r) * p.out_stride;
    if (VEC) {
      u32x4 v;
      v.x = o[0], v.y = o[N > 1 ? 1 : 0], v.z = o[N > 2 ? 2 : 0], v.w = o[N > 3 ? 3 : 0];
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + 16 * i));
    } else {
      *reinterpret_cast<uint32_t *>(out + 4 * i) = o[0];
    }
  }
}

const char *launch_unconvert(const DecodeParams &t, const UnconvertBatch &b, size_t in_stride, size_t out_stride, uint32_t width, uint32_t height,
                             bool vec, bool quantiser, hipStream_t stream) {
  UnconvertParams p;
  std::memset(&p, 0, sizeof p);
  p.in = static_cast<const uint8_t *>(b.in[0]);
  p.out = static_cast<uint8_t *>(b.out[0]);
  p.in_step = b.in_step;
  p.out_step = b.out_step;
  p.use_table = b.uniform ? 0u : 1u;
  if (!b.uniform)
    for (int i = 0; i < b.count && i < kMaxBatch; ++i) p.ins[i] = static_cast<const uint8_t *>(b.in[i]), p.outs[i] = static_cast<uint8_t *>(b.out[i]);
  p.in_stride = static_cast<uint32_t>(in_stride);
  p.out_stride = static_cast<uint32_t>(out_stride);
  p.width = width;
  p.height = height;
  p.table_unit = t.table_unit;
  p.table_unit_bytes = t.table_unit_bytes;
  p.unit1_magic = t.unit1_magic;
  p.unit1_first = t.unit1_first;
  p.unit1_shift = t.unit1_shift;
  p.alpha_word = t.alpha_word;
  const uint32_t groups = vec ? width / 4 : width;
  const dim3 grid((groups + kBlockThreads - 1) / kBlockThreads, vec ? height / 2 : height, static_cast<uint32_t>(b.count));  // vec: two rows per workgroup row
  const size_t lds = quantiser ? 0 : t.table_unit_bytes;
  if (vec) {
    if (quantiser) hipLaunchKernelGGL((unconvert_packed444<true, true>), grid, dim3(kBlockThreads), lds, stream, p);
    else hipLaunchKernelGGL((unconvert_packed444<true, false>), grid, dim3(kBlockThreads), lds, stream, p);
  } else {
    if (quantiser) hipLaunchKernelGGL((unconvert_packed444<false, true>), grid, dim3(kBlockThreads), lds, stream, p);
    else hipLaunchKernelGGL((unconvert_packed444<false, false>), grid, dim3(kBlockThreads), lds, stream, p);
  }
  return vec ? "unconvert_packed444<vec>" : "unconvert_packed444";
}

// ---------------------------------------------------------------------------
// host-callable launchers (no HIP types in the signature beyond hipStream_t)
// ---------------------------------------------------------------------------
LaunchShape &last_launch_shape() {
  static thread_local LaunchShape shape = {};
  return shape;
}

const char *launch_decode(const DecodeParams &p_in, int frames, int variant, bool has_alpha, bool quantiser, bool nontemporal,
                          int xcd_bands, uint32_t grid_x, uint32_t block_threads, hipStream_t stream) {
  const bool quant = quantiser || has_alpha;  // the sRGB mode: arithmetic, no table
  const size_t lds = quant ? 0 : p_in.table_unit_bytes;
  if (variant == kVariantQuads && xcd_bands && p_in.uniform && frames > kXcdBandMinFrames && frames % 8 != 0) {
    // a long launch of a frame count that is not a multiple of 8: the XCD-aware map over the multiple of 8, the plain map over
    // the (up to 7) frames left, back to back on the stream
    const int head = frames - frames % 8;
    launch_decode(p_in, head, variant, has_alpha, quantiser, nontemporal, xcd_bands, grid_x, block_threads, stream);
    DecodeParams tail = p_in;
    FramePlanes &f = tail.frames[0];
    f.y += static_cast<int64_t>(head) * tail.step_y;
    f.cbcr += static_cast<int64_t>(head) * tail.step_cbcr;
    if (f.alpha) f.alpha += static_cast<int64_t>(head) * tail.step_alpha;
    f.out += static_cast<int64_t>(head) * tail.step_out;
    return launch_decode(tail, frames - head, variant, has_alpha, quantiser, nontemporal, 0, grid_x, block_threads, stream);
  }
  if (variant == kVariantQuads) {
    // grid_x = tiles per row pair; narrow frames stack row pairs in blockDim.y
    const uint32_t by = quads_rows_per_block(block_threads, grid_x);
    dim3 grid(grid_x, (p_in.height / 2 + by - 1) / by, static_cast<uint32_t>(frames));
    const dim3 block(block_threads, by, 1);
    DecodeParams banded = p_in;
    if (xcd_bands && frames >= kXcdBandMinFrames && frames % 8 == 0) {  // see the kernel: 8 contiguous bands of frames, one per XCD class
      banded.xcd_bands = static_cast<uint32_t>(xcd_bands);
      banded.frames_per_band = static_cast<uint32_t>(frames) / 8u;
      grid = dim3(grid_x * 8u, grid.y, banded.frames_per_band);
    }
    const DecodeParams &p = banded;
    LaunchShape &shape = last_launch_shape();
    if (shape.launches++ == 0) {
      shape.grid[0] = grid.x, shape.grid[1] = grid.y, shape.grid[2] = grid.z;
      shape.block[0] = block.x, shape.block[1] = block.y, shape.block[2] = block.z;
      shape.xcd_bands = static_cast<int32_t>(banded.xcd_bands);
    }
    if (has_alpha) {
      hipLaunchKernelGGL((decode_nv12_quads<true, true, true>), grid, block, lds, stream, p);
      return "decode_nv12_quads<alpha>";
    }
    if (quant) {
      if (nontemporal) hipLaunchKernelGGL((decode_nv12_quads<false, true, true>), grid, block, lds, stream, p);
      else hipLaunchKernelGGL((decode_nv12_quads<false, false, true>), grid, block, lds, stream, p);
      return nontemporal ? "decode_nv12_quads<nt,quantiser>" : "decode_nv12_quads<quantiser>";
    }
    if (p_in.unit1_shift != 0) {  // log-bucket table (the LINEAR mode)
      if (nontemporal) hipLaunchKernelGGL((decode_nv12_quads_log<true>), grid, block, lds, stream, p);
      else hipLaunchKernelGGL((decode_nv12_quads_log<false>), grid, block, lds, stream, p);
      return nontemporal ? "decode_nv12_quads_log<nt>" : "decode_nv12_quads_log";
    }
    if (nontemporal) {
      hipLaunchKernelGGL((decode_nv12_quads<false, true, false>), grid, block, lds, stream, p);
      return "decode_nv12_quads<nt>";
    }
    hipLaunchKernelGGL((decode_nv12_quads<false, false, false>), grid, block, lds, stream, p);
    return "decode_nv12_quads";
  }
  // grid_x = workgroups per frame, grid-strided over row pairs
  const DecodeParams &p = p_in;
  const dim3 grid(grid_x, static_cast<uint32_t>(frames), 1);
  const dim3 block(kBlockThreads, 1, 1);
  LaunchShape &shape = last_launch_shape();
  if (shape.launches++ == 0) {
    shape.grid[0] = grid.x, shape.grid[1] = grid.y, shape.grid[2] = grid.z;
    shape.block[0] = block.x, shape.block[1] = block.y, shape.block[2] = block.z;
    shape.xcd_bands = 0;
  }
  if (has_alpha) {
    hipLaunchKernelGGL((decode_nv12_blocks<true, true>), grid, block, lds, stream, p);
    return "decode_nv12_blocks<alpha>";
  }
  if (quant) {
    hipLaunchKernelGGL((decode_nv12_blocks<false, true>), grid, block, lds, stream, p);
    return "decode_nv12_blocks<quantiser>";
  }
  hipLaunchKernelGGL((decode_nv12_blocks<false, false>), grid, block, lds, stream, p);
  return "decode_nv12_blocks";
}

hipError_t prepare_kernels() {
  const int cap = 160 * 1024;  // gfx950: 160 KiB LDS per workgroup
  const void *fns[] = {
      reinterpret_cast<const void *>(&decode_nv12_quads<true, true, true>),
      reinterpret_cast<const void *>(&decode_nv12_quads<false, true, true>),
      reinterpret_cast<const void *>(&decode_nv12_quads<false, false, true>),
      reinterpret_cast<const void *>(&decode_nv12_quads<false, true, false>),
      reinterpret_cast<const void *>(&decode_nv12_quads<false, false, false>),
      reinterpret_cast<const void *>(&decode_nv12_quads_log<true>),
      reinterpret_cast<const void *>(&decode_nv12_quads_log<false>),
      reinterpret_cast<const void *>(&unconvert_packed444<true, false>),
      reinterpret_cast<const void *>(&unconvert_packed444<false, false>),
      reinterpret_cast<const void *>(&decode_nv12_blocks<true, true>),
      reinterpret_cast<const void *>(&decode_nv12_blocks<false, true>),
      reinterpret_cast<const void *>(&decode_nv12_blocks<false, false>),
  };
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace bt709
