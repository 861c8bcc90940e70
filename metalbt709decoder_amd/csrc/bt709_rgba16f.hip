// CDNA4 (gfx950) kernel of pass 1 into an RGBA16Float render target.
//
// Where sRGB texture writes are unavailable the reference renders BT709ToLinearSRGBKernel & siblings
// into an RGBA16Float intermediate (Renderer/AAPLRenderer.m:143-170): the shader's LINEAR-light
// float4 is stored as four IEEE binary16 values, R,G,B,A, 8 bytes per pixel -- "about 2x slower
// for IO bound shader" (:157).  Per channel that is
//       H(x) = half(curve_to_linear(x)),   x = the saturated non-linear value of bt709_device.h
// with the reference's double-precision pow inside the curve.  No GPU pow is bit-identical to libm's, so:
//   * a CANDIDATE whose half h0 is H or H - 1 is settled exactly by the one threshold above h0:
//     H = h0 + (x >= T[h0 + 1])   (transfer_tables.h HalfTable).  The candidate is ONE fma over the entry {intercept, slope} of
//     x's bucket: above the curve's split point the tangent of the curve at the start of the bucket (it lies below the convex
//     curve by at most 1.0e-4 of the value, a fifth of a half's spacing); below it -- bucket 0 -- {0, low_scale}: the
//     reference's exact product, whose half IS H.  The bucket is the round-toward-zero binary16 of x * index_scale shifted
//     right by 3, held at a floor; index_scale puts the split point ON a bucket boundary, so no comparison with the split is
//     left in the kernel.  Channels go in PAIRS through the packed instructions: v_pk_mul_f32, v_cvt_pkrtz_f16_f32,
//     v_pk_max_u16 for the bucket; v_cvt_pk_f16_f32 and v_pk_sub_u16 (saturating) for h0 and its index into T; the pair (R, G)
//     and the pair (B, A) of a pixel come out of the conversion already packed as the two words of the texel.
//     (Rounds 2-3: exp2(g * log2(base)) from v_log_f32 / v_exp_f32, two quarter-rate instructions per channel -- 36 to 40
//     issue cycles per channel; round 4: x - x_q and an fma, 18; round 5: 12 instructions; this form: 7.)
// T is indexed by the OUTPUT code: one entry per step of H, 24-34 KiB in LDS, plus 4-5 KiB of candidate entries, staged once
// per workgroup -- which is why a workgroup covers 4 blocks x 2 row pairs per lane (below), where the 8-bit kernel with its
// 4 KiB table covers one tile of one row pair.
//
// Memory plan (HBM-bound: 1.5 B read + 8 B written per pixel): a lane owns 2x2 blocks -- two
// 16-byte stores each, one per output row, consecutive lanes writing consecutive 16 bytes (a store
// instruction must fill whole lines; tools/f16_shape_lab.hip: half lines cost 2.4x) -- and reads a block's 4 luma
// bytes and its CbCr pair with 2-byte loads where the layout allows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "bt709_device.h"

namespace bt709 {
namespace {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float *LdsFloatPtr;

struct HalfLookup {
  float index_scale;  // HalfTable::index_scale
  uint32_t below2;    // (h_min - 1) in both halves: the saturating subtraction that indexes T
};
// LDS address of a bucket's entry = its (masked) binary16 bits + this: the ds_read_b64's immediate offset.  ABSOLUTE LDS
// addresses: the dynamic allocation is the kernel's only LDS and starts at byte 0 (no __shared__ variable in this file; a CPU
// test reads .group_segment_fixed_size = 0 from the code object's metadata).
constexpr uint32_t kCandBias = kHalfCandLds - kHalfCandFloor;
static_assert((kHalfCandFloor & 7u) == 0 && (kHalfCandLds & 15u) == 0 && kHalfCandLds - kHalfCandFloor + 0x3ff8u < 0x10000u, "candidate entries: aligned, and the offset fits the instruction");

// The table image of a launch (thresholds, then candidate entries) -> LDS: the thresholds to byte 0, the entries to
// kHalfCandLds.  Every load of a lane is issued before its first write (bt709_device.h stage_table: a round of the loop is an
// L2 round trip inside the workgroup's lifetime), five at a time: at most 40 KiB = 2 560 sixteen-byte words, one round for a
// 512-lane workgroup, two for 256 lanes, more for the 64-lane workgroups of very narrow frames.
__device__ __forceinline__ void stage_half_tables(unsigned char *lds, const void *src, uint32_t cand_offset, uint32_t bytes) {
  const u32x4 *s = reinterpret_cast<const u32x4 *>(src);
  const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
  const uint32_t n = bytes / 16, n_thresholds = cand_offset / 16, gap = (kHalfCandLds - cand_offset) / 16;
  u32x4 *d = reinterpret_cast<u32x4 *>(lds);
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
      if (i < n) d[i < n_thresholds ? i : i + gap] = v[k];
    }
  }
}

// two values -> their binary16 codes in one word (v_cvt_pk_f16_f32, round to nearest even: the same conversion as
// v_cvt_f16_f32, two at a time)
__device__ __forceinline__ uint32_t half_bits2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// The twelve channels of a 2x2 block (x = R, G, B of its four pixels, saturated) and the pixels' alpha values (1.0f without an
// alpha plane) -> the texels' words {R | G << 16, B | A << 16}.  The LDS reads are BATCHED: all twelve entry reads are issued,
// then waited for once; all twelve threshold reads are issued, then waited for once (written one value at a time hipcc puts an
// s_waitcnt behind every read and each wave sits out the LDS latency 24 times per block).
template <bool HAS_TABLE>
__device__ __forceinline__ void half_texels(const HalfLookup &t, const float *x, const float *alpha, uint32_t *w) {
  if (!HAS_TABLE) {  // no curve: the conversion alone
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      w[2 * px] = half_bits2(x[3 * px], x[3 * px + 1]);
      w[2 * px + 1] = half_bits2(x[3 * px + 2], alpha[px]);
    }
    return;
  }
  // buckets, two channels per instruction; the floor holds everything below the split point (and anything the conversion
  // makes of a binary16 subnormal) in bucket 0
  u32x2 c[12];
  const u16x2 floor2 = {static_cast<uint16_t>(kHalfCandFloor), static_cast<uint16_t>(kHalfCandFloor)};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const f32x2 u = f32x2{x[2 * i], x[2 * i + 1]} * t.index_scale;  // binary32 products (v_pk_mul_f32; -ffp-contract=off)
    u16x2 hb = __builtin_bit_cast(u16x2, __builtin_amdgcn_cvt_pkrtz(u.x, u.y));
    hb = __builtin_elementwise_max(hb, floor2);
    const uint32_t pk = __builtin_bit_cast(uint32_t, hb);
    c[2 * i] = *reinterpret_cast<LdsPairPtr>((pk & 0xfff8u) + kCandBias);  // {intercept, slope}
    c[2 * i + 1] = *reinterpret_cast<LdsPairPtr>(((pk >> 16) & 0xfff8u) + kCandBias);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) asm volatile("" : "+v"(c[i]));  // one wait for the batch
  // candidates -> h0 of (R, G) and (B, A) -> index into T.  The fma belongs to the CANDIDATE, not to the reference's
  // arithmetic -- any value in (true * (1 - 4.8e-4), true] gives the same H (the host proves it for every float:
  // tests/native/half_candidate_sweep.cpp) -- except in bucket 0, where it is the reference's product x * low_scale, rounded
  // to binary32 first and THEN to binary16, as on the CPU: the empty asm keeps hipcc from fusing it with the conversion
  // into v_fma_mixlo_f16 (one rounding instead of two).
  float e[12];
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] = __builtin_fmaf(x[3 * px + k], __uint_as_float(c[3 * px + k].y), __uint_as_float(c[3 * px + k].x));
      asm("" : "+v"(p[k]));
    }
    w[2 * px] = half_bits2(p[0], p[1]);
    w[2 * px + 1] = half_bits2(p[2], alpha[px]);
    // T[h0 + 1] at LDS byte 4 * (h0 - (h_min - 1)) + 4; the subtraction saturates at 0: a code below the table (the piece
    // below the split) is compared with T[h_min], the smallest x of the WHOLE curve that reaches h_min, and nothing is added
    const u16x2 below = __builtin_bit_cast(u16x2, t.below2);
    const uint32_t drg = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, w[2 * px]), below));
    const uint32_t db = __builtin_elementwise_sub_sat(static_cast<uint16_t>(w[2 * px + 1]), static_cast<uint16_t>(t.below2));  // B alone: v_sub_u16, upper half zero
    e[3 * px] = *reinterpret_cast<LdsFloatPtr>(((drg & 0xffffu) << 2) + 4u);
    e[3 * px + 1] = *reinterpret_cast<LdsFloatPtr>(((drg >> 16) << 2) + 4u);
    e[3 * px + 2] = *reinterpret_cast<LdsFloatPtr>((db << 2) + 4u);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) asm volatile("" : "+v"(e[i]));  // one wait for the batch
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    uint32_t g_step = x[3 * px + 1] >= e[3 * px + 1] ? 0x10000u : 0u;
    asm("" : "+v"(g_step));  // keeps the two steps of the word apart: one select, one add-with-carry
    w[2 * px] = w[2 * px] + g_step + (x[3 * px] >= e[3 * px] ? 1u : 0u);
    w[2 * px + 1] += x[3 * px + 2] >= e[3 * px + 2] ? 1u : 0u;
  }
}

}  // namespace

// Work shape (round 5).  Rounds 2-4 ran 512-lane workgroups that walked up to 16 row pairs with a one-ahead prefetch (the 24-34 KiB
// table staged once per workgroup): 0.696 of the roofline at 128 frames per launch -- and 0.696 with the arithmetic DELETED
// (profiles/r05_ab_rgba16f_ceiling.txt): the shape was the bound, as the 1:1 kernel's walking forms had been.
// tools/f16_shape_lab.hip timed the traffic pattern alone, shape by shape, over one pair of slabs: the walking shape 0.59-0.61,
// persistent workgroups with a compact front 0.44-0.66, and STRAIGHT-LINE short-lived workgroups -- every load of the tile issued
// first, then the table, then the stores, dispatched in address order like the 1:1 kernel -- 0.81-0.83 without a table and
// 0.74-0.79 with 21-35 KiB staged per workgroup when a workgroup covers enough pixels to pay for it (profiles/r05_f16_shape_lab*.txt).
// So: grid = (tiles, groups of RP row pairs, frames), a lane owns NB 2x2 blocks of each of RP row pairs (block j of a row pair at
// tile * blockDim * NB + j * blockDim + lane: consecutive lanes own consecutive blocks, so every 16-byte store instruction of a
// wave fills 1 KiB of whole lines), no loop.
// Which (NB, RP, lanes per tile)?  Same process, same ring, all with the arithmetic (profiles/r05_ab_rgba16f_shapes*.txt; the
// walking shape of rounds 2-4 = 0.698 / 0.611 / 0.443 at 128 / 16 / 1 frames per launch):
//   (4, 2, 512)  0.745 / 0.691 / 0.434      (4, 3, 512)  0.750 / 0.629 / 0.496      (8, 2, 256)  0.749 / 0.659 / 0.457
//   (2, 3, 960)  0.716 / 0.682 / 0.536      (2, 2, 960)  0.654 / 0.630 / 0.457      (4, 1, 512)  0.628 / 0.583 / 0.487
// Shipped: (4, 2, 512) -- a 4K row pair is one tile of 480 busy lanes, 8 waves per workgroup, four workgroups per CU -- and
// (2, 3, 960) for launches too small to fill the chip with it (a single 4K frame is 540 such workgroups).  With the lookups in
// their packed-pair form (half_texels above; those numbers were taken with 12 instructions per channel) the shipped shape runs
// at 0.76-0.78 and within 1.4 % of the same launch with its arithmetic deleted: what is left to the table-less pattern's 0.84
// is the 39 KiB staged per workgroup (profiles/r05_ab_rgba16f_packed.txt; the shapes re-measured there: (4, 2, 512) still).
#ifndef BT709_RGBA16F_NB
#define BT709_RGBA16F_NB 4
#endif
#ifndef BT709_RGBA16F_RP
#define BT709_RGBA16F_RP 2
#endif
#ifndef BT709_RGBA16F_TILE_LANES
#define BT709_RGBA16F_TILE_LANES 512
#endif
struct F16Shape {
  int nb, rp, lanes;
};
constexpr F16Shape kF16Large = {BT709_RGBA16F_NB, BT709_RGBA16F_RP, BT709_RGBA16F_TILE_LANES};
// The XCD-aware work map pays from 8 frames per launch on here (the 1:1 kernel: from 64, bt709_kernels.h): same process, same
// ring, 4K 32 / 16 / 8 frames per launch 0.722 / 0.689 / 0.660 against 0.685 / 0.670 / 0.650 with the plain map, 1080p 32
// frames 0.643 against 0.612 (profiles/r05_ab_rgba16f_packed.txt).
#ifndef BT709_RGBA16F_BAND_MIN_FRAMES
#define BT709_RGBA16F_BAND_MIN_FRAMES 8
#endif
constexpr int kF16BandMinFrames = BT709_RGBA16F_BAND_MIN_FRAMES;
// Rows of at most 1 024 blocks (1080p: 240 busy lanes of a 256-lane workgroup): a third row pair per staged table -- for LONG
// launches only (1080p, 512 frames per launch: 0.759 against 0.726; 128 frames: 0.696 against 0.721; 32: 0.635 against 0.689).
#ifndef BT709_RGBA16F_NARROW_RP
#define BT709_RGBA16F_NARROW_RP 3
#endif
constexpr F16Shape kF16Narrow = {4, BT709_RGBA16F_NARROW_RP, 512};
constexpr uint64_t kF16NarrowMinGroupsPerSlot = 48;  // ... of the 4 x CUs workgroup slots: 1080p from 273 frames per launch on
constexpr F16Shape kF16Small = {2, 3, 960};   // launches of fewer than 4 workgroups per CU in the large shape

// CURVE: 0 = no curve (LINEAR: the conversion alone), 1 = a power curve above a split point (Apple, sRGB, ITU: the tables decide which)
template <int CURVE, bool HAS_ALPHA, bool PAIRS, int NB, int RP>
__global__ void __launch_bounds__(1024)
decode_nv12_rgba16f(const DecodeParams p, const HalfParams hp) {
  constexpr bool HAS_TABLE = CURVE != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  // XCD-aware work map (p.xcd_bands, launches of a multiple of 8 frames): see bt709_kernels.hip decode_nv12_quads
  const uint32_t tile = p.xcd_bands ? blockIdx.x >> 3 : blockIdx.x;
  const FramePlanes f = frame_planes(p, p.xcd_bands ? (blockIdx.x & 7u) * p.frames_per_band + blockIdx.z : blockIdx.z);
  const uint32_t blocks = p.width >> 1, row_pairs = p.height >> 1;
  // blockDim.y SLICES of a workgroup share its table, each with row pairs of its own (narrow frames: a 1080p row pair keeps 240
  // lanes busy -- two slices make the 512-lane workgroup a 4K row pair gets).  blockDim.x is a multiple of 64: a wave lies in one
  // slice, its row pointers stay scalar.
  const uint32_t slice = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const uint32_t rp_base = (blockIdx.y * blockDim.y + slice) * RP;
  const uint32_t bx0 = tile * (blockDim.x * NB) + threadIdx.x;

  // Every load of the tile first: the bytes of a 2x2 block as loaded -- Y top | Y bottom, CbCr, alpha top | alpha bottom (two
  // bytes each).  Lanes past the row's end and row pairs past the frame's load a clamped (valid) block; only their stores are
  // predicated (a divergent region around the arithmetic would put a wait for the stores at its join, bt709_kernels.hip).
  // Frame bytes are read once: non-temporal.
  uint32_t ya[RP][NB], yb[RP][NB], cw[RP][NB], aa[RP][NB], ab[RP][NB];
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const uint32_t rp = min(rp_base + r, row_pairs - 1);
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride;
    const uint8_t *a0 = HAS_ALPHA ? f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride : nullptr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const uint32_t bx = min(bx0 + j * blockDim.x, blocks - 1);
      if (PAIRS) {
        ya[r][j] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(y0 + 2 * bx));
        yb[r][j] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(y1 + 2 * bx));
        cw[r][j] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(cc + 2 * bx));
      } else {
        ya[r][j] = static_cast<uint32_t>(y0[2 * bx]) | (static_cast<uint32_t>(y0[2 * bx + 1]) << 8);
        yb[r][j] = static_cast<uint32_t>(y1[2 * bx]) | (static_cast<uint32_t>(y1[2 * bx + 1]) << 8);
        cw[r][j] = static_cast<uint32_t>(cc[2 * bx]) | (static_cast<uint32_t>(cc[2 * bx + 1]) << 8);
      }
      if (HAS_ALPHA) {
        const uint8_t *a1 = a0 + p.alpha_stride;
        if (PAIRS) {
          aa[r][j] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(a0 + 2 * bx));
          ab[r][j] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(a1 + 2 * bx));
        } else {
          aa[r][j] = static_cast<uint32_t>(a0[2 * bx]) | (static_cast<uint32_t>(a0[2 * bx + 1]) << 8);
          ab[r][j] = static_cast<uint32_t>(a1[2 * bx]) | (static_cast<uint32_t>(a1[2 * bx + 1]) << 8);
        }
      }
    }
  }
  if (HAS_TABLE) {
    // after the tile's loads are in flight; the device copy starts with the guard entry T[h_min - 1], the candidate entries
    // follow the thresholds in memory and go to their fixed place in LDS (bt709_kernels.h kHalfCandLds)
    stage_half_tables(lds_raw, hp.table, hp.cand_offset, hp.table_bytes);
    __syncthreads();
  }
  // pin every loaded word here: one wait for all of the tile's loads, before the first store (bt709_kernels.hip)
#pragma unroll
  for (int r = 0; r < RP; ++r)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      asm volatile("" : "+v"(ya[r][j]), "+v"(yb[r][j]), "+v"(cw[r][j]));
      if (HAS_ALPHA) asm volatile("" : "+v"(aa[r][j]), "+v"(ab[r][j]));
    }
  HalfLookup t;
  t.index_scale = hp.index_scale;
  t.below2 = (hp.h_min - 1u) * 0x10001u;

#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const uint32_t rp = rp_base + r;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const uint32_t bx = bx0 + j * blockDim.x;
      float yv[4], av[4] = {0.f, 0.f, 0.f, 0.f}, alpha[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      yv[0] = byte_of(ya[r][j], 0), yv[1] = byte_of(ya[r][j], 1), yv[2] = byte_of(yb[r][j], 0), yv[3] = byte_of(yb[r][j], 1);
      const float cb = byte_of(cw[r][j], 0), cr = byte_of(cw[r][j], 1);
      if (HAS_ALPHA) av[0] = byte_of(aa[r][j], 0), av[1] = byte_of(aa[r][j], 1), av[2] = byte_of(ab[r][j], 0), av[3] = byte_of(ab[r][j], 1);
      const Chroma c = chroma_terms(cb, cr);
      float x[12];  // R, G, B of the block's four pixels
#pragma unroll
      for (int px = 0; px < 4; ++px) pixel_rgb(yv[px], c, x[3 * px], x[3 * px + 1], x[3 * px + 2]);
      if (HAS_ALPHA) {
#pragma unroll
        for (int px = 0; px < 4; ++px) alpha[px] = alpha_value(av[px]);  // linear alpha, unquantised
      }
      uint32_t w[8];  // per pixel {R | G << 16, B | A << 16}
      half_texels<HAS_TABLE>(t, x, alpha, w);
      if (bx < blocks && rp < row_pairs) {
        uint8_t *o0 = f.out + static_cast<size_t>(2 * rp) * p.out_stride + 16 * bx;
        uint8_t *o1 = o0 + p.out_stride;
        if (hp.wide_store) {
          store16<true>(o0, u32x4{w[0], w[1], w[2], w[3]});
          store16<true>(o1, u32x4{w[4], w[5], w[6], w[7]});
        } else {  // 8-byte aligned target
          store8<true>(o0, u32x2{w[0], w[1]});
          store8<true>(o0 + 8, u32x2{w[2], w[3]});
          store8<true>(o1, u32x2{w[4], w[5]});
          store8<true>(o1 + 8, u32x2{w[6], w[7]});
        }
      }
    }
  }
}

const char *launch_decode_rgba16f(const DecodeParams &p_in, const HalfParams &hp_in, int frames, bool has_alpha,
                                  uint32_t in_align, uint32_t out_align, uint32_t compute_units, bool xcd_bands, hipStream_t stream) {
  if (xcd_bands && p_in.uniform && frames > kXcdBandMinFrames && frames % 8 != 0) {
    // any count of 64 frames or more: the XCD-aware map over the multiple of 8, the plain map over the rest (launch_decode); shorter
    // launches take the map only when their count IS a multiple of 8 (a second launch would cost more than the map returns)
    const int head = frames - frames % 8;
    launch_decode_rgba16f(p_in, hp_in, head, has_alpha, in_align, out_align, compute_units, xcd_bands, stream);
    DecodeParams tail = p_in;
    FramePlanes &f = tail.frames[0];
    f.y += static_cast<int64_t>(head) * tail.step_y;
    f.cbcr += static_cast<int64_t>(head) * tail.step_cbcr;
    if (f.alpha) f.alpha += static_cast<int64_t>(head) * tail.step_alpha;
    f.out += static_cast<int64_t>(head) * tail.step_out;
    return launch_decode_rgba16f(tail, hp_in, frames - head, has_alpha, in_align, out_align, compute_units, false, stream);
  }
  HalfParams hp = hp_in;
  DecodeParams p = p_in;
  const uint32_t blocks = p.width / 2, row_pairs = p.height / 2;
  // equal tiles of at most lanes * NB blocks, lanes rounded up to whole waves: 3840 -> 1 tile x 512 lanes x 4 blocks (480 busy),
  // 1920 -> 1 x 256, 7680 -> 2 x 512
  auto tiles_of = [&](const F16Shape &sh) { return (blocks + static_cast<uint32_t>(sh.lanes * sh.nb) - 1) / static_cast<uint32_t>(sh.lanes * sh.nb); };
  auto groups_of = [&](const F16Shape &sh) { return static_cast<uint64_t>(tiles_of(sh)) * ((row_pairs + sh.rp - 1) / sh.rp) * static_cast<uint32_t>(frames); };
  const uint64_t fill = 4ull * (compute_units ? compute_units : 256u);
  // 0 = large, 1 = narrow, 2 = small
  const int which = groups_of(kF16Large) < fill ? 2 : (blocks <= 256u * static_cast<uint32_t>(kF16Narrow.nb) && groups_of(kF16Narrow) >= kF16NarrowMinGroupsPerSlot * fill ? 1 : 0);
  const F16Shape sh = which == 2 ? kF16Small : (which == 1 ? kF16Narrow : kF16Large);
  const uint32_t tiles = tiles_of(sh);
  uint32_t threads = ((blocks + tiles - 1) / tiles + static_cast<uint32_t>(sh.nb) - 1) / static_cast<uint32_t>(sh.nb);
  threads = (threads + 63) / 64 * 64;
  hp.row_pairs_per_block = static_cast<uint32_t>(sh.rp);
  hp.wide_store = out_align >= 16 ? 1 : 0;
  dim3 grid(tiles, (row_pairs + static_cast<uint32_t>(sh.rp) - 1) / static_cast<uint32_t>(sh.rp), static_cast<uint32_t>(frames));
  // slices (see the kernel), for rows that fill fewer than 256 lanes: as many as make a 512-lane workgroup, fewer while the launch
  // would otherwise have too few workgroups.  Same process, same ring (profiles/r05_ab_rgba16f_packed.txt 7): 720p 1 024 / 64
  // frames per launch 0.646 / 0.601 against 0.562 / 0.545 without, 640 x 360 0.58-0.59 against 0.38-0.41; NOT for 1080p's 256
  // lanes (two slices: 0.70-0.71 against 0.72, 32 frames per launch 0.646 against 0.676).
#ifndef BT709_RGBA16F_MAX_SLICES
#define BT709_RGBA16F_MAX_SLICES 8
#endif
#ifndef BT709_RGBA16F_SLICE_MIN_FILL
#define BT709_RGBA16F_SLICE_MIN_FILL 2
#endif
  uint32_t slices = 1;
  if (which != 2) {
    slices = threads >= 256u ? 1u : std::min<uint32_t>(512u / threads, BT709_RGBA16F_MAX_SLICES);
    while (slices > 1 && static_cast<uint64_t>(tiles) * ((grid.y + slices - 1) / slices) * static_cast<uint32_t>(frames) < BT709_RGBA16F_SLICE_MIN_FILL * fill) --slices;
    grid.y = (grid.y + slices - 1) / slices;
  }
  if (xcd_bands && frames >= kF16BandMinFrames && frames % 8 == 0) {
    p.xcd_bands = 1;
    p.frames_per_band = static_cast<uint32_t>(frames) / 8u;
    grid = dim3(tiles * 8u, grid.y, p.frames_per_band);
  }
  const dim3 block(threads, slices);
  LaunchShape &shape = last_launch_shape();
  if (shape.launches++ == 0) {
    shape.grid[0] = grid.x, shape.grid[1] = grid.y, shape.grid[2] = grid.z;
    shape.block[0] = block.x, shape.block[1] = block.y, shape.block[2] = block.z;
    shape.xcd_bands = static_cast<int32_t>(p.xcd_bands);
  }
  const bool pairs = in_align >= 2;
  const int curve = hp.table_bytes == 0 ? 0 : 1;
  // thresholds from byte 0, candidate entries from kHalfCandLds (bt709_kernels.h); the host refuses a table that does not fit
#ifndef BT709_RGBA16F_LDS_FLOOR
#define BT709_RGBA16F_LDS_FLOOR 0  // lab: a larger allocation = fewer workgroups per CU (41 KiB: three instead of four)
#endif
  const size_t lds = curve ? std::max<size_t>(kHalfCandLds + (hp.table_bytes - hp.cand_offset), BT709_RGBA16F_LDS_FLOOR) : 16;
#define BT709_LAUNCH_RGBA16F(C, A, P)                                                                                               \
  do {                                                                                                                              \
    if (which == 2) hipLaunchKernelGGL((decode_nv12_rgba16f<C, A, P, kF16Small.nb, kF16Small.rp>), grid, block, lds, stream, p, hp);        \
    else if (which == 1) hipLaunchKernelGGL((decode_nv12_rgba16f<C, A, P, kF16Narrow.nb, kF16Narrow.rp>), grid, block, lds, stream, p, hp); \
    else hipLaunchKernelGGL((decode_nv12_rgba16f<C, A, P, kF16Large.nb, kF16Large.rp>), grid, block, lds, stream, p, hp);                  \
  } while (0)
#define BT709_LAUNCH_RGBA16F_AP(C)                                                            \
  do {                                                                                        \
    if (has_alpha) { if (pairs) BT709_LAUNCH_RGBA16F(C, true, true); else BT709_LAUNCH_RGBA16F(C, true, false); } \
    else { if (pairs) BT709_LAUNCH_RGBA16F(C, false, true); else BT709_LAUNCH_RGBA16F(C, false, false); }         \
  } while (0)
  if (curve == 0) BT709_LAUNCH_RGBA16F_AP(0);
  else BT709_LAUNCH_RGBA16F_AP(1);
#undef BT709_LAUNCH_RGBA16F_AP
#undef BT709_LAUNCH_RGBA16F
  return has_alpha ? "decode_nv12_rgba16f<alpha>" : "decode_nv12_rgba16f";
}

hipError_t prepare_rgba16f_kernels() {
  const int cap = 160 * 1024;
#define BT709_F16_FNS(NB, RP)                                                                                                              \
  reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, true, true, NB, RP>), reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, true, false, NB, RP>),   \
  reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, false, true, NB, RP>), reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, false, false, NB, RP>), \
  reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, true, true, NB, RP>), reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, true, false, NB, RP>),   \
  reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, false, true, NB, RP>), reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, false, false, NB, RP>)
  const void *fns[] = {BT709_F16_FNS(kF16Large.nb, kF16Large.rp), BT709_F16_FNS(kF16Narrow.nb, kF16Narrow.rp), BT709_F16_FNS(kF16Small.nb, kF16Small.rp)};
#undef BT709_F16_FNS
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace bt709
