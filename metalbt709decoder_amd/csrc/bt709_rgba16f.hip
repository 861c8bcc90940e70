// CDNA4 (gfx950) kernel of pass 1 into an RGBA16Float render target.
//
// Where sRGB texture writes are unavailable the reference renders BT709ToLinearSRGBKernel & siblings
// into an RGBA16Float intermediate (Renderer/AAPLRenderer.m:143-170): the shader's LINEAR-light
// float4 is stored as four IEEE binary16 values, R,G,B,A, 8 bytes per pixel -- "about 2x slower
// for IO bound shader" (:157).  Per channel that is
//       H(x) = half(curve_to_linear(x)),   x = the saturated non-linear value of bt709_device.h
// with the reference's double-precision pow inside the curve.  No GPU pow is bit-identical to libm's, so:
//   * below the curve's split point the reference multiplies by a constant: one exact float multiply
//     and the hardware conversion (v_cvt_f16_f32, round to nearest even) give H directly;
//   * above it, a CANDIDATE whose half h0 is H or H - 1 is settled exactly by the one threshold above h0:
//     H = h0 + (x >= T[h0 + 1])   (transfer_tables.h HalfTable).  The candidate is the tangent of the curve at the
//     start of x's bucket (641 buckets {intercept, slope}: the floats sharing an exponent and 7 mantissa bits), ONE
//     fma: it lies below the convex curve by at most 1.0e-4 of the value, a fifth of a half's spacing.  (Rounds
//     2-3: exp2(g * log2(base)) from v_log_f32 / v_exp_f32, two quarter-rate instructions per channel -- 36 to 40
//     issue cycles per channel; round 4: x - x_q and an fma, 18; now 14.)
// T is indexed by the OUTPUT code: one entry per step of H, 24-34 KiB in LDS, plus 5 KiB of tangents, staged once
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

struct HalfLookup {
  float split, low_scale;
  uint32_t h_below;    // h_min - 1
  uint32_t table_off;  // LDS address of T[h_min] - 4 * h_min
  uint32_t cand_off;   // LDS address of the candidate tangents - 8 * kHalfCandFirst's low 10 bits (see half_code)
};

__device__ __forceinline__ uint32_t half_bits(float v) {
  const _Float16 h = static_cast<_Float16>(v);  // v_cvt_f16_f32, round to nearest even
  return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h));
}

// H(x) for a saturated x in [0, 1]
template <bool HAS_TABLE>
__device__ __forceinline__ uint32_t half_code(const HalfLookup &t, float x) {
  // The product is rounded to binary32 first, THEN to binary16, as on the CPU: the empty asm keeps hipcc
  // from fusing multiply and conversion into v_fma_mixlo_f16 (one rounding instead of two).
  float lowv = __fmul_rn(x, t.low_scale);  // exact below the split for 1/16; x itself when there is no curve
  asm("" : "+v"(lowv));
  if (!HAS_TABLE) return half_bits(lowv);
  // Above the split: the tangent of the curve at the start of x's bucket (transfer_tables.h), in slope / intercept form:
  // p = x * slope + intercept, ONE fma (round 4 formed x - x_q first: two more instructions per channel).  Bucket = bits(x) >> 16;
  // its low 10 bits index the table (v_bfe_u32 + v_lshl_add_u32: cand_off already holds "- 8 * (kHalfCandFirst & 0x3ff)"); an x
  // below 2^-5 reads some bytes in front of or behind the 641 real entries instead -- inside the LDS allocation, which covers all
  // 1 024 values of the index, and unused: such an x is below every split and takes the exact product.  The fma belongs to the
  // CANDIDATE, not to the reference's arithmetic -- any value in (true * (1 - 4.8e-4), true] gives the same H (the host proves it
  // for every float: tests/native/half_candidate_sweep.cpp).
  // Below the split the exact product takes the candidate's place before the one conversion; its half IS H,
  // and it goes through the same settlement: the index is held at h_min - 1 from below, T[h_min] is the
  // smallest x of the WHOLE curve that reaches h_min, so nothing is added (and a value of the low piece that
  // already rounds to h_min is compared with T[h_min + 1] > split).  No clamp of x, no second conversion.
  const uint32_t xb = __float_as_uint(x);
  const u32x2 c = *reinterpret_cast<LdsPairPtr>((((xb >> 16) & 0x3ffu) << 3) + t.cand_off);  // {intercept, slope}
  const float p = __builtin_fmaf(x, __uint_as_float(c.y), __uint_as_float(c.x));
  const uint32_t h0 = half_bits(x < t.split ? lowv : p);
  typedef __attribute__((address_space(3))) const float *LdsFloatPtr;
  const LdsFloatPtr e = reinterpret_cast<LdsFloatPtr>((max(h0, t.h_below) << 2) + t.table_off);
  return h0 + (x >= e[1] ? 1u : 0u);
}

// The same settlement for N values at once, the LDS reads BATCHED: all N tangent reads are issued, then waited for once; all N
// threshold reads are issued, then waited for once.  Written one value at a time (half_code above) hipcc puts an s_waitcnt
// behind every read -- 108 waits for the 96 reads of a lane's four blocks -- and each wave sits out the LDS latency 96 times.
// Same float operations per value in the same order: the bytes cannot differ (the exhaustive sweeps run on this form).
template <bool HAS_TABLE, int N>
__device__ __forceinline__ void half_codes(const HalfLookup &t, const float *x, uint32_t *h) {
  if (!HAS_TABLE) {
#pragma unroll
    for (int i = 0; i < N; ++i) h[i] = half_code<false>(t, x[i]);
    return;
  }
  u32x2 c[N];
#pragma unroll
  for (int i = 0; i < N; ++i) c[i] = *reinterpret_cast<LdsPairPtr>((((__float_as_uint(x[i]) >> 16) & 0x3ffu) << 3) + t.cand_off);  // {intercept, slope}
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(c[i]));  // one wait for the batch
  typedef __attribute__((address_space(3))) const float *LdsFloatPtr;
  uint32_t h0[N];
  float e[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float lowv = __fmul_rn(x[i], t.low_scale);
    asm("" : "+v"(lowv));  // the product is rounded to binary32 first, then to binary16 (see half_code)
    const float p = __builtin_fmaf(x[i], __uint_as_float(c[i].y), __uint_as_float(c[i].x));
    h0[i] = half_bits(x[i] < t.split ? lowv : p);
    e[i] = reinterpret_cast<LdsFloatPtr>((max(h0[i], t.h_below) << 2) + t.table_off)[1];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(e[i]));  // one wait for the batch
#pragma unroll
  for (int i = 0; i < N; ++i) h[i] = h0[i] + (x[i] >= e[i] ? 1u : 0u);
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
// (2, 3, 960) for launches too small to fill the chip with it (a single 4K frame is 540 such workgroups).  The same shapes
// with the arithmetic deleted stream at 0.84: what is left between 0.75 and that is the lookups' own cost (225 VALU
// instructions and 24 LDS gathers per 2x2 block), not the traffic pattern.
#ifndef BT709_RGBA16F_NB
#define BT709_RGBA16F_NB 4
#endif
#ifndef BT709_RGBA16F_RP
#define BT709_RGBA16F_RP 2
#endif
#ifndef BT709_RGBA16F_TILE_LANES
#define BT709_RGBA16F_TILE_LANES 512
#endif
#ifndef BT709_RGBA16F_BATCH
#define BT709_RGBA16F_BATCH 12  // values settled per LDS batch: 12 = a whole 2x2 block, 6 = two pixels
#endif
constexpr int kF16Batch = BT709_RGBA16F_BATCH;
struct F16Shape {
  int nb, rp, lanes;
};
constexpr F16Shape kF16Large = {BT709_RGBA16F_NB, BT709_RGBA16F_RP, BT709_RGBA16F_TILE_LANES};
constexpr F16Shape kF16Small = {2, 3, 960};  // launches of fewer than 4 workgroups per CU in the large shape

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
  const uint32_t rp_base = blockIdx.y * RP;
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
    stage_table(lds_raw, hp.table, hp.table_bytes);  // after the tile's loads are in flight; the device copy starts with the guard entry T[h_min - 1]
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
  t.split = hp.split;
  t.low_scale = hp.low_scale;
  t.h_below = hp.h_min - 1u;
  t.table_off = lds_address(lds_raw) + 4u - (hp.h_min << 2);
  t.cand_off = lds_address(lds_raw) + hp.cand_offset - ((kHalfCandFirst & 0x3ffu) << 3);
  const uint32_t opaque = 0x3c00u << 16;  // A = 1.0

#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const uint32_t rp = rp_base + r;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const uint32_t bx = bx0 + j * blockDim.x;
      float yv[4], av[4] = {0.f, 0.f, 0.f, 0.f};
      yv[0] = byte_of(ya[r][j], 0), yv[1] = byte_of(ya[r][j], 1), yv[2] = byte_of(yb[r][j], 0), yv[3] = byte_of(yb[r][j], 1);
      const float cb = byte_of(cw[r][j], 0), cr = byte_of(cw[r][j], 1);
      if (HAS_ALPHA) av[0] = byte_of(aa[r][j], 0), av[1] = byte_of(aa[r][j], 1), av[2] = byte_of(ab[r][j], 0), av[3] = byte_of(ab[r][j], 1);
      const Chroma c = chroma_terms(cb, cr);
      float x[12];  // R, G, B of the block's four pixels
#pragma unroll
      for (int px = 0; px < 4; ++px) pixel_rgb(yv[px], c, x[3 * px], x[3 * px + 1], x[3 * px + 2]);
      uint32_t hc[12];
      if constexpr (kF16Batch >= 12) {
        half_codes<HAS_TABLE, 12>(t, x, hc);
      } else {
        half_codes<HAS_TABLE, kF16Batch>(t, x, hc);
        half_codes<HAS_TABLE, 12 - kF16Batch>(t, x + kF16Batch, hc + kF16Batch);
      }
      uint32_t w[8];  // per pixel {R | G << 16, B | A << 16}
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const uint32_t ha = HAS_ALPHA ? (half_bits(alpha_value(av[px])) << 16) : opaque;  // linear alpha, unquantised
        w[2 * px] = hc[3 * px] | (hc[3 * px + 1] << 16);
        w[2 * px + 1] = hc[3 * px + 2] | ha;
      }
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
    // any count of 64 frames or more: the XCD-aware map over the multiple of 8, the plain map over the rest (launch_decode)
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
  const uint64_t large_groups = static_cast<uint64_t>(tiles_of(kF16Large)) * ((row_pairs + kF16Large.rp - 1) / kF16Large.rp) * static_cast<uint32_t>(frames);
  const bool small = large_groups < 4ull * (compute_units ? compute_units : 256u);
  const F16Shape sh = small ? kF16Small : kF16Large;
  const uint32_t tiles = tiles_of(sh);
  uint32_t threads = ((blocks + tiles - 1) / tiles + static_cast<uint32_t>(sh.nb) - 1) / static_cast<uint32_t>(sh.nb);
  threads = (threads + 63) / 64 * 64;
  hp.row_pairs_per_block = static_cast<uint32_t>(sh.rp);
  hp.wide_store = out_align >= 16 ? 1 : 0;
  dim3 grid(tiles, (row_pairs + static_cast<uint32_t>(sh.rp) - 1) / static_cast<uint32_t>(sh.rp), static_cast<uint32_t>(frames));
  if (xcd_bands && frames >= kXcdBandMinFrames && frames % 8 == 0) {
    p.xcd_bands = 1;
    p.frames_per_band = static_cast<uint32_t>(frames) / 8u;
    grid = dim3(tiles * 8u, grid.y, p.frames_per_band);
  }
  const dim3 block(threads);
  const bool pairs = in_align >= 2;
  const int curve = hp.table_bytes == 0 ? 0 : 1;
  // the candidate index is 10 bits of x's float pattern (half_code): the allocation covers every value of it, the part behind the
  // 641 real entries is never staged and never used (an x that reads it is below every split point)
  const size_t lds = curve ? std::max<size_t>(hp.table_bytes, hp.cand_offset + 8u * (0x400u - (kHalfCandFirst & 0x3ffu))) : 16;
#define BT709_LAUNCH_RGBA16F(C, A, P)                                                                                               \
  do {                                                                                                                              \
    if (small) hipLaunchKernelGGL((decode_nv12_rgba16f<C, A, P, kF16Small.nb, kF16Small.rp>), grid, block, lds, stream, p, hp);      \
    else hipLaunchKernelGGL((decode_nv12_rgba16f<C, A, P, kF16Large.nb, kF16Large.rp>), grid, block, lds, stream, p, hp);           \
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
  const void *fns[] = {BT709_F16_FNS(kF16Large.nb, kF16Large.rp), BT709_F16_FNS(kF16Small.nb, kF16Small.rp)};
#undef BT709_F16_FNS
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace bt709
