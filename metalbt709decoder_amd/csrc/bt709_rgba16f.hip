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
//     start of x's bucket (641 buckets {value, slope}: the floats sharing an exponent and 7 mantissa bits), one
//     exact subtraction and one fma: it lies below the convex curve by at most 1.0e-4 of the value, a fifth of a
//     half's spacing.  (Rounds 2-3: exp2(g * log2(base)) from v_log_f32 / v_exp_f32, two quarter-rate
//     instructions per channel -- 36 to 40 issue cycles per channel against 18 now.)
// T is indexed by the OUTPUT code: one entry per step of H, 24-34 KiB in LDS, plus 5 KiB of tangents, staged once
// per workgroup; a workgroup therefore walks several row pairs (the 8-bit kernel's 4 KiB table allows
// one short-lived workgroup per row pair, this one's does not).
//
// Memory plan (HBM-bound: 1.5 B read + 8 B written per pixel): a lane owns a 2x2 block -- two
// 16-byte stores, one per output row, consecutive lanes writing consecutive 16 bytes (a store
// instruction must fill whole lines) -- and reads its 4 luma bytes and one CbCr pair with 2-byte
// loads where the layout allows.
#include <hip/hip_runtime.h>

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
  // Above the split: the tangent at the start of x's bucket (transfer_tables.h).  Bucket = bits(x) >> 16; its low 10 bits
  // index the table (v_bfe_u32 + v_lshl_add_u32: cand_off already holds "- 8 * (kHalfCandFirst & 0x3ff)"); an x below
  // 2^-5 reads some bytes of the threshold table in front of the tangents instead -- inside the LDS allocation, and unused:
  // such an x is below every split and takes the exact product.  x - x_q is exact (same binade); the fma belongs to the
  // CANDIDATE, not to the reference's arithmetic -- any value in (true * (1 - 4.8e-4), true] gives the same H.
  // Below the split the exact product takes the candidate's place before the one conversion; its half IS H,
  // and it goes through the same settlement: the index is held at h_min - 1 from below, T[h_min] is the
  // smallest x of the WHOLE curve that reaches h_min, so nothing is added (and a value of the low piece that
  // already rounds to h_min is compared with T[h_min + 1] > split).  No clamp of x, no second conversion.
  const uint32_t xb = __float_as_uint(x);
  const u32x2 c = *reinterpret_cast<LdsPairPtr>((((xb >> 16) & 0x3ffu) << 3) + t.cand_off);  // {value, slope}
  const float dx = __fadd_rn(x, -__uint_as_float(xb & 0xffff0000u));
  const float p = __builtin_fmaf(dx, __uint_as_float(c.y), __uint_as_float(c.x));
  const uint32_t h0 = half_bits(x < t.split ? lowv : p);
  typedef __attribute__((address_space(3))) const float *LdsFloatPtr;
  const LdsFloatPtr e = reinterpret_cast<LdsFloatPtr>((max(h0, t.h_below) << 2) + t.table_off);
  return h0 + (x >= e[1] ? 1u : 0u);
}

}  // namespace

// grid = (tiles of blockDim 2x2 blocks, groups of row_pairs_per_block row pairs, frames)
// CURVE: 0 = no curve (LINEAR: the conversion alone), 1 = a power curve above a split point (Apple, sRGB, ITU: the tables decide which)
template <int CURVE, bool HAS_ALPHA, bool PAIRS>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_rgba16f(const DecodeParams p, const HalfParams hp) {
  constexpr bool HAS_TABLE = CURVE != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  if (HAS_TABLE) stage_table(lds_raw, hp.table, hp.table_bytes);  // the device copy starts with the guard entry T[h_min - 1]
  __syncthreads();
  HalfLookup t;
  t.split = hp.split;
  t.low_scale = hp.low_scale;
  t.h_below = hp.h_min - 1u;
  t.table_off = lds_address(lds_raw) + 4u - (hp.h_min << 2);
  t.cand_off = lds_address(lds_raw) + hp.cand_offset - ((kHalfCandFirst & 0x3ffu) << 3);

  // XCD-aware work map (p.xcd_bands, launches of a multiple of 8 frames): see bt709_kernels.hip decode_nv12_quads
  const uint32_t tile = p.xcd_bands ? blockIdx.x >> 3 : blockIdx.x;
  const FramePlanes f = frame_planes(p, p.xcd_bands ? (blockIdx.x & 7u) * p.frames_per_band + blockIdx.z : blockIdx.z);
  const uint32_t blocks = p.width >> 1, row_pairs = p.height >> 1;
  const uint32_t bx = tile * blockDim.x + threadIdx.x;
  if (bx >= blocks) return;
  const uint32_t rp0 = blockIdx.y * hp.row_pairs_per_block, rp1 = min(rp0 + hp.row_pairs_per_block, row_pairs);
  const uint32_t opaque = 0x3c00u << 16;  // A = 1.0

  // The bytes of one row pair of this lane's 2x2 block, as loaded: Y top | Y bottom, CbCr, alpha top | alpha bottom (two bytes
  // each).  The next row pair is fetched before the current one is converted (a workgroup walks row_pairs_per_block pairs):
  // without it every pair waited for its own loads -- hidden while a bench ring fitted the Infinity Cache (16 x 12.4 MB), 40 %
  // of the time once it did not (ring of 64: 17.8 us per 4K frame against 12.7).  Frame bytes are read once: non-temporal.
  struct PairIn {
    uint32_t ya, yb, cc, aa, ab;
  };
  auto fetch = [&](uint32_t rp) {
    PairIn v = {};
    const uint8_t *y0 = f.y + static_cast<size_t>(2 * rp) * p.y_stride + 2 * bx;
    const uint8_t *y1 = y0 + p.y_stride;
    const uint8_t *cc = f.cbcr + static_cast<size_t>(rp) * p.cbcr_stride + 2 * bx;
    if (PAIRS) {
      v.ya = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(y0));
      v.yb = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(y1));
      v.cc = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(cc));
    } else {
      v.ya = static_cast<uint32_t>(y0[0]) | (static_cast<uint32_t>(y0[1]) << 8);
      v.yb = static_cast<uint32_t>(y1[0]) | (static_cast<uint32_t>(y1[1]) << 8);
      v.cc = static_cast<uint32_t>(cc[0]) | (static_cast<uint32_t>(cc[1]) << 8);
    }
    if (HAS_ALPHA) {
      const uint8_t *a0 = f.alpha + static_cast<size_t>(2 * rp) * p.alpha_stride + 2 * bx;
      const uint8_t *a1 = a0 + p.alpha_stride;
      if (PAIRS) {
        v.aa = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(a0));
        v.ab = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(a1));
      } else {
        v.aa = static_cast<uint32_t>(a0[0]) | (static_cast<uint32_t>(a0[1]) << 8);
        v.ab = static_cast<uint32_t>(a1[0]) | (static_cast<uint32_t>(a1[1]) << 8);
      }
    }
    return v;
  };
  PairIn cur = fetch(rp0);
  for (uint32_t rp = rp0; rp < rp1; ++rp) {
    PairIn nxt = cur;
    if (rp + 1 < rp1) nxt = fetch(rp + 1);  // uniform branch
    float yv[4], cb, cr, av[4] = {0.f, 0.f, 0.f, 0.f};
    yv[0] = byte_of(cur.ya, 0), yv[1] = byte_of(cur.ya, 1), yv[2] = byte_of(cur.yb, 0), yv[3] = byte_of(cur.yb, 1);
    cb = byte_of(cur.cc, 0), cr = byte_of(cur.cc, 1);
    if (HAS_ALPHA) av[0] = byte_of(cur.aa, 0), av[1] = byte_of(cur.aa, 1), av[2] = byte_of(cur.ab, 0), av[3] = byte_of(cur.ab, 1);
    cur = nxt;
    const Chroma c = chroma_terms(cb, cr);
    uint32_t w[8];  // per pixel {R | G << 16, B | A << 16}
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float r, g, b;
      pixel_rgb(yv[px], c, r, g, b);
      const uint32_t hr = half_code<HAS_TABLE>(t, r), hg = half_code<HAS_TABLE>(t, g), hb = half_code<HAS_TABLE>(t, b);
      const uint32_t ha = HAS_ALPHA ? (half_bits(alpha_value(av[px])) << 16) : opaque;  // linear alpha, unquantised
      w[2 * px] = hr | (hg << 16);
      w[2 * px + 1] = hb | ha;
    }
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
  uint32_t threads = (blocks + 63) / 64 * 64;
  if (threads > static_cast<uint32_t>(kMaxBlockThreads)) threads = kMaxBlockThreads;
  const uint32_t tiles = (blocks + threads - 1) / threads;
  // row pairs per workgroup: as many (<= 16) as still leave ~6 workgroups per CU; table staging is per workgroup
  // (1 / 2 / 3 / 4 / 6 / 8 per CU, one 4K frame per launch: 28.9 / 21.3 / 18.0 / 18.5 / 17.7 / 17.7 us; 16 per launch: 13.0 / 12.9 / 12.8 / 13.0 / 12.8 / 13.1)
#ifndef BT709_RGBA16F_WG_PER_CU
#define BT709_RGBA16F_WG_PER_CU 6
#endif
#ifndef BT709_RGBA16F_MAX_RPB
#define BT709_RGBA16F_MAX_RPB 16
#endif
  const uint64_t want = static_cast<uint64_t>(BT709_RGBA16F_WG_PER_CU) * (compute_units ? compute_units : 256u);
  uint32_t rpb = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * row_pairs * static_cast<uint32_t>(frames) / want);
  rpb = rpb < 1 ? 1 : (rpb > BT709_RGBA16F_MAX_RPB ? BT709_RGBA16F_MAX_RPB : rpb);
  hp.row_pairs_per_block = rpb;
  hp.wide_store = out_align >= 16 ? 1 : 0;
  dim3 grid(tiles, (row_pairs + rpb - 1) / rpb, static_cast<uint32_t>(frames));
  if (xcd_bands && frames >= kXcdBandMinFrames && frames % 8 == 0) {
    p.xcd_bands = 1;
    p.frames_per_band = static_cast<uint32_t>(frames) / 8u;
    grid = dim3(tiles * 8u, grid.y, p.frames_per_band);
  }
  const dim3 block(threads);
  const bool pairs = in_align >= 2;
  const int curve = hp.table_bytes == 0 ? 0 : 1;
  const size_t lds = curve ? hp.table_bytes : 16;
#define BT709_LAUNCH_RGBA16F(C, A, P) hipLaunchKernelGGL((decode_nv12_rgba16f<C, A, P>), grid, block, lds, stream, p, hp)
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
  const void *fns[] = {
      reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, true, true>),  reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, true, false>),
      reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, false, true>), reinterpret_cast<const void *>(&decode_nv12_rgba16f<0, false, false>),
      reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, true, true>),  reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, true, false>),
      reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, false, true>), reinterpret_cast<const void *>(&decode_nv12_rgba16f<1, false, false>),
  };
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace bt709
