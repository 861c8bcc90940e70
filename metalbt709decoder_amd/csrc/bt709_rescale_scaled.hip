// CDNA4 (gfx950) kernels of the FUSED decode + rescale to ANY output size, and of pass 2 alone
// (-[MetalScaleRenderContext renderScaled:...] + samplingShader: Renderer/MetalScaleRenderContext.m:55-105,
// Renderer/AAPLShaders.metal:73-85).  Arithmetic and tables: bt709_rescale.h.
//
//   decode_nv12_scaled     any output size, bilinear taps, one lane per output column walking strips of rows
//   render_scaled          pass 2 alone from an 8-bit or RGBA16Float intermediate
#include <atomic>

#include "bt709_rescale.h"

namespace bt709 {

// ---------------------------------------------------------------------------
// Fused decode + bilinear rescale to ANY output size (MetalScaleRenderContext -renderScaled:
// for a view that is not an exact 2:1 of the frame).  Definition (ours: the reference leaves it
// to the sampler hardware):
//   sx = (ox + 0.5f) * (W / OW) - 0.5f,  x0 = floor(sx), fx = sx - x0, taps clamped to the edge
//   (same in y); each tap is decoded to its 8-bit sRGB value and linearised as the sRGB8 sampler
//   does; v = (((w00*l00 + w01*l01) + w10*l10) + w11*l11) with w00 = (1-fx)(1-fy), ...;
//   sRGB-encode, quantise.  For an exact 2:1 ratio every weight is 0.25 and this is bit for bit
//   the decode_nv12_half result.
// One lane per output column, walking a strip of `rows` consecutive output rows (grid = (ceil(OW /
// blockDim), strips, frames)): the horizontal tap positions and weights are computed once per lane, the
// vertical ones once per strip (lane i does row i; rows read them with v_readlane_b32), the 14 KiB of
// tables are staged once per workgroup -- of the launch, in the persistent form (see the kernel).  What round 2
// changed (4K -> 1440p, 8 frames per launch: 204 -> 255 Gpixel/s out; DESIGN 6.5 has every shape and every step):
//   * the ROW CACHE: which source rows an output row needs is the same for every lane, and consecutive
//     output rows share source rows whenever the vertical ratio is below 2 (always when enlarging), so
//     the two linearised rows of the previous output row stay in registers and only rows not seen yet are
//     decoded (scalar branches); the chroma products are kept the same way (two luma rows share a CbCr
//     row).  12 lookups per output pixel become 6 * scale_y;
//   * fetches are unconditional, one output row ahead, in two explicit register sets (see the loop);
//   * planes are raw buffer resources (scalar row offset, 32-bit lane offset: no VALU address arithmetic).
// Tap fetch, as wide as the layout allows:
//   TAPS_WIDE  (planes and strides 4-byte aligned, width % 4 == 0, width >= 8): per source row ONE
//              aligned 8-byte load per plane that contains both horizontal taps, and one v_perm_b32 with
//              a per-lane selector (computed once) picks them out;
//   TAPS_PAIRS (CbCr plane 2-byte aligned): a tap's Cb,Cr with one 2-byte load;
//   TAPS_BYTES any layout: byte loads.
// 4-byte coalesced stores.
// ---------------------------------------------------------------------------
enum : int { TAPS_BYTES = 0, TAPS_PAIRS = 1, TAPS_WIDE = 2, TAPS_SHARED = 3, TAPS_ONCE = 4 };

// output rows whose source rows are fetched ahead of the row being produced (scaled_strip, "HOW FAR AHEAD"): per-lane tap
// fetches (8 VGPRs per row in flight) and the two by-wave forms (4 per row)
#ifndef BT709_SCALED_AHEAD
#define BT709_SCALED_AHEAD 1
#endif
#ifndef BT709_SCALED_AHEAD_WAVE
#define BT709_SCALED_AHEAD_WAVE 3
#endif
constexpr int kScaledAhead = BT709_SCALED_AHEAD, kScaledAheadWave = BT709_SCALED_AHEAD_WAVE;
#ifndef BT709_SCALED_STORE_AUX
#define BT709_SCALED_STORE_AUX 2  // cache-policy bits of the output store (buffer instruction aux operand): 2 = slc, a streaming store (the
                                  // output is written once and not read again: 1080p -> 4K +6 %, one frame per launch +13 %, profiles/r06_ab_scaled_ahead.txt)
#endif
constexpr int kScaledStoreAux = BT709_SCALED_STORE_AUX;
// pass 2 alone (render_scaled): rows fetched ahead from a BGRA8 (4 VGPRs per row in flight) / RGBA16Float (8) intermediate
#ifndef BT709_RENDER_AHEAD8
#define BT709_RENDER_AHEAD8 1
#endif
#ifndef BT709_RENDER_AHEAD16
#define BT709_RENDER_AHEAD16 1
#endif
constexpr int kRenderAhead8 = BT709_RENDER_AHEAD8, kRenderAhead16 = BT709_RENDER_AHEAD16;

// Vertical taps of a strip of at most 64 output rows starting at oy0: lane i holds row oy0 + i (sy = (oy + 0.5f) *
// scale_y - 0.5f, y0 = floor(sy), fy = sy - y0).  gfx950 has no scalar float unit, so one evaluation costs 8 VALU
// instructions per row whichever way it is written; done once per strip by the lanes in parallel, a row takes its
// two numbers with v_readlane_b32 -- which also puts them in SGPRs, so row offsets and the row-cache tests are scalar
// work.  MUST run while all 64 lanes of the wave are alive: v_readlane_b32 reads a lane's register whatever EXEC says,
// but a lane that left before this point never wrote it.
struct StripTaps {
  float fy;
  int yi;
};
__device__ __forceinline__ StripTaps strip_taps(uint32_t oy0, float scale_y) {
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const float sy = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(oy0 + lane), 0.5f), scale_y), -0.5f);
  const float y0f = __builtin_floorf(sy);
  StripTaps t;
  t.fy = __fadd_rn(sy, -y0f);
  t.yi = static_cast<int>(y0f);
  // pinned HERE: the values are pure functions of the lane id, and hipcc otherwise sinks them past the caller's
  // early return of the lanes beyond the row's end -- whose registers the other lanes read
  asm volatile("" : "+v"(t.fy), "+v"(t.yi));
  return t;
}

// Output column `ox_raw` of frame `f`, output rows [oy0, oy1) (at most 64).  `vt` = the strip's vertical taps,
// worked out by ALL 64 lanes of the wave before any of them left (strip_taps): lane i holds row oy0 + i.
//   TAPS_BYTES / TAPS_PAIRS / TAPS_WIDE: the lane fetches its own taps (see above); lanes past the
//     row's end must not call (a predicated store in their place cost 8 % in the same call).
//   TAPS_SHARED (layout as TAPS_WIDE, 64 * scale_x + 12 <= 252): the WAVE fetches a source row -- lane l
//     loads the l-th dword of the 256-byte span that starts at lane 0's window, one fully coalesced
//     access per plane (4 cache accesses per wave instruction against ~17 for per-lane 8-byte windows at
//     4-byte granularity) -- and a lane picks its windows out of its neighbours' registers with four
//     ds_bpermute_b32 when (and only when) the row is decoded.  Pays when most fetched rows are not
//     decoded, i.e. when enlarging; the launcher picks it for scale_y < 1.  All 64 lanes must call;
//     `live` masks the store.
//   TAPS_ONCE (CbCr plane 2-byte aligned, 63 * scale_x + 2 <= 63: enlarging): the WAVE decodes a source row ONCE.
//     Neighbouring output columns of an enlargement sit on the same source columns (at 2x every source pixel is a tap
//     of four lanes), and with per-lane taps every one of them runs the matrix and the three lookups again.  Here lane
//     l fetches and decodes source column wx0 + l (wx0 = lane 0's left tap; the 64 columns cover every tap of the
//     wave) -- one byte + one CbCr pair loaded, one pixel_rgb, three lookups instead of six -- and a lane takes the
//     linear values of its two taps out of its neighbours' registers with six ds_bpermute_b32 (no LDS bank conflicts,
//     no table traffic).  Same floats per source pixel whoever computes them: bit-identical output.  All 64 lanes must
//     call; `live` masks the store.
//   UNIFORM_ENCODE: the encode side goes through the uniform table (staged with sum_log2 = 0).
template <int TAPS, bool HAS_ALPHA, bool UNIFORM_ENCODE>
__device__ __forceinline__ void scaled_strip(const DecodeParams &p, const RescaleLookup &r,
                                             const FramePlanes &f, uint32_t ox_raw, uint32_t oy0, uint32_t oy1, const StripTaps &vt) {
  // TAPS_SHARED / TAPS_ONCE: every lane of the wave stays alive; one past the row's end works on the last column again and does not store
  constexpr bool BY_WAVE = TAPS == TAPS_SHARED || TAPS == TAPS_ONCE;
  const bool live = ox_raw < p.out_width;
  const uint32_t ox = BY_WAVE ? min(ox_raw, p.out_width - 1u) : ox_raw;

  const float sx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(ox), 0.5f), p.scale_x), -0.5f);
  const float x0f = __builtin_floorf(sx);
  const float fx = __fadd_rn(sx, -x0f), gx = __fadd_rn(1.0f, -fx);
  const int wmax = static_cast<int>(p.width) - 1, hmax = static_cast<int>(p.height) - 1;
  const int xi = static_cast<int>(x0f);
  const uint32_t xs[2] = {static_cast<uint32_t>(min(max(xi, 0), wmax)), static_cast<uint32_t>(min(max(xi + 1, 0), wmax))};
  const uint32_t cx[2] = {2u * (xs[0] >> 1), 2u * (xs[1] >> 1)};
  // TAPS_WIDE / TAPS_SHARED: 8-byte windows [ybase, ybase + 8) and [cbase, cbase + 8) hold both taps of a row
  const uint32_t ybase = min(xs[0] & ~3u, p.width - 8u), cbase = min(cx[0] & ~3u, p.width - 8u);
  const uint32_t ysel = ((xs[1] - ybase) << 8) | (xs[0] - ybase);  // v_perm_b32 selector: {Y0, Y1, -, -}
  const uint32_t k0 = cx[0] - cbase, k1 = cx[1] - cbase;
  const uint32_t csel = ((k1 + 1u) << 24) | (k1 << 16) | ((k0 + 1u) << 8) | k0;  // {Cb0, Cr0, Cb1, Cr1}
  // TAPS_SHARED: the wave's spans start at lane 0's windows (the windows move right with the lane);
  // ysrc / csrc = 4 * (lane that holds the first dword of this lane's window): the ds_bpermute address
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const uint32_t wybase = __builtin_amdgcn_readfirstlane(ybase), wcbase = __builtin_amdgcn_readfirstlane(cbase);
  const uint32_t yoff = min(wybase + 4u * lane, p.width - 4u), coff = min(wcbase + 4u * lane, p.width - 4u);
  const uint32_t ysrc = ybase - wybase, csrc = cbase - wcbase;
  // TAPS_ONCE: this lane's own source column (the wave's columns start at lane 0's left tap) and the ds_bpermute
  // addresses (4 * lane) of the lanes that hold its two taps
  const uint32_t wx0 = __builtin_amdgcn_readfirstlane(xs[0]);
  const uint32_t own_x = min(wx0 + lane, p.width - 1u), own_c = 2u * (own_x >> 1);
  const uint32_t tap_lane[2] = {4u * (xs[0] - wx0), 4u * (xs[1] - wx0)};

  // vertical taps of a row: the same for every lane, taken from the strip's lanes (strip_taps)
  struct RowTaps {
    int ys[2];
    float fy;
  };
  auto row_taps = [&](uint32_t oy) {
    RowTaps rt;
    const int k = static_cast<int>(oy - oy0);
    rt.fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vt.fy), k));
    const int yi = __builtin_amdgcn_readlane(vt.yi, k);
    rt.ys[0] = min(max(yi, 0), hmax);
    rt.ys[1] = min(max(yi + 1, 0), hmax);
    return rt;
  };
  // What the loads of one source row return, untouched: nothing consumes a load inside the block that
  // issues it, so the wait sits in front of the next row's decode, a whole iteration later.
  struct Fetched1 {
    uint32_t y[2], c[4], a[2];
  };
  // Planes as raw buffer resources: a row's offset rides in the instruction's SCALAR offset operand and the
  // lane's position in its 32-bit vector offset, so no address is formed in the VALU (with 64-bit global
  // pointers hipcc kept plane + lane offset in a VGPR pair and added the row offset per load).  The launcher
  // refuses planes of 2 GiB and more.
  auto plane = [](const uint8_t *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ry = plane(f.y), rc = plane(f.cbcr), ra = plane(HAS_ALPHA ? f.alpha : f.y), ro = plane(f.out);
  auto fetch_row = [&](int srow) {
    Fetched1 v = {};
    const int yo = srow * static_cast<int>(p.y_stride), co = (srow >> 1) * static_cast<int>(p.cbcr_stride);
    const int ao = HAS_ALPHA ? srow * static_cast<int>(p.alpha_stride) : 0;
    if (TAPS == TAPS_ONCE) {
      v.y[0] = __builtin_amdgcn_raw_buffer_load_b8(ry, own_x, yo, 0);
      v.c[0] = __builtin_amdgcn_raw_buffer_load_b16(rc, own_c, co, 0);
      if (HAS_ALPHA) v.a[0] = __builtin_amdgcn_raw_buffer_load_b8(ra, own_x, ao, 0);
    } else if (TAPS == TAPS_SHARED) {
      v.y[0] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff, yo, 0);
      v.c[0] = __builtin_amdgcn_raw_buffer_load_b32(rc, coff, co, 0);
      if (HAS_ALPHA) v.a[0] = __builtin_amdgcn_raw_buffer_load_b32(ra, yoff, ao, 0);
    } else if (TAPS == TAPS_WIDE) {
      const u32x2 yw = __builtin_amdgcn_raw_buffer_load_b64(ry, ybase, yo, 0);
      const u32x2 cw = __builtin_amdgcn_raw_buffer_load_b64(rc, cbase, co, 0);
      v.y[0] = yw.x, v.y[1] = yw.y, v.c[0] = cw.x, v.c[1] = cw.y;
      if (HAS_ALPHA) {
        const u32x2 aw = __builtin_amdgcn_raw_buffer_load_b64(ra, ybase, ao, 0);
        v.a[0] = aw.x, v.a[1] = aw.y;
      }
    } else {
      v.y[0] = __builtin_amdgcn_raw_buffer_load_b8(ry, xs[0], yo, 0);
      v.y[1] = __builtin_amdgcn_raw_buffer_load_b8(ry, xs[1], yo, 0);
      if (HAS_ALPHA) {
        v.a[0] = __builtin_amdgcn_raw_buffer_load_b8(ra, xs[0], ao, 0);
        v.a[1] = __builtin_amdgcn_raw_buffer_load_b8(ra, xs[1], ao, 0);
      }
      if (TAPS == TAPS_PAIRS) {
        v.c[0] = __builtin_amdgcn_raw_buffer_load_b16(rc, cx[0], co, 0);
        v.c[1] = __builtin_amdgcn_raw_buffer_load_b16(rc, cx[1], co, 0);
      } else {
        v.c[0] = __builtin_amdgcn_raw_buffer_load_b8(rc, cx[0], co, 0);
        v.c[1] = __builtin_amdgcn_raw_buffer_load_b8(rc, cx[0] + 1, co, 0);
        v.c[2] = __builtin_amdgcn_raw_buffer_load_b8(rc, cx[1], co, 0);
        v.c[3] = __builtin_amdgcn_raw_buffer_load_b8(rc, cx[1] + 1, co, 0);
      }
    }
    return v;
  };
  // this lane's two taps of that row: Y0 | Y1 << 8, Cb0 | Cr0 << 8 | Cb1 << 16 | Cr1 << 24, A0 | A1 << 8
  struct TapBytes {
    uint32_t yy, cc, aa;
  };
  auto tap_bytes = [&](const Fetched1 &v) {
    TapBytes t;
    t.aa = 0;
    if (TAPS == TAPS_SHARED) {
      const int ylo = __builtin_amdgcn_ds_bpermute(static_cast<int>(ysrc), static_cast<int>(v.y[0]));
      const int yhi = __builtin_amdgcn_ds_bpermute(static_cast<int>(ysrc + 4u), static_cast<int>(v.y[0]));
      const int clo = __builtin_amdgcn_ds_bpermute(static_cast<int>(csrc), static_cast<int>(v.c[0]));
      const int chi = __builtin_amdgcn_ds_bpermute(static_cast<int>(csrc + 4u), static_cast<int>(v.c[0]));
      t.yy = __builtin_amdgcn_perm(static_cast<uint32_t>(yhi), static_cast<uint32_t>(ylo), ysel);
      t.cc = __builtin_amdgcn_perm(static_cast<uint32_t>(chi), static_cast<uint32_t>(clo), csel);
      if (HAS_ALPHA) {
        const int alo = __builtin_amdgcn_ds_bpermute(static_cast<int>(ysrc), static_cast<int>(v.a[0]));
        const int ahi = __builtin_amdgcn_ds_bpermute(static_cast<int>(ysrc + 4u), static_cast<int>(v.a[0]));
        t.aa = __builtin_amdgcn_perm(static_cast<uint32_t>(ahi), static_cast<uint32_t>(alo), ysel);
      }
    } else if (TAPS == TAPS_WIDE) {
      t.yy = __builtin_amdgcn_perm(v.y[1], v.y[0], ysel);
      t.cc = __builtin_amdgcn_perm(v.c[1], v.c[0], csel);
      if (HAS_ALPHA) t.aa = __builtin_amdgcn_perm(v.a[1], v.a[0], ysel);
    } else {
      t.yy = v.y[0] | (v.y[1] << 8);
      if (HAS_ALPHA) t.aa = v.a[0] | (v.a[1] << 8);
      t.cc = TAPS == TAPS_PAIRS ? (v.c[0] | (v.c[1] << 16)) : (v.c[0] | (v.c[1] << 8) | (v.c[2] << 16) | (v.c[3] << 24));
    }
    return t;
  };

  // One source row of this lane: its two horizontal taps, linearised (times 2^-40).  Consecutive
  // output rows share source rows whenever the vertical ratio is below 2 (always when enlarging), and
  // which rows an output row needs is the same for every lane, so the two rows of the previous output
  // row stay in registers and only rows not seen yet are fetched and decoded (scalar branches).  The
  // chroma products are kept the same way: two luma rows share a CbCr row.
  struct RowLin {
    float v[6];  // R, G, B of tap 0; R, G, B of tap 1
    float a[2];  // byteNorm of the two alpha taps (alpha decoders)
  };
  int chroma_row = -1;
  Chroma ch0 = {}, ch1 = {};
  auto decode_row = [&](const Fetched1 &raw, int srow) {
    if (TAPS == TAPS_ONCE) {  // this lane's OWN source pixel, then the two taps from the lanes that hold them
      if ((srow >> 1) != chroma_row) {
        ch0 = chroma_terms(byte_of(raw.c[0], 0), byte_of(raw.c[0], 1));
        chroma_row = srow >> 1;
      }
      float x[4], own[4];
      pixel_rgb(byte_of(raw.y[0], 0), ch0, x[0], x[1], x[2]);
      x[3] = 0.0f;
      linearise3(r, x, own);
      if (HAS_ALPHA) own[3] = alpha_norm_arith(byte_of(raw.a[0], 0));
      RowLin rl;
      rl.a[0] = rl.a[1] = 0.0f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          rl.v[3 * t + k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(static_cast<int>(tap_lane[t]), __builtin_bit_cast(int, own[k])));
        if (HAS_ALPHA) rl.a[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(static_cast<int>(tap_lane[t]), __builtin_bit_cast(int, own[3])));
      }
      return rl;
    }
    const TapBytes fr = tap_bytes(raw);
    if ((srow >> 1) != chroma_row) {
      ch0 = chroma_terms(byte_of(fr.cc, 0), byte_of(fr.cc, 1));
      ch1 = chroma_terms(byte_of(fr.cc, 2), byte_of(fr.cc, 3));
      chroma_row = srow >> 1;
    }
    float x[6];
    pixel_rgb(byte_of(fr.yy, 0), ch0, x[0], x[1], x[2]);
    pixel_rgb(byte_of(fr.yy, 1), ch1, x[3], x[4], x[5]);
    RowLin rl;
    linearise6(r, x, rl.v);
    rl.a[0] = rl.a[1] = 0.0f;
    if (HAS_ALPHA) {
      rl.a[0] = alpha_norm_arith(byte_of(fr.aa, 0));
      rl.a[1] = alpha_norm_arith(byte_of(fr.aa, 1));
    }
    return rl;
  };

  int have_top = -1, have_bot = -1;  // source rows held in `top` / `bot`
  RowLin top = {}, bot = {};
  // The loads of a fetched row are waited for HERE whether or not the row gets decoded: a load still in
  // flight at a skipped decode would leave its destination registers pending, and hipcc then drains vmcnt
  // (stores included) wherever it reuses one of them.  No instruction is emitted, only the s_waitcnt.
  auto landed = [&](const Fetched1 &v) {
    if (BY_WAVE) asm volatile("" ::"v"(v.y[0]), "v"(v.c[0]));
    else if (TAPS == TAPS_BYTES) asm volatile("" ::"v"(v.y[0]), "v"(v.y[1]), "v"(v.c[0]), "v"(v.c[1]), "v"(v.c[2]), "v"(v.c[3]));
    else asm volatile("" ::"v"(v.y[0]), "v"(v.y[1]), "v"(v.c[0]), "v"(v.c[1]));
    if (HAS_ALPHA) {
      if (BY_WAVE) asm volatile("" ::"v"(v.a[0]));
      else asm volatile("" ::"v"(v.a[0]), "v"(v.a[1]));
    }
  };
  // one output row from the fetched bytes of its two source rows
  auto output_row = [&](uint32_t oy, const RowTaps &rt, const Fetched1 &f0, const Fetched1 &f1) {
    landed(f0);
    landed(f1);
    if (rt.ys[0] == have_bot) top = bot;  // the previous bottom row is this row's top row
    else if (rt.ys[0] != have_top) top = decode_row(f0, rt.ys[0]);
    if (rt.ys[1] == rt.ys[0]) bot = top;  // both taps clamped onto one row
    else if (rt.ys[1] != have_bot) bot = decode_row(f1, rt.ys[1]);
    have_top = rt.ys[0];
    have_bot = rt.ys[1];
    const float fy = rt.fy, gy = __fadd_rn(1.0f, -fy);
    const float w[4] = {__fmul_rn(gx, gy), __fmul_rn(fx, gy), __fmul_rn(gx, fy), __fmul_rn(fx, fy)};
    float acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      acc[k] = __fmul_rn(w[0], top.v[k]);
      acc[k] = __fadd_rn(acc[k], __fmul_rn(w[1], top.v[3 + k]));
      acc[k] = __fadd_rn(acc[k], __fmul_rn(w[2], bot.v[k]));
      acc[k] = __fadd_rn(acc[k], __fmul_rn(w[3], bot.v[3 + k]));
    }
    const uint32_t R = UNIFORM_ENCODE ? encode_byte_uniform(r, acc[0]) : encode_byte(r, acc[0]);
    const uint32_t G = UNIFORM_ENCODE ? encode_byte_uniform(r, acc[1]) : encode_byte(r, acc[1]);
    const uint32_t B = UNIFORM_ENCODE ? encode_byte_uniform(r, acc[2]) : encode_byte(r, acc[2]);
    uint32_t aw = p.alpha_word;
    if (HAS_ALPHA) {
      float av = __fmul_rn(w[0], top.a[0]);
      av = __fadd_rn(av, __fmul_rn(w[1], top.a[1]));
      av = __fadd_rn(av, __fmul_rn(w[2], bot.a[0]));
      av = __fadd_rn(av, __fmul_rn(w[3], bot.a[1]));
      aw = alpha_word_of(av);
    }
    if (!BY_WAVE || live)
      __builtin_amdgcn_raw_buffer_store_b32(pack_bgra(R, G, B, aw), ro, ox * 4u, oy * p.out_stride, kScaledStoreAux);
  };

  // The FETCH is unconditional and one output row ahead (a load the row does not need after all is an L2
  // hit; fetching only the new rows was measured: -22 % instructions, but the waits then covered the loads
  // just issued); the DECODE is what is skipped.  Two output rows per trip through explicit A / B register
  // sets: rotating one set through copies made hipcc drain vmcnt -- the row's STORE included -- at the end
  // of every row.
  // (Past the strip's end the fetch repeats the last row instead of being branched around: hipcc's vmcnt
  // accounting takes the path with the fewest loads in flight, so one conditional fetch turns every
  // wait of the loop into a full drain.)
  const uint32_t last = oy1 - 1;
  // HOW FAR AHEAD (round 6).  gfx950 counts loads and stores in ONE in-order counter (vmcnt): waiting for the loads of row j
  // also waits for every store issued before them.  One row ahead, the store of row j - 2 must have been acknowledged when row
  // j starts -- and a wave's row takes ~1.2 us here, about what a store takes to come back from HBM under this write load: with
  // the stores deleted, or the loads, the enlarging launch runs 27 % faster, with every lookup and all arithmetic deleted 9 %
  // (profiles/r06_ab_scaled_parts.txt).  Fetching D rows ahead gives a store D row-times.  D + 1 register sets, D + 1 rows per trip.
  constexpr int D = BY_WAVE ? kScaledAheadWave : kScaledAhead;
  if constexpr (D == 1) {
    RowTaps rta = row_taps(oy0), rtb;
    Fetched1 a0 = fetch_row(rta.ys[0]), a1 = fetch_row(rta.ys[1]), b0, b1;
    for (uint32_t oy = oy0; oy < oy1; oy += 2) {
      rtb = row_taps(min(oy + 1, last));
      b0 = fetch_row(rtb.ys[0]);
      b1 = fetch_row(rtb.ys[1]);
      output_row(oy, rta, a0, a1);
      if (oy + 1 >= oy1) {  // uniform.  Nothing stays in flight past the strip: a dangling load is a pending
        landed(b0);         // write to registers the next strip reuses, i.e. a drain in every trip of ITS loop
        landed(b1);
        break;
      }
      rta = row_taps(min(oy + 2, last));
      a0 = fetch_row(rta.ys[0]);
      a1 = fetch_row(rta.ys[1]);
      output_row(oy + 1, rtb, b0, b1);
    }
    landed(a0);
    landed(a1);
  } else {
    struct RowFetch {
      RowTaps rt;
      Fetched1 f0, f1;
    };
    auto fetch_for = [&](uint32_t oy) {  // past the strip's end the last row again (see above)
      RowFetch q;
      q.rt = row_taps(min(oy, last));
      q.f0 = fetch_row(q.rt.ys[0]);
      q.f1 = fetch_row(q.rt.ys[1]);
      return q;
    };
    RowFetch s[D + 1];
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = fetch_for(oy0 + static_cast<uint32_t>(k));
    for (uint32_t oy = oy0; oy < oy1; oy += D + 1) {
#pragma unroll
      for (int u = 0; u <= D; ++u) {
        s[(u + D) % (D + 1)] = fetch_for(oy + static_cast<uint32_t>(u + D));
        if (oy + static_cast<uint32_t>(u) < oy1) {  // uniform
          output_row(oy + static_cast<uint32_t>(u), s[u].rt, s[u].f0, s[u].f1);
        } else {
          landed(s[u].f0);
          landed(s[u].f1);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {  // nothing stays in flight past the strip
      landed(s[k].f0);
      landed(s[k].f1);
    }
  }
}

// A workgroup = 256 output columns x kScaledStrips strips of `scaled_rows` output rows of one frame; its
// waves share nothing but the single-copy tables.  Measured on 4K -> 1440p, 8 frames per launch, same call
// (profiles/r02_ab_scaled.txt): one strip + the (then) 6 KiB two-resolution encode table (14 KiB staged per
// workgroup) 240 Gpixel/s; two strips per workgroup 224; the 24 KiB uniform encode table (9 fewer VALU
// instructions per pixel, but 32 KiB staged per 4 096 output pixels and 5 workgroups per CU) 194.
#ifndef BT709_SCALED_STRIPS
#define BT709_SCALED_STRIPS 1
#endif
#ifndef BT709_SCALED_UNIFORM
#define BT709_SCALED_UNIFORM 0
#endif
// interleaved copies of the decode-side table in a scaled-kernel workgroup (2^n; lane l reads copy l & (2^n - 1)).
// Measured with the persistent strips (round 3, tools/bench_scaled_set.sh, same call): see the launcher.
#ifndef BT709_SCALED_DEC_COPIES_LOG2
#define BT709_SCALED_DEC_COPIES_LOG2 0
#endif
constexpr uint32_t kScaledDecCopiesLog2 = BT709_SCALED_DEC_COPIES_LOG2;
#ifndef BT709_SCALED_ENC_COPIES_LOG2
#define BT709_SCALED_ENC_COPIES_LOG2 0
#endif
constexpr uint32_t kScaledEncCopiesLog2 = BT709_SCALED_ENC_COPIES_LOG2;  // interleaved copies of the 5 KiB encode-side table
constexpr uint32_t kScaledStrips = BT709_SCALED_STRIPS;
constexpr bool kScaledUniform = BT709_SCALED_UNIFORM != 0;
// PERSISTENT: the launch has as many workgroups as the chip holds at once and workgroup g takes the work items g,
// g + G, ... (an item = 256 columns x the workgroup's strips of one frame, column tiles fastest, so the items in
// flight are neighbours in memory): the 14 KiB of tables are staged once per workgroup of the LAUNCH instead of
// once per 4 096 output pixels.  Same call, 4K -> 1440p x 8: 240 -> 258 Gpixel/s; one frame 167 -> 175.  Not used
// with TAPS_SHARED: the loop costs that variant 5 VGPRs = one wave per SIMD of occupancy (1080p -> 4K: 396 -> 352).
template <int TAPS, bool HAS_ALPHA, bool PERSISTENT>
__global__ void __launch_bounds__(kBlockThreads *kScaledStrips)
decode_nv12_scaled(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const RescaleLookup r = stage_rescale_tables<kScaledUniform>(lds_raw, p, kScaledDecCopiesLog2, kScaledEncCopiesLog2, 0);
  __syncthreads();
  if (PERSISTENT) {
    const uint32_t strips = (p.out_height + p.scaled_rows - 1) / p.scaled_rows;
    const uint32_t strip_groups = (strips + kScaledStrips - 1) / kScaledStrips;
    for (uint32_t item = blockIdx.x; item < p.tile_rows; item += gridDim.x) {
      const uint32_t tile = item % p.tiles_x, rest = item / p.tiles_x;
      const uint32_t sg = rest % strip_groups, frame = rest / strip_groups;
      const FramePlanes f = frame_planes(p, frame);
      const uint32_t ox = tile * blockDim.x + threadIdx.x;
      const uint32_t oy0 = (sg * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y)) * p.scaled_rows;
      if (oy0 >= p.out_height) continue;  // the whole wave
      const StripTaps vt = strip_taps(oy0, p.scale_y);  // before any lane is masked off
      if (TAPS == TAPS_SHARED || TAPS == TAPS_ONCE || ox < p.out_width)
        scaled_strip<TAPS, HAS_ALPHA, kScaledUniform>(p, r, f, ox, oy0, min(oy0 + p.scaled_rows, p.out_height), vt);
    }
    return;
  }
  const FramePlanes f = frame_planes(p, blockIdx.z);
  const uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t oy0 = (blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y)) * p.scaled_rows;
  if (oy0 >= p.out_height) return;  // the whole wave
  const StripTaps vt = strip_taps(oy0, p.scale_y);  // before any lane leaves
  if (TAPS != TAPS_SHARED && TAPS != TAPS_ONCE && ox >= p.out_width) return;  // TAPS_SHARED / TAPS_ONCE: the wave works together
  scaled_strip<TAPS, HAS_ALPHA, kScaledUniform>(p, r, f, ox, oy0, min(oy0 + p.scaled_rows, p.out_height), vt);
}

// ---------------------------------------------------------------------------
// Pass 2 ALONE: -[MetalScaleRenderContext renderScaled:...] + samplingShader
// (Renderer/MetalScaleRenderContext.m:55-105, AAPLShaders.metal:73-85) for a caller that keeps the
// reference's two passes, or whose pass 1 rendered into RGBA16Float.  Same sampling geometry,
// weights and summation order as decode_nv12_scaled, so pass 1 into a BGRA8 intermediate followed by
// this kernel equals the fused kernel bit for bit.  One lane per output column walking `rows` rows.
//   IN_RGBA16F = false: a tap is a BGRA8 word; rgb linearised through lin[256] (the sRGB8 sampler's
//                decode), alpha a plain unorm (byte * (1/255f))
//   IN_RGBA16F = true:  a tap is four halves, linear light already (v_cvt_f32_f16)
// The sum is saturated (a unorm render target clamps), rgb goes through the sRGB-encode table,
// alpha is round(255 v) in arithmetic (alpha_word_of).
// ---------------------------------------------------------------------------
template <bool IN_RGBA16F>
__global__ void __launch_bounds__(kBlockThreads)
render_scaled(const RenderParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  {  // stage: lin[256] | encode buckets.  lin[] sits at LDS address 0 (this kernel has no static LDS, so its dynamic segment
     // starts there; trapped below if that ever changes): a texel's byte then becomes its table address by ONE SDWA shift.
    const uint32_t tid = threadIdx.x, n = blockDim.x;
    u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
    const u32x4 *e = reinterpret_cast<const u32x4 *>(p.table_encode);
    const u32x4 *l = reinterpret_cast<const u32x4 *>(p.table_lin);
    const uint32_t ne = p.table_encode_bytes / 16;
    stage_batched(d, ne + 64u, tid, n, [&](uint32_t i) { return i < 64u ? l[i] : e[i - 64u]; });  // one batch: both tables' loads in flight together
    if (lds_address(lds_raw) != 0u) __builtin_trap();
  }
  __syncthreads();
  RescaleLookup r = {};
  r.enc_shift = 3;
  r.enc_off = 1024u;  // behind lin[256]
  r.enc_add = p.encode_log_add;
  r.enc_log_off = r.enc_off - (p.encode_log_first << r.enc_shift);
  r.quarter_unscale = 1.0f;  // the filter's sums are unit-range values here: the table's own domain, edges staged as they are
  typedef __attribute__((address_space(3))) const float *LdsFloatPtr;
  uint32_t two = 2u;  // SDWA operands cannot be inline constants
  asm("" : "+v"(two));

  const uint32_t oy0 = blockIdx.y * p.rows, oy1 = min(oy0 + p.rows, p.out_height);
  const StripTaps vt = strip_taps(oy0, p.scale_y);  // before any lane leaves
  const uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x;
  if (ox >= p.out_width) return;
  const float sx = __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(ox), 0.5f), p.scale_x), -0.5f);
  const float x0f = __builtin_floorf(sx);
  const float fx = __fadd_rn(sx, -x0f), gx = __fadd_rn(1.0f, -fx);
  const int wmax = static_cast<int>(p.width) - 1, hmax = static_cast<int>(p.height) - 1;
  const int xi = static_cast<int>(x0f);
  const uint32_t xs[2] = {static_cast<uint32_t>(min(max(xi, 0), wmax)), static_cast<uint32_t>(min(max(xi + 1, 0), wmax))};
  constexpr uint32_t kTexel = IN_RGBA16F ? 8u : 4u;

  // Same walk as decode_nv12_scaled (see scaled_strip): vertical taps once per strip (lane i does row i,
  // rows read them with v_readlane_b32), the two LINEARISED source rows of the previous output row kept
  // in registers and only rows not seen yet converted, fetches unconditional and one output row ahead in
  // two explicit register sets, the intermediate as a raw buffer resource (scalar row offset).
  struct RowTaps {
    int ys[2];
    float fy;
  };
  auto row_taps = [&](uint32_t oy) {
    RowTaps rt;
    const int k = static_cast<int>(oy - oy0);
    rt.fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vt.fy), k));
    const int yi = __builtin_amdgcn_readlane(vt.yi, k);
    rt.ys[0] = min(max(yi, 0), hmax);
    rt.ys[1] = min(max(yi + 1, 0), hmax);
    return rt;
  };
  // surface blockIdx.z of a batched launch (bt709hip_render_scaled_batch: evenly spaced surfaces)
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t *>(p.in) + static_cast<int64_t>(blockIdx.z) * p.in_step, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(p.out + static_cast<int64_t>(blockIdx.z) * p.out_step, 0, 0x7fffffff, 0x00020000);
  struct Fetched {  // the two texels of a source row, untouched
    uint32_t w[IN_RGBA16F ? 4 : 2];
  };
  auto fetch_row = [&](int srow) {
    Fetched v;
    const int ro = srow * static_cast<int>(p.in_stride);
    if (IN_RGBA16F) {
      const u32x2 t0 = __builtin_amdgcn_raw_buffer_load_b64(rin, xs[0] * kTexel, ro, 0);
      const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(rin, xs[1] * kTexel, ro, 0);
      v.w[0] = t0.x, v.w[1] = t0.y, v.w[2] = t1.x, v.w[3] = t1.y;
    } else {
      v.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rin, xs[0] * kTexel, ro, 0);
      v.w[1] = __builtin_amdgcn_raw_buffer_load_b32(rin, xs[1] * kTexel, ro, 0);
    }
    return v;
  };
  auto landed = [&](const Fetched &v) {
    if (IN_RGBA16F) asm volatile("" ::"v"(v.w[0]), "v"(v.w[1]), "v"(v.w[2]), "v"(v.w[3]));
    else asm volatile("" ::"v"(v.w[0]), "v"(v.w[1]));
  };
  struct RowLin {
    float s[8];  // R, G, B, A of tap 0; R, G, B, A of tap 1: what the sampler hands the filter
  };
  auto convert_row = [&](const Fetched &f) {
    RowLin rl;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float *s = rl.s + 4 * t;
      if (IN_RGBA16F) {
        const uint32_t lo = f.w[2 * t], hi = f.w[2 * t + 1];
        s[0] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(lo & 0xffffu)));
        s[1] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(lo >> 16)));
        s[2] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(hi & 0xffffu)));
        s[3] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(hi >> 16)));
      } else {
        const uint32_t v = f.w[t];
        // lin[byte]: byte select and << 2 in one v_lshlrev_b32_sdwa (the encoder's form, bt709_encode.hip byte_entry)
        uint32_t ar, ag, ab;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(ar) : "v"(two), "v"(v));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ag) : "v"(two), "v"(v));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(ab) : "v"(two), "v"(v));
        s[0] = *reinterpret_cast<LdsFloatPtr>(ar);  // R: byte 2
        s[1] = *reinterpret_cast<LdsFloatPtr>(ag);  // G: byte 1
        s[2] = *reinterpret_cast<LdsFloatPtr>(ab);  // B: byte 0
        s[3] = __fmul_rn(byte_of(v, 3), kInv255);                                // byteNorm
      }
    }
    return rl;
  };
  int have_top = -1, have_bot = -1;
  RowLin top = {}, bot = {};
  auto output_row = [&](uint32_t oy, const RowTaps &rt, const Fetched &f0, const Fetched &f1) {
    landed(f0);
    landed(f1);
    if (rt.ys[0] == have_bot) top = bot;
    else if (rt.ys[0] != have_top) top = convert_row(f0);
    if (rt.ys[1] == rt.ys[0]) bot = top;
    else if (rt.ys[1] != have_bot) bot = convert_row(f1);
    have_top = rt.ys[0];
    have_bot = rt.ys[1];
    const float fy = rt.fy, gy = __fadd_rn(1.0f, -fy);
    const float w[4] = {__fmul_rn(gx, gy), __fmul_rn(fx, gy), __fmul_rn(gx, fy), __fmul_rn(fx, fy)};
    float acc[4];  // R, G, B, A
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] = __fmul_rn(w[0], top.s[k]);
      acc[k] = __fadd_rn(acc[k], __fmul_rn(w[1], top.s[4 + k]));
      acc[k] = __fadd_rn(acc[k], __fmul_rn(w[2], bot.s[k]));
      acc[k] = __fadd_rn(acc[k], __fmul_rn(w[3], bot.s[4 + k]));
    }
    const uint32_t R = encode_byte(r, add_sat(acc[0], 0.0f));
    const uint32_t G = encode_byte(r, add_sat(acc[1], 0.0f));
    const uint32_t B = encode_byte(r, add_sat(acc[2], 0.0f));
    const uint32_t A = alpha_word_of(acc[3]);
    __builtin_amdgcn_raw_buffer_store_b32(pack_bgra(R, G, B, A), rout, ox * 4u, oy * p.out_stride, kScaledStoreAux);
  };
  const uint32_t last = oy1 - 1;
  // rows fetched ahead: as in scaled_strip ("HOW FAR AHEAD"): D + 1 register sets, D + 1 rows per trip
  constexpr int D = IN_RGBA16F ? kRenderAhead16 : kRenderAhead8;
  struct RowFetch {
    RowTaps rt;
    Fetched f0, f1;
  };
  auto fetch_for = [&](uint32_t oy) {  // past the strip's end the last row again
    RowFetch q;
    q.rt = row_taps(min(oy, last));
    q.f0 = fetch_row(q.rt.ys[0]);
    q.f1 = fetch_row(q.rt.ys[1]);
    return q;
  };
  RowFetch s[D + 1];
#pragma unroll
  for (int k = 0; k < D; ++k) s[k] = fetch_for(oy0 + static_cast<uint32_t>(k));
  for (uint32_t oy = oy0; oy < oy1; oy += D + 1) {
#pragma unroll
    for (int u = 0; u <= D; ++u) {
      s[(u + D) % (D + 1)] = fetch_for(oy + static_cast<uint32_t>(u + D));
      if (oy + static_cast<uint32_t>(u) < oy1) {  // uniform
        output_row(oy + static_cast<uint32_t>(u), s[u].rt, s[u].f0, s[u].f1);
      } else {
        landed(s[u].f0);
        landed(s[u].f1);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {  // nothing stays in flight past the strip
    landed(s[k].f0);
    landed(s[k].f1);
  }
}

const char *launch_render_scaled(const RenderParams &p_in, int frames, bool in_rgba16f, uint32_t compute_units, hipStream_t stream) {
  RenderParams p = p_in;
  const uint32_t cols = (p.out_width + kBlockThreads - 1) / kBlockThreads;
  const uint64_t want = 8ull * (compute_units ? compute_units : 256u);
  uint32_t rows = static_cast<uint32_t>(static_cast<uint64_t>(cols) * p.out_height * static_cast<uint32_t>(frames) / want);
  rows = rows < 1 ? 1 : (rows > 16 ? 16 : rows);
  p.rows = rows;
  // the kernel forms row offsets in 32 bits
  if (static_cast<uint64_t>(p.height) * p.in_stride >= (1ull << 31) || static_cast<uint64_t>(p.out_height) * p.out_stride >= (1ull << 31)) return nullptr;
  const dim3 grid(cols, (p.out_height + rows - 1) / rows, static_cast<uint32_t>(frames));
  const size_t lds = static_cast<size_t>(p.table_encode_bytes) + 1024;
  if (in_rgba16f) hipLaunchKernelGGL(render_scaled<true>, grid, dim3(kBlockThreads), lds, stream, p);
  else hipLaunchKernelGGL(render_scaled<false>, grid, dim3(kBlockThreads), lds, stream, p);
  return in_rgba16f ? "render_scaled<rgba16f>" : "render_scaled<bgra8>";
}

const char *launch_decode_scaled(const DecodeParams &p_in, int frames, bool has_alpha, uint32_t in_align,
                                 uint32_t compute_units, hipStream_t stream) {
  DecodeParams p = p_in;
  const uint32_t cus = compute_units ? compute_units : 256u;
  // the kernels form row offsets in 32 bits
  const uint64_t kPlaneLimit = 1ull << 31;
  if (static_cast<uint64_t>(p.height) * p.y_stride >= kPlaneLimit || static_cast<uint64_t>(p.height / 2) * p.cbcr_stride >= kPlaneLimit ||
      (has_alpha && static_cast<uint64_t>(p.height) * p.alpha_stride >= kPlaneLimit) ||
      static_cast<uint64_t>(p.out_height) * p.out_stride >= kPlaneLimit)
    return nullptr;
  // widest tap fetch the layout allows (see the kernel); the frame spacing of a uniform batch counts too
  uint32_t align = in_align > 4 ? 4 : in_align;
  auto fold = [&align](uint64_t v) { while (align > 1 && v % align) align /= 2; };
  if (p.uniform) fold(static_cast<uint64_t>(p.step_y)), fold(static_cast<uint64_t>(p.step_cbcr)), fold(static_cast<uint64_t>(p.step_alpha));
  int taps = (align == 4 && p.width % 4 == 0 && p.width >= 8) ? TAPS_WIDE : (align >= 2 ? TAPS_PAIRS : TAPS_BYTES);
  // Fetching by the wave pays when most fetched rows are not decoded (enlarging: the two loads per output
  // row dominate) and costs when they are (four ds_bpermute per decoded row on an LDS pipe the lookups
  // keep busy): 1080p -> 4K +6.6 %, 4K -> 1440p -8 % (same call).  A wave's 64 windows must fit one 256-byte span.
#ifndef BT709_SCALED_SHARED_BELOW
#define BT709_SCALED_SHARED_BELOW 1.0f
#endif
  if (taps == TAPS_WIDE && p.scale_y < BT709_SCALED_SHARED_BELOW && p.scale_x * 64.0f + 12.0f <= 252.0f) taps = TAPS_SHARED;
  // Enlarging horizontally: the wave decodes each source pixel once (TAPS_ONCE).  A wave's taps must lie within 64
  // source columns of lane 0's left tap: x0(lane 63) - x0(lane 0) <= floor(63 scale_x) + 1, plus one for the right tap.
#ifndef BT709_SCALED_ONCE_BELOW
#define BT709_SCALED_ONCE_BELOW 0.95f  // 63 * 0.95 + 2 = 61.85: within the 64 lanes with margin for the rounding of sx
#endif
#ifndef BT709_SCALED_ONCE_PERSISTENT
#define BT709_SCALED_ONCE_PERSISTENT 0
#endif
  if (align >= 2 && p.scale_x <= BT709_SCALED_ONCE_BELOW && p.scale_y < BT709_SCALED_SHARED_BELOW) taps = TAPS_ONCE;

  // rows per strip: as many as still leave ~8 workgroups per CU (table staging is per workgroup)
#ifndef BT709_SCALED_WG_PER_CU
#define BT709_SCALED_WG_PER_CU 8
#endif
#ifndef BT709_SCALED_MAX_ROWS
#define BT709_SCALED_MAX_ROWS 16
#endif
  static_assert(BT709_SCALED_MAX_ROWS <= 64, "one lane per row of a strip works out its vertical taps");
  const uint32_t cols = (p.out_width + kBlockThreads - 1) / kBlockThreads;
  const uint64_t want = static_cast<uint64_t>(BT709_SCALED_WG_PER_CU) * cus * kScaledStrips;
  uint32_t rows = static_cast<uint32_t>(static_cast<uint64_t>(cols) * p.out_height * static_cast<uint32_t>(frames) / want);
#ifndef BT709_SCALED_MAX_ROWS_WAVE
#define BT709_SCALED_MAX_ROWS_WAVE 32  // the by-wave forms fetch 3 rows ahead: a longer strip pays its prologue and drain less often
#endif
  static_assert(BT709_SCALED_MAX_ROWS_WAVE <= 64, "one lane per row of a strip works out its vertical taps");
  const uint32_t max_rows = (taps == TAPS_ONCE || taps == TAPS_SHARED) ? BT709_SCALED_MAX_ROWS_WAVE : BT709_SCALED_MAX_ROWS;
  rows = rows < 1 ? 1 : (rows > max_rows ? max_rows : rows);
  // the by-wave forms produce kScaledAheadWave + 1 rows per trip of their loop: whole trips only (a partial trip still fetches for all its rows)
  if ((taps == TAPS_ONCE || taps == TAPS_SHARED) && rows > static_cast<uint32_t>(kScaledAheadWave + 1)) rows -= rows % static_cast<uint32_t>(kScaledAheadWave + 1);
  const size_t lds = (static_cast<size_t>(p.table_linear_bytes) << kScaledDecCopiesLog2) + ((kScaledUniform ? p.table_encode_u_bytes : p.table_encode_bytes) << kScaledEncCopiesLog2);
  const dim3 block(kBlockThreads, kScaledStrips);
  const bool persistent = taps == TAPS_ONCE ? BT709_SCALED_ONCE_PERSISTENT != 0 : taps != TAPS_SHARED;
  const void *fn = nullptr;
#define BT709_PICK_SCALED(T, P)                                                                                        \
  fn = has_alpha ? reinterpret_cast<const void *>(&decode_nv12_scaled<T, true, P>) : reinterpret_cast<const void *>(&decode_nv12_scaled<T, false, P>)
  if (taps == TAPS_ONCE) BT709_PICK_SCALED(TAPS_ONCE, BT709_SCALED_ONCE_PERSISTENT != 0);
  else if (taps == TAPS_SHARED) BT709_PICK_SCALED(TAPS_SHARED, false);
  else if (taps == TAPS_WIDE) BT709_PICK_SCALED(TAPS_WIDE, true);
  else if (taps == TAPS_PAIRS) BT709_PICK_SCALED(TAPS_PAIRS, true);
  else BT709_PICK_SCALED(TAPS_BYTES, true);
#undef BT709_PICK_SCALED
  uint64_t resident = 0;
  {
    // as many workgroups as the chip holds at once (what the registers and the tables' LDS allow per CU).  The answer
    // depends on the kernel variant, on the dynamic LDS (the decode-side table's size follows the gamma's bucket count)
    // and on the device: a small cache keyed on all three (a miss just asks
    // again; a torn entry can only mis-size the grid -- the item loop strides by gridDim.x -- never change a result)
    struct Occupancy {
      std::atomic<const void *> fn{nullptr};
      std::atomic<uint64_t> key{0};
      std::atomic<int> per_cu{0};
    };
    static Occupancy cache[16];
    int device = 0;
    (void)hipGetDevice(&device);
    const uint64_t key = (static_cast<uint64_t>(lds) << 16) | static_cast<uint32_t>(device & 0xffff);
    Occupancy &slot = cache[((reinterpret_cast<uintptr_t>(fn) >> 4) ^ lds ^ static_cast<uint32_t>(device)) & 15];
    int per_cu = 0;
    if (slot.fn.load(std::memory_order_acquire) == fn && slot.key.load(std::memory_order_relaxed) == key)
      per_cu = slot.per_cu.load(std::memory_order_relaxed);
    if (per_cu == 0) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, static_cast<int>(kBlockThreads * kScaledStrips), lds) != hipSuccess || per_cu < 1)
        per_cu = 4;
      slot.fn.store(nullptr, std::memory_order_release);  // invalidate while the fields change
      slot.key.store(key, std::memory_order_relaxed);
      slot.per_cu.store(per_cu, std::memory_order_relaxed);
      slot.fn.store(fn, std::memory_order_release);
    }
    resident = static_cast<uint64_t>(per_cu) * cus;
#ifndef BT709_SCALED_BALANCE
#define BT709_SCALED_BALANCE 1
#endif
    // ONE GENERATION (round 6): the resident workgroups take the items w, w + G, ...  A launch small enough for every workgroup
    // to get ONE item should be cut that way: one 4K -> 1440p frame in strips of 7 rows (the 8-per-CU rule) is 2 057 items for
    // 1 280 workgroups -- two for most, one for the rest -- in strips of 12 rows 1 200 items, one each: 19.6 -> 18.8 us.  Longer
    // launches keep the rule (balancing them by the same count of items per workgroup measured 4-7 % SLOWER: the workgroups do not
    // march in generations; profiles/r06_ab_scaled_ahead.txt).
    // The forms that are not persistent (one workgroup per item, dispatched by the hardware) have the same tail: 2 700 workgroups
    // for 2 048 places are 1.3 generations.  Their strips stay whole trips of the fetch loop (kScaledAheadWave + 1 rows).
    const uint32_t step = persistent ? 1u : static_cast<uint32_t>(kScaledAheadWave + 1);
    if (BT709_SCALED_BALANCE && kScaledStrips == 1 && static_cast<uint64_t>(cols) * p.out_height * static_cast<uint32_t>(frames) <= resident * max_rows) {
      for (uint32_t r = 4; r <= max_rows; r += step)
        if (static_cast<uint64_t>(cols) * ((p.out_height + r - 1) / r) * static_cast<uint32_t>(frames) <= resident) {
          rows = r;
          break;
        }
    }
  }
  p.scaled_rows = rows;
  const uint32_t strips = (p.out_height + rows - 1) / rows;
  const uint32_t strip_groups = (strips + kScaledStrips - 1) / kScaledStrips;
  dim3 grid(cols, strip_groups, static_cast<uint32_t>(frames));
  if (persistent) {
    const uint64_t items = static_cast<uint64_t>(cols) * strip_groups * static_cast<uint32_t>(frames);
    if (items > 0x7fffffffull) return nullptr;
    p.tiles_x = cols;
    p.tile_rows = static_cast<uint32_t>(items);
    grid = dim3(static_cast<uint32_t>(items < resident ? items : resident), 1, 1);
  }
  void *args[] = {&p};
  (void)hipLaunchKernel(fn, grid, block, args, lds, stream);  // a failure is picked up by the caller's hipGetLastError
  return has_alpha ? "decode_nv12_scaled<alpha>" : "decode_nv12_scaled";
}

hipError_t prepare_scaled_kernels() {
  const int cap = static_cast<int>(kRepLdsBytes);  // gfx950: 160 KiB LDS per workgroup
  const void *fns[] = {
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_BYTES, false, true>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_PAIRS, false, true>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_WIDE, false, true>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_BYTES, true, true>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_PAIRS, true, true>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_WIDE, true, true>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_SHARED, false, false>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_SHARED, true, false>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_ONCE, false, BT709_SCALED_ONCE_PERSISTENT != 0>),
      reinterpret_cast<const void *>(&decode_nv12_scaled<TAPS_ONCE, true, BT709_SCALED_ONCE_PERSISTENT != 0>),
      reinterpret_cast<const void *>(&render_scaled<true>),
      reinterpret_cast<const void *>(&render_scaled<false>),
  };
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace bt709
