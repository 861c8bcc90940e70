// CDNA4 (gfx950) kernels of the FUSED decode + EXACT 2:1 rescale (BASELINE config 4, 8K -> 4K): what the reference does in two
// Metal passes -- BT709ToLinearSRGBKernel into an sRGB8 intermediate (Renderer/AAPLShaders.metal:336-407), then
// MetalScaleRenderContext -renderScaled: / samplingShader (73-85) -- is ONE kernel here: the 132.7 MB 8K intermediate is never
// written.  Arithmetic and tables: bt709_rescale.h.
//
// The kernels are bound by VALU issue slots and by the LDS pipe together (12 decode-side + 3 encode-side lookups per output
// pixel): per decode-side lookup a saturating add, the magic add, the address, a subtract and a median-of-three
// select.  The alpha channel of an alpha decoder is pure arithmetic.
//
//   decode_nv12_half       one short-lived workgroup per tile of an output row (small launches, any layout)
//   decode_nv12_half_rep   persistent workgroups, bank-conflict-free LDS tables (with and without alpha)
#include "bt709_rescale.h"

namespace bt709 {
namespace {

// One output pixel of the exact 2:1 rescale: the four source pixels of a 2x2 block share one
// CbCr sample; (((a+b)+c)+d) * 0.25f per channel.
template <bool UNIFORM_ENCODE = false>
__device__ __forceinline__ uint32_t half_px(const RescaleLookup &r, float y00, float y01, float y10, float y11,
                                            const Chroma &c, uint32_t alpha_word) {
  float x[12];  // r0..r3, g0..g3, b0..b3
  pixel_rgb(y00, c, x[0], x[4], x[8]);
  pixel_rgb(y01, c, x[1], x[5], x[9]);
  pixel_rgb(y10, c, x[2], x[6], x[10]);
  pixel_rgb(y11, c, x[3], x[7], x[11]);
  float lin[12];
  linearise12(r, x, lin);
  const float *lr = lin, *lg = lin + 4, *lb = lin + 8;
  // the average and the scaling into the encode table's domain are both exact powers of two
  const float sr = __fadd_rn(__fadd_rn(__fadd_rn(lr[0], lr[1]), lr[2]), lr[3]);
  const float sg = __fadd_rn(__fadd_rn(__fadd_rn(lg[0], lg[1]), lg[2]), lg[3]);
  const float sb = __fadd_rn(__fadd_rn(__fadd_rn(lb[0], lb[1]), lb[2]), lb[3]);
  if (UNIFORM_ENCODE) return pack_bgra(encode_byte_uniform(r, sr), encode_byte_uniform(r, sg), encode_byte_uniform(r, sb), alpha_word);
  return pack_bgra(encode_byte(r, sr), encode_byte(r, sg), encode_byte(r, sb), alpha_word);
}

// The two output pixels of a quad through the uniform encode table, SOFTWARE-PIPELINED over the LDS (round 4).  half_px
// issues a batch of six bucket reads and waits for it at once (s_waitcnt lgkmcnt(0) right behind the ds_read_b128s): for the
// whole LDS latency the wave has nothing to issue, and with 4 waves per SIMD (one 1024-lane workgroup per CU: the tables fill
// the LDS) the other three do not always cover it -- measured: 371 us per 16-frame launch when the gathers are conflict-free
// (flat content) = the VALU issue time, 413 us on uniform random bytes.  Here the four decode-side batches (a, b: pixel 0;
// c, d: pixel 1) and the two encode-side triples rotate: a batch is consumed while the next two are in flight, so every
// s_waitcnt leaves 6 to 12 reads outstanding (the LGKM counter holds 15).  Same arithmetic, same order of float operations
// per value as half_px: the bytes cannot differ (tests compare both kernels with the oracle).
#ifndef BT709_HALF_PIPELINE
#define BT709_HALF_PIPELINE 1  // 0: the unpipelined form (half_px twice), for A/B runs
#endif
struct Batch6 {
  u32x4 e[6];
};
__device__ __forceinline__ void batch_load(const RescaleLookup &r, const uint32_t *t, Batch6 &b) {
#pragma unroll
  for (int i = 0; i < 6; ++i) b.e[i] = *reinterpret_cast<LdsQuadPtr>((t[i] << r.dec_shift) + r.dec_off);
}
__device__ __forceinline__ void batch_use(const float *x, Batch6 &b, float *lin) {
  asm volatile("" : "+v"(b.e[0]), "+v"(b.e[1]), "+v"(b.e[2]), "+v"(b.e[3]), "+v"(b.e[4]), "+v"(b.e[5]));  // one wait per batch
#pragma unroll
  for (int i = 0; i < 6; ++i)
    lin[i] = __builtin_amdgcn_fmed3f(__uint_as_float(b.e[i].y), __uint_as_float(b.e[i].z), __fadd_rn(x[i], -__uint_as_float(b.e[i].x)));
}
struct Encode3 {
  u32x2 e[3];
  float s[3];
};
__device__ __forceinline__ void encode_load(const RescaleLookup &r, const float *lin, Encode3 &q) {
  const float *lr = lin, *lg = lin + 4, *lb = lin + 8;
  q.s[0] = __fadd_rn(__fadd_rn(__fadd_rn(lr[0], lr[1]), lr[2]), lr[3]);
  q.s[1] = __fadd_rn(__fadd_rn(__fadd_rn(lg[0], lg[1]), lg[2]), lg[3]);
  q.s[2] = __fadd_rn(__fadd_rn(__fadd_rn(lb[0], lb[1]), lb[2]), lb[3]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t t = __float_as_uint(__builtin_fmaf(q.s[k], r.sum_to_xs, 8388608.0f));  // as encode_byte_uniform
    q.e[k] = *reinterpret_cast<LdsPairPtr>((t << r.enc_shift) + r.enc_u_off);
  }
}
__device__ __forceinline__ uint32_t encode_use(Encode3 &q, uint32_t alpha_word) {
  asm volatile("" : "+v"(q.e[0]), "+v"(q.e[1]), "+v"(q.e[2]));
  uint32_t b[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b[k] = q.e[k].y + (q.s[k] >= __uint_as_float(q.e[k].x) ? 1u : 0u);
  return pack_bgra(b[0], b[1], b[2], alpha_word);
}

__device__ __forceinline__ u32x2 half_quad_pipelined(const RescaleLookup &r, uint32_t ya, uint32_t yb, uint32_t cw, uint32_t aw0,
                                                     uint32_t aw1) {
  const Chroma c0 = chroma_terms(byte_of(cw, 0), byte_of(cw, 1));
  const Chroma c1 = chroma_terms(byte_of(cw, 2), byte_of(cw, 3));
  float x0[12], x1[12];  // r0..r3, g0..g3, b0..b3 of each output pixel's 2x2 block
  pixel_rgb(byte_of(ya, 0), c0, x0[0], x0[4], x0[8]);
  pixel_rgb(byte_of(ya, 1), c0, x0[1], x0[5], x0[9]);
  pixel_rgb(byte_of(yb, 0), c0, x0[2], x0[6], x0[10]);
  pixel_rgb(byte_of(yb, 1), c0, x0[3], x0[7], x0[11]);
  uint32_t t0[12], t1[12];
  magic_index12(x0, t0, r.magic);
  Batch6 a, b, c, d;
  batch_load(r, t0, a);
  batch_load(r, t0 + 6, b);
  __builtin_amdgcn_sched_barrier(0);
  pixel_rgb(byte_of(ya, 2), c1, x1[0], x1[4], x1[8]);
  pixel_rgb(byte_of(ya, 3), c1, x1[1], x1[5], x1[9]);
  pixel_rgb(byte_of(yb, 2), c1, x1[2], x1[6], x1[10]);
  pixel_rgb(byte_of(yb, 3), c1, x1[3], x1[7], x1[11]);
  magic_index12(x1, t1, r.magic);
  float lin0[12], lin1[12];
  batch_use(x0, a, lin0);          // waits for a: b stays in flight
  batch_load(r, t1, c);
  __builtin_amdgcn_sched_barrier(0);
  batch_use(x0 + 6, b, lin0 + 6);  // c in flight
  batch_load(r, t1 + 6, d);
  Encode3 e0, e1;
  encode_load(r, lin0, e0);
  __builtin_amdgcn_sched_barrier(0);
  batch_use(x1, c, lin1);          // d, e0 in flight
  __builtin_amdgcn_sched_barrier(0);
  batch_use(x1 + 6, d, lin1 + 6);  // e0 in flight
  encode_load(r, lin1, e1);
  __builtin_amdgcn_sched_barrier(0);
  u32x2 v;
  v.x = encode_use(e0, aw0);       // e1 in flight
  v.y = encode_use(e1, aw1);
  return v;
}

// the two output pixels of a quad (4x2 source pixels); aw0 / aw1 = their alpha words
template <bool UNIFORM_ENCODE = false>
__device__ __forceinline__ u32x2 half_quad(const RescaleLookup &r, uint32_t ya, uint32_t yb, uint32_t cw,
                                           uint32_t aw0, uint32_t aw1) {
  const Chroma c0 = chroma_terms(byte_of(cw, 0), byte_of(cw, 1));
  const Chroma c1 = chroma_terms(byte_of(cw, 2), byte_of(cw, 3));
  u32x2 v;
  v.x = half_px<UNIFORM_ENCODE>(r, byte_of(ya, 0), byte_of(ya, 1), byte_of(yb, 0), byte_of(yb, 1), c0, aw0);
  v.y = half_px<UNIFORM_ENCODE>(r, byte_of(ya, 2), byte_of(ya, 3), byte_of(yb, 2), byte_of(yb, 3), c1, aw1);
  return v;
}

}  // namespace

// ---------------------------------------------------------------------------
// Exact 2:1, one workgroup per tile of an output row (small launches, any layout).
// WIDE: a lane owns quads = 4x2 source pixels = 2 output pixels (two dword luma loads, one
// dword CbCr load, one 8-byte store); grid = (tiles, H/2, frames) as in the 1:1 kernel.
// Preconditions: width % 4 == 0, planes/strides 4-byte aligned, output 8-byte aligned.
// !WIDE: one lane per output pixel, byte loads, any layout.
// ---------------------------------------------------------------------------
template <bool NT, bool WIDE, bool HAS_ALPHA>
__global__ void __launch_bounds__(kMaxBlockThreads)
decode_nv12_half(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const FramePlanes f = frame_planes(p, blockIdx.z);
  const uint32_t out_rows = p.height >> 1;
  const uint32_t orow_raw = blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y);  // wave-uniform
  const uint32_t orow = min(orow_raw, out_rows - 1);
  const uint8_t *y0 = f.y + static_cast<size_t>(2 * orow) * p.y_stride;
  const uint8_t *y1 = y0 + p.y_stride;
  const uint8_t *cc = f.cbcr + static_cast<size_t>(orow) * p.cbcr_stride;
  const uint8_t *a0 = HAS_ALPHA ? f.alpha + static_cast<size_t>(2 * orow) * p.alpha_stride : nullptr;
  const uint8_t *a1 = HAS_ALPHA ? a0 + p.alpha_stride : nullptr;
  uint8_t *o = f.out + static_cast<size_t>(orow) * p.out_stride;

  if (WIDE) {
    constexpr int UNROLL = kQuadsPerLane;
    const uint32_t quads = p.width >> 2;
    const uint32_t q0 = blockIdx.x * (blockDim.x * UNROLL) + threadIdx.x;
    uint32_t ya[UNROLL], yb[UNROLL], cw[UNROLL], aa[UNROLL], ab[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t q = min(q0 + u * blockDim.x, quads - 1);  // clamped load, predicated store (see 1:1 kernel)
      ya[u] = load32<NT>(y0 + 4 * q);
      yb[u] = load32<NT>(y1 + 4 * q);
      cw[u] = load32<NT>(cc + 4 * q);
      if (HAS_ALPHA) {
        aa[u] = load32<NT>(a0 + 4 * q);
        ab[u] = load32<NT>(a1 + 4 * q);
      }
    }
    const RescaleLookup r = stage_rescale_tables(lds_raw, p, 0, 0);  // after the tile's loads are in flight
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {  // see 1:1 kernel
      asm volatile("" : "+v"(ya[u]), "+v"(yb[u]), "+v"(cw[u]));
      if (HAS_ALPHA) asm volatile("" : "+v"(aa[u]), "+v"(ab[u]));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t q = q0 + u * blockDim.x;
      uint32_t aw0 = p.alpha_word, aw1 = p.alpha_word;
      if (HAS_ALPHA) {
        aw0 = half_alpha_arith(byte_of(aa[u], 0), byte_of(aa[u], 1), byte_of(ab[u], 0), byte_of(ab[u], 1));
        aw1 = half_alpha_arith(byte_of(aa[u], 2), byte_of(aa[u], 3), byte_of(ab[u], 2), byte_of(ab[u], 3));
      }
      const u32x2 v = half_quad(r, ya[u], yb[u], cw[u], aw0, aw1);
      if (q < quads && orow_raw < out_rows) store8<NT>(o + 8 * q, v);
    }
  } else {
    const RescaleLookup r = stage_rescale_tables(lds_raw, p, 0, 0);
    __syncthreads();
    const uint32_t out_w = p.width >> 1;
    for (uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x; ox < out_w && orow_raw < out_rows;
         ox += gridDim.x * blockDim.x) {
      const Chroma c = chroma_terms(byte_value(cc[2 * ox]), byte_value(cc[2 * ox + 1]));
      uint32_t aw = p.alpha_word;
      if (HAS_ALPHA)
        aw = half_alpha_arith(byte_value(a0[2 * ox]), byte_value(a0[2 * ox + 1]), byte_value(a1[2 * ox]),
                        byte_value(a1[2 * ox + 1]));
      reinterpret_cast<uint32_t *>(o)[ox] = half_px(r, byte_value(y0[2 * ox]), byte_value(y0[2 * ox + 1]),
                                                    byte_value(y1[2 * ox]), byte_value(y1[2 * ox + 1]), c, aw);
    }
  }
}

// ---------------------------------------------------------------------------
// Exact 2:1, CONFLICT-FREE form for large launches: same arithmetic, same bytes out.  On random
// content a bucket lookup from a single LDS copy of the table costs ~2.4x its conflict-free
// cycles.  Here the decode-side table sits in LDS in R = 16 interleaved copies and lane l reads
// copy l & 15: the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...:
// MI355X_MICROARCH.md, LDS) then hit 16 different 16-byte bank groups whatever their q, so each
// lookup costs its 4 LDS cycles and no more.  The encode table gets the copies that still fit (4
// for the default gamma: 131 + 24 KiB of the CU's 160).  One workgroup per CU can hold that, so
// workgroups are PERSISTENT: the tables are staged once per launch, then workgroup w walks tile
// rows w, w + G, w + 2G, ... (G = gridDim.x; at any moment the CUs work on neighbouring row
// pairs, i.e. the DRAM stream stays address-ordered).  A tile row = blockDim.x quads of one row
// pair; the cursor (tile, row pair, frame) advances by G decomposed on the host: no division in
// the loop.
// ---------------------------------------------------------------------------
namespace {

// encode side of the persistent kernel: the uniform non-power-of-two table (index = one fma); round 1's two-resolution
// table (convert, shift, add, min) is what the short-lived kernel uses and a lab variant here (tools/lab_variants.py)
constexpr bool kRepUniformEncode = true;
// (The luma terms Yn * My from a 256-entry LDS table -- one SDWA shift + ds_read_b32 instead of convert, fma,
// multiply, 8 fewer VALU instructions per output pixel in a VALU-issue-bound kernel -- measured 3 % SLOWER in the
// same call, profiles/r02_ab_half_luma_table.txt: at 60 % busy the LDS pipe has no room for four more conflicted
// gathers per pixel, and no LDS is left to replicate that table.  The commit before this comment holds the code.)
// (A form that runs the quad's two output pixels on VGPR pairs -- v_pk_fma/mul/add_f32, 25 % fewer
// instructions -- measured 4.7 % SLOWER in the same call, profiles/r02_ab_half_packed_f32.txt: packed f32 ops
// run at half rate, so the VALU cycles do not change, and the pairing costs scheduling freedom.  Commit e0a028e.)

struct TileCursor {
  uint32_t tx, rp, f;
};

struct QuadIn {
  uint32_t ya, yb, cw, aa, ab;  // aa / ab: the alpha plane's two rows (alpha decoders)
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

template <bool NT, bool HAS_ALPHA>
__device__ __forceinline__ QuadIn load_quad(const DecodeParams &p, const TileCursor &c, uint32_t quads) {
  const FramePlanes f = frame_planes(p, c.f);
  const uint8_t *y0 = f.y + static_cast<size_t>(2 * c.rp) * p.y_stride;
  const uint8_t *cc = f.cbcr + static_cast<size_t>(c.rp) * p.cbcr_stride;
  const uint32_t q = min(c.tx * blockDim.x + threadIdx.x, quads - 1);  // clamped: see the store
  QuadIn in;
  in.ya = load32<NT>(y0 + 4 * q);
  in.yb = load32<NT>(y0 + p.y_stride + 4 * q);
  in.cw = load32<NT>(cc + 4 * q);
  in.aa = in.ab = 0;
  if (HAS_ALPHA) {
    const uint8_t *a0 = f.alpha + static_cast<size_t>(2 * c.rp) * p.alpha_stride;
    in.aa = load32<NT>(a0 + 4 * q);
    in.ab = load32<NT>(a0 + p.alpha_stride + 4 * q);
  }
  return in;
}

}  // namespace

template <bool NT, int U, bool HAS_ALPHA>
__global__ void __launch_bounds__(kRepBlockThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))  // one 16-wave workgroup per CU: 128 VGPRs are free
decode_nv12_half_rep(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
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
  // row -- same loads, same result, same store -- so every step is exactly 3U (5U with an alpha plane) loads and U stores
  // and the in-order vmcnt hipcc derives never has to cover a shorter path.
  QuadIn in[U];
  {
    const TileCursor first = pre;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool have = t + u * G < p.tile_rows;
      const TileCursor c = {have ? pre.tx : first.tx, have ? pre.rp : first.rp, have ? pre.f : first.f};
      in[u] = load_quad<NT, HAS_ALPHA>(p, c, quads);
      advance(pre, p, row_pairs);
    }
  }

  const RescaleLookup r = stage_rescale_tables<kRepUniformEncode>(lds_raw, p, p.rep_dec_log2, p.rep_enc_log2);
  __syncthreads();

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
        nx[u] = load_quad<NT, HAS_ALPHA>(p, c, quads);
        advance(pre, p, row_pairs);
      }
    }
    const TileCursor first = cur;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool have = t + u * G < p.tile_rows;
      const TileCursor c = {have ? cur.tx : first.tx, have ? cur.rp : first.rp, have ? cur.f : first.f};
      uint32_t aw0 = p.alpha_word, aw1 = p.alpha_word;
      if (HAS_ALPHA) {
        aw0 = half_alpha_arith(byte_of(in[u].aa, 0), byte_of(in[u].aa, 1), byte_of(in[u].ab, 0), byte_of(in[u].ab, 1));
        aw1 = half_alpha_arith(byte_of(in[u].aa, 2), byte_of(in[u].aa, 3), byte_of(in[u].ab, 2), byte_of(in[u].ab, 3));
      }
      const u32x2 v = BT709_HALF_PIPELINE ? half_quad_pipelined(r, in[u].ya, in[u].yb, in[u].cw, aw0, aw1)
                                          : half_quad<kRepUniformEncode>(r, in[u].ya, in[u].yb, in[u].cw, aw0, aw1);
      const FramePlanes f = frame_planes(p, c.f);
      uint8_t *o = f.out + static_cast<size_t>(c.rp) * p.out_stride;
      // lanes past the row's end loaded the last quad (clamp), hold its result and store it again
      const uint32_t q = min(c.tx * blockDim.x + threadIdx.x, quads - 1);
      store8<NT>(o + 8 * q, v);
      advance(cur, p, row_pairs);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) in[u] = nx[u];  // hipcc waits here for the loads issued at the top (not for the stores)
  }
}

// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------
#ifndef BT709_REP_STEP
#define BT709_REP_STEP 2  // tile rows per step of the persistent kernel (loads run one step ahead)
#endif

const char *launch_decode_half(const DecodeParams &p, int frames, bool wide, bool has_alpha, bool nontemporal,
                               uint32_t grid_x, uint32_t block_threads, hipStream_t stream) {
  const uint32_t by = wide ? quads_rows_per_block(block_threads, grid_x) : 1;
  const dim3 grid(grid_x, (p.height / 2 + by - 1) / by, static_cast<uint32_t>(frames));
  const dim3 block(block_threads, by, 1);
  const size_t lds = static_cast<size_t>(p.table_linear_bytes) + p.table_encode_bytes;
  if (has_alpha) {
    if (wide) hipLaunchKernelGGL((decode_nv12_half<true, true, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((decode_nv12_half<false, false, true>), grid, block, lds, stream, p);
    return wide ? "decode_nv12_half<wide,alpha>" : "decode_nv12_half<narrow,alpha>";
  }
  if (wide) {
    if (nontemporal) hipLaunchKernelGGL((decode_nv12_half<true, true, false>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((decode_nv12_half<false, true, false>), grid, block, lds, stream, p);
    return "decode_nv12_half<wide>";
  }
  hipLaunchKernelGGL((decode_nv12_half<false, false, false>), grid, block, lds, stream, p);
  return "decode_nv12_half<narrow>";
}

const char *launch_decode_half_rep(const DecodeParams &p_in, int frames, bool has_alpha, bool nontemporal, uint32_t workgroups,
                                   uint32_t lds_budget, hipStream_t stream) {
  DecodeParams p = p_in;
  const uint64_t kRepLdsBytes = (lds_budget < 16384u ? 16384u : (lds_budget > bt709::kRepLdsBytes ? bt709::kRepLdsBytes : lds_budget));
  // copies: as many as fit the CU's LDS, decode side first (12 of the 15 lookups per output pixel)
  const uint64_t enc_bytes = kRepUniformEncode ? p.table_encode_u_bytes : p.table_encode_bytes;
  uint32_t r1 = 4, r2 = 0;
  while (r1 > 0 && (static_cast<uint64_t>(p.table_linear_bytes) << r1) + enc_bytes > kRepLdsBytes) --r1;
  if ((static_cast<uint64_t>(p.table_linear_bytes) << r1) + enc_bytes > kRepLdsBytes) return nullptr;
  while (r2 < 5 && (static_cast<uint64_t>(p.table_linear_bytes) << r1) + (enc_bytes << (r2 + 1)) <= kRepLdsBytes) ++r2;
  p.rep_dec_log2 = r1;
  p.rep_enc_log2 = r2;
  const uint32_t quads = p.width / 4, row_pairs = p.height / 2;
  p.tiles_x = (quads + kRepBlockThreads - 1) / kRepBlockThreads;
  const uint32_t threads = ((quads + p.tiles_x - 1) / p.tiles_x + 63) / 64 * 64;
  const uint64_t total = static_cast<uint64_t>(p.tiles_x) * row_pairs * static_cast<uint32_t>(frames);
  if (total == 0 || total > 0x7fffffffu) return nullptr;
  p.tile_rows = static_cast<uint32_t>(total);
  if (workgroups > p.tile_rows) workgroups = p.tile_rows;
  if (workgroups == 0) workgroups = 1;
  p.cursor_tx = workgroups % p.tiles_x;
  p.cursor_rp = (workgroups / p.tiles_x) % row_pairs;
  p.cursor_f = (workgroups / p.tiles_x) / row_pairs;
  const size_t lds = (static_cast<size_t>(p.table_linear_bytes) << r1) + (static_cast<size_t>(enc_bytes) << r2);
  if (has_alpha) {
    if (nontemporal) hipLaunchKernelGGL((decode_nv12_half_rep<true, BT709_REP_STEP, true>), dim3(workgroups), dim3(threads), lds, stream, p);
    else hipLaunchKernelGGL((decode_nv12_half_rep<false, BT709_REP_STEP, true>), dim3(workgroups), dim3(threads), lds, stream, p);
    return "decode_nv12_half_rep<alpha>";
  }
  if (nontemporal)
    hipLaunchKernelGGL((decode_nv12_half_rep<true, BT709_REP_STEP, false>), dim3(workgroups), dim3(threads), lds, stream, p);
  else
    hipLaunchKernelGGL((decode_nv12_half_rep<false, BT709_REP_STEP, false>), dim3(workgroups), dim3(threads), lds, stream, p);
  return "decode_nv12_half_rep";
}

hipError_t prepare_rescale_kernels() {
  const int cap = static_cast<int>(kRepLdsBytes);  // gfx950: 160 KiB LDS per workgroup
  const void *fns[] = {
      reinterpret_cast<const void *>(&decode_nv12_half<true, true, false>),
      reinterpret_cast<const void *>(&decode_nv12_half<false, true, false>),
      reinterpret_cast<const void *>(&decode_nv12_half<false, false, false>),
      reinterpret_cast<const void *>(&decode_nv12_half<true, true, true>),
      reinterpret_cast<const void *>(&decode_nv12_half<false, false, true>),
      reinterpret_cast<const void *>(&decode_nv12_half_rep<true, BT709_REP_STEP, false>),
      reinterpret_cast<const void *>(&decode_nv12_half_rep<false, BT709_REP_STEP, false>),
      reinterpret_cast<const void *>(&decode_nv12_half_rep<true, BT709_REP_STEP, true>),
      reinterpret_cast<const void *>(&decode_nv12_half_rep<false, BT709_REP_STEP, true>),
  };
  for (const void *fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return prepare_scaled_kernels();
}

}  // namespace bt709
