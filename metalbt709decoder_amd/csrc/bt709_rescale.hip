// CDNA4 (gfx950) kernels of the FUSED decode + rescale: what the reference does in two Metal
// passes -- BT709ToLinearSRGBKernel into an sRGB8 intermediate (Renderer/AAPLShaders.metal:
// 336-407), then MetalScaleRenderContext -renderScaled: / samplingShader (73-85) when the view
// is smaller than the frame -- is ONE kernel here: the 132.7 MB 8K intermediate is never written.
//
// Two-pass-equivalent arithmetic ("rescale"; the reference has no CPU twin of pass 2,
// so parity is against the oracle's restatement of this definition, itself pinned to goldens composed
// of the reference's own inlines, tests/golden/pass2.json): each source pixel is
// decoded to its 8-bit sRGB value and linearised as the sRGB8 sampler would -- the decode-side
// table returns that linear float directly, {edge, lin(base), lin(base + 1)} in one 16-byte
// bucket (transfer_tables.h TransferBucketLinear) -- the taps are combined in linear light, and the result is sRGB-encoded and quantised
// through the LINEAR-mode composite, held as a log-bucket table (transfer_tables.h TransferTable::buckets_log: 645
// buckets, index by one fma and one shift; rounds 1-5a: a two-resolution table, three instructions more per lookup).
// The persistent 2:1 kernel keeps a uniform table with a non-power-of-two bucket count (index by ONE fma).
//
// The kernels are bound by VALU issue slots and by the LDS pipe together (12 decode-side + 3 encode-side lookups
// per output pixel; DESIGN.md 6.0 has the counters): per decode-side lookup a saturating add, the magic add, the
// address, a subtract and a median-of-three select.  The alpha channel of an alpha decoder is pure arithmetic.
//
//   decode_nv12_half       exact 2:1, one short-lived workgroup per tile of an output row (small launches, any layout)
//   decode_nv12_half_rep   exact 2:1, persistent workgroups, bank-conflict-free LDS tables (with and without alpha)
//   decode_nv12_scaled     any output size, bilinear taps, one lane per output column walking strips of rows
//   render_scaled          pass 2 alone from an 8-bit or RGBA16Float intermediate
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "bt709_device.h"

namespace bt709 {
namespace {

// LDS image of the two tables and the constants of a lookup.  Entry q of copy c of the decode
// side sits at byte (q * R + c) * 16 (R = 2^r1 copies), of the encode side at (q * R2 + c) * 8
// behind it; a lane reads copy lane & (R - 1).
struct RescaleLookup {
  float magic;         // 2^23 / N: bits(x + magic), rounded toward zero, = bits(magic) + floor(x N)
  uint32_t dec_shift;  // log2(16 R)
  uint32_t dec_off;    // LDS address + lane's copy offset - (bits(magic) << dec_shift)
  uint32_t enc_shift;  // log2(8 R2)
  uint32_t enc_off;    // LDS address of the encode table + lane's copy offset
  // log-bucket encode table (every kernel but the persistent 2:1 one): bucket of a value a in the kernel's own domain =
  // (bits(fma(a, quarter_unscale, enc_add)) >> 16) - first; a * quarter_unscale is the mean in [0, 1], exact (a power of two)
  float enc_add;
  uint32_t enc_log_off;  // enc_off - (first << enc_shift)
  // uniform encode table (persistent kernel): v = sum * quarter_unscale is the mean itself, xs = v * enc_n
  float quarter_unscale;  // 0.25 * 2^40
  float enc_n;
  float sum_to_xs;        // quarter_unscale * enc_n
  uint32_t enc_u_off;     // LDS address of the table + lane's copy offset - (bits(2^23) << enc_shift)
};

// Stages both tables in 2^r1 / 2^r2 interleaved copies (0 / 0: plain) and returns the lookup
// constants of this lane.  The caller synchronises.
// sum_log2 (uniform encode table only): the value handed to encode_byte_uniform is 2^sum_log2 times the
// mean -- 2 for the four-tap sum of the exact 2:1 kernel, 0 for the weighted sum of the any-ratio one.
template <bool UNIFORM_ENCODE = false>
__device__ __forceinline__ RescaleLookup stage_rescale_tables(unsigned char *lds_raw, const DecodeParams &p, uint32_t r1,
                                                              uint32_t r2, uint32_t sum_log2 = 2) {
  const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
  u32x4 *d = reinterpret_cast<u32x4 *>(lds_raw);
  const u32x4 *src = reinterpret_cast<const u32x4 *>(p.table_linear);
  const uint32_t n = (p.table_linear_bytes / 16) << r1;
  stage_batched(d, n, tid, nthreads, [&](uint32_t i) { return src[i >> r1]; });
  const uint32_t dec_bytes = p.table_linear_bytes << r1;
  u32x2 *d2 = reinterpret_cast<u32x2 *>(lds_raw + dec_bytes);
  const u32x2 *src2 = reinterpret_cast<const u32x2 *>(UNIFORM_ENCODE ? p.table_encode_u : p.table_encode);
  const uint32_t n2 = ((UNIFORM_ENCODE ? p.table_encode_u_bytes : p.table_encode_bytes) / 8) << r2;
  {
    // edges move into the domain of the value the kernel compares -- the taps' sum, or their weighted sum: edge * 2^sum_log2 *
    // 2^-40 (a power of two: exact; +inf stays +inf)
    const float to_sum = __uint_as_float(static_cast<uint32_t>(127 + sum_log2 + kLinearScaleLog2) << 23);
    stage_batched(d2, n2, tid, nthreads, [&](uint32_t i) {
      u32x2 e = src2[i >> r2];
      e.x = __float_as_uint(__fmul_rn(__uint_as_float(e.x), to_sum));
      return e;
    });
  }

  const uint32_t base = lds_address(lds_raw);
  RescaleLookup r;
  r.magic = p.unit_magic;
  r.dec_shift = 4u + r1;
  r.dec_off = base + (tid & ((1u << r1) - 1u)) * 16u - (__float_as_uint(r.magic) << r.dec_shift);
  r.enc_shift = 3u + r2;
  r.enc_off = base + dec_bytes + (tid & ((1u << r2) - 1u)) * 8u;
  r.enc_add = p.encode_log_add;
  r.enc_log_off = r.enc_off - (p.encode_log_first << r.enc_shift);
  asm volatile("" : "+v"(r.enc_log_off));  // ONE addend of the v_lshl_add
  const float unscale = __uint_as_float(static_cast<uint32_t>(127 - kLinearScaleLog2) << 23);  // 2^40
  r.quarter_unscale = __fmul_rn(__uint_as_float((127u - sum_log2) << 23), unscale);  // 2^-sum_log2 * 2^40
  r.enc_n = p.encode_u_n;
  r.enc_u_off = r.enc_off - (0x4b000000u << r.enc_shift);
  asm volatile("" : "+v"(r.enc_u_off));  // keep it ONE addend of the v_lshl_add (hipcc otherwise subtracts bits(2^23) per lookup)
  r.sum_to_xs = __fmul_rn(r.quarter_unscale, r.enc_n);  // exact: quarter_unscale is a power of two
  return r;
}

// sRGB byte of the linear-light SUM s of the four taps (times 2^-40) through the uniform table
// (transfer_tables.h UniformTable).  The mean v = s * quarter_unscale is never formed: the index comes from
// s * (quarter_unscale * n) -- the same float as v * n, the factor being a power of two times n -- and the
// bucket's edge is compared in the sum's own domain (edges pre-divided by quarter_unscale at staging).
__device__ __forceinline__ uint32_t encode_byte_uniform(const RescaleLookup &r, float s) {
  // ONE fma: bits(2^23) + round(v n + 2^23 as a real number).  An index function only has to be monotone and the
  // same on the host (transfer_tables.cpp uniform_index files the thresholds under it); it is not reference arithmetic.
  const uint32_t t = __float_as_uint(__builtin_fmaf(s, r.sum_to_xs, 8388608.0f));
  const u32x2 e = *reinterpret_cast<LdsPairPtr>((t << r.enc_shift) + r.enc_u_off);
  return e.y + (s >= __uint_as_float(e.x) ? 1u : 0u);
}

// sRGB byte of a linear-light value a in the kernel's own domain (a sum of taps times 2^-40, a weighted sum, a unit-range mean:
// RescaleLookup::quarter_unscale takes it to the mean v in [0, 1]) through the LOG-bucket table: bucket = (bits(v + add) >> 16) -
// first -- ONE fma (the product inside is exact: a power of two), one shift -- and the bucket's edge compared in a's own domain
// (edges pre-divided at staging).  Rounds 2-5a used a two-resolution table here: multiply, convert, shift, add, min (three
// instructions more per lookup).
__device__ __forceinline__ uint32_t encode_byte(const RescaleLookup &r, float a) {
  const uint32_t t = __float_as_uint(__builtin_fmaf(a, r.quarter_unscale, r.enc_add)) >> 16;
  const u32x2 e = *reinterpret_cast<LdsPairPtr>((t << r.enc_shift) + r.enc_log_off);
  return e.y + (a >= __uint_as_float(e.x) ? 1u : 0u);
}

// linear-light values (times 2^-40) of 12 saturated channel values: kLinBatch buckets in flight per wait
#ifndef BT709_LIN_BATCH
#define BT709_LIN_BATCH 6
#endif
constexpr int kLinBatch = BT709_LIN_BATCH;  // 6 or 12 (12: one wait per pixel, 48 VGPRs of buckets in flight)
__device__ __forceinline__ void linearise12(const RescaleLookup &r, const float *x, float *lin) {
  uint32_t t[12];
  magic_index12(x, t, r.magic);
#pragma unroll
  for (int h = 0; h < 12 / kLinBatch; ++h) {
    u32x4 e[kLinBatch];  // {edge, lin(base), lin(base + 1), base}: whole vectors keep the read a ds_read_b128
#pragma unroll
    for (int i = 0; i < kLinBatch; ++i) e[i] = *reinterpret_cast<LdsQuadPtr>((t[kLinBatch * h + i] << r.dec_shift) + r.dec_off);
    if (kLinBatch == 6) asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]));  // one wait per batch
    else {
      asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6 % kLinBatch]), "+v"(e[7 % kLinBatch]),
                        "+v"(e[8 % kLinBatch]), "+v"(e[9 % kLinBatch]), "+v"(e[10 % kLinBatch]), "+v"(e[11 % kLinBatch]));
    }
#pragma unroll
    for (int i = 0; i < kLinBatch; ++i)  // transfer_tables.h TransferBucketLinear: below / above by sub + med3
      lin[kLinBatch * h + i] = __builtin_amdgcn_fmed3f(__uint_as_float(e[i].y), __uint_as_float(e[i].z),
                                                       __fadd_rn(x[kLinBatch * h + i], -__uint_as_float(e[i].x)));
  }
}

// one batch of six (the two horizontal taps of a source row in decode_nv12_scaled)
__device__ __forceinline__ void linearise6(const RescaleLookup &r, const float *x, float *lin) {
  const float xp[8] = {x[0], x[1], x[2], x[3], x[4], x[5], 0.0f, 0.0f};
  uint32_t t[8];
  u32x4 e[6];
  magic_index4(xp, t, r.magic);
  magic_index4(xp + 4, t + 4, r.magic);
#pragma unroll
  for (int i = 0; i < 6; ++i) e[i] = *reinterpret_cast<LdsQuadPtr>((t[i] << r.dec_shift) + r.dec_off);
  asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]));  // one wait
#pragma unroll
  for (int i = 0; i < 6; ++i)
    lin[i] = __builtin_amdgcn_fmed3f(__uint_as_float(e[i].y), __uint_as_float(e[i].z), __fadd_rn(x[i], -__uint_as_float(e[i].x)));
}

// one pixel (decode_nv12_scaled's wave-decodes-once form); x[3] is padding
__device__ __forceinline__ void linearise3(const RescaleLookup &r, const float *x, float *lin) {
  uint32_t t[4];
  u32x4 e[3];
  magic_index4(x, t, r.magic);
#pragma unroll
  for (int i = 0; i < 3; ++i) e[i] = *reinterpret_cast<LdsQuadPtr>((t[i] << r.dec_shift) + r.dec_off);
  asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]));  // one wait
#pragma unroll
  for (int i = 0; i < 3; ++i)
    lin[i] = __builtin_amdgcn_fmed3f(__uint_as_float(e[i].y), __uint_as_float(e[i].z), __fadd_rn(x[i], -__uint_as_float(e[i].x)));
}

// Alpha decoders only.  Pass 2 reads the alpha channel of the 8-bit intermediate as a plain unorm
// (AAPLShaders.metal:411-438 writes it, the sampler of MetalScaleRenderContext.m:55-105 filters it): each tap is
// byteNorm(decoded alpha byte), the result round(255 v).  No tables: an alpha decoder runs the sRGB mode, whose
// composite is the plain quantiser, so the decoded alpha byte of a sample is (int)round(x * 255.0f) of its
// saturated luma term x (BT709.h:881-883) and byteNorm is byte * (1/255f) (sRGB.h:32-36).  The arithmetic lives in
// bt709_quantise.h, compiled for the host too: tests/test_quantiser_exact.py replays it against the oracle over
// every input it can meet (256 sample codes, 256^4 ordered tap tuples of the 2:1 filter, every float in [0, 1] for
// the any-ratio filter's result).  8 VALU instructions per sample, 7 per output pixel of the 2:1 filter.
// (Round 2's first form went through a byteNorm bucket table and the quantiser table in LDS:
// 151 against 217 Gpixel/s on 8K -> 4K with alpha, and the tables kept alpha out of the persistent kernel.)
__device__ __forceinline__ float alpha_norm_arith(float abyte) {
  return alpha_norm_of_unit(alpha_value(abyte));
}

// (alpha byte << 24) of a FILTERED alpha value: round(255 * saturate(v)).  The argument is a weighted sum (any
// ratio) or comes out of an RGBA16Float intermediate: not enumerable, so the quantiser is the one that is exact
// for every float (round 2 used the three-instruction form here: one LSB off at v * 255 = 0.49999997).
__device__ __forceinline__ uint32_t alpha_word_of(float v) {
  return quantise_exact(add_sat(v, 0.0f)) << 24;
}

// exact 2:1: the four taps of a block, each one of 256 values -- all 256^4 ordered tuples are replayed on the host
__device__ __forceinline__ uint32_t half_alpha_arith(float a00, float a01, float a10, float a11) {
  return half_alpha_sum_to_byte(alpha_norm_arith(a00), alpha_norm_arith(a01), alpha_norm_arith(a10), alpha_norm_arith(a11)) << 24;
}

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
      __builtin_amdgcn_raw_buffer_store_b32(pack_bgra(R, G, B, aw), ro, ox * 4u, oy * p.out_stride, 0);
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
    __builtin_amdgcn_raw_buffer_store_b32(pack_bgra(R, G, B, A), rout, ox * 4u, oy * p.out_stride, 0);
  };
  const uint32_t last = oy1 - 1;
  RowTaps rta = row_taps(oy0), rtb;
  Fetched a0 = fetch_row(rta.ys[0]), a1 = fetch_row(rta.ys[1]), b0, b1;
  for (uint32_t oy = oy0; oy < oy1; oy += 2) {
    rtb = row_taps(min(oy + 1, last));
    b0 = fetch_row(rtb.ys[0]);
    b1 = fetch_row(rtb.ys[1]);
    output_row(oy, rta, a0, a1);
    if (oy + 1 >= oy1) {
      landed(b0);
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
  rows = rows < 1 ? 1 : (rows > BT709_SCALED_MAX_ROWS ? BT709_SCALED_MAX_ROWS : rows);
  p.scaled_rows = rows;
  const uint32_t strips = (p.out_height + rows - 1) / rows;
  const uint32_t strip_groups = (strips + kScaledStrips - 1) / kScaledStrips;
  const size_t lds = (static_cast<size_t>(p.table_linear_bytes) << kScaledDecCopiesLog2) + ((kScaledUniform ? p.table_encode_u_bytes : p.table_encode_bytes) << kScaledEncCopiesLog2);
  const dim3 block(kBlockThreads, kScaledStrips);
  const bool persistent = taps == TAPS_ONCE ? BT709_SCALED_ONCE_PERSISTENT != 0 : taps != TAPS_SHARED;
  dim3 grid(cols, strip_groups, static_cast<uint32_t>(frames));
  const void *fn = nullptr;
#define BT709_PICK_SCALED(T, P)                                                                                        \
  fn = has_alpha ? reinterpret_cast<const void *>(&decode_nv12_scaled<T, true, P>) : reinterpret_cast<const void *>(&decode_nv12_scaled<T, false, P>)
  if (taps == TAPS_ONCE) BT709_PICK_SCALED(TAPS_ONCE, BT709_SCALED_ONCE_PERSISTENT != 0);
  else if (taps == TAPS_SHARED) BT709_PICK_SCALED(TAPS_SHARED, false);
  else if (taps == TAPS_WIDE) BT709_PICK_SCALED(TAPS_WIDE, true);
  else if (taps == TAPS_PAIRS) BT709_PICK_SCALED(TAPS_PAIRS, true);
  else BT709_PICK_SCALED(TAPS_BYTES, true);
#undef BT709_PICK_SCALED
  if (persistent) {
    const uint64_t items = static_cast<uint64_t>(cols) * strip_groups * static_cast<uint32_t>(frames);
    if (items > 0x7fffffffull) return nullptr;
    p.tiles_x = cols;
    p.tile_rows = static_cast<uint32_t>(items);
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
    const uint64_t resident = static_cast<uint64_t>(per_cu) * cus;
    grid = dim3(static_cast<uint32_t>(items < resident ? items : resident), 1, 1);
  }
  void *args[] = {&p};
  (void)hipLaunchKernel(fn, grid, block, args, lds, stream);  // a failure is picked up by the caller's hipGetLastError
  return has_alpha ? "decode_nv12_scaled<alpha>" : "decode_nv12_scaled";
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
