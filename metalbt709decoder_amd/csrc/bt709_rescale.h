// Device-side building blocks shared by the two files of the FUSED decode + rescale (bt709_rescale_half.hip: the exact 2:1
// kernels; bt709_rescale_scaled.hip: any output size and pass 2 alone): the LDS image of the two lookup tables and the lookups.
//
// Two-pass-equivalent arithmetic (the reference has no CPU twin of pass 2, so parity is against the oracle's
// restatement of this definition, itself pinned to goldens composed of the reference's own inlines, tests/golden/pass2.json):
// each source pixel is decoded to its 8-bit sRGB value and linearised as the sRGB8 sampler would -- the decode-side table
// returns that linear float directly, {edge, lin(base), lin(base + 1)} in one 16-byte bucket (transfer_tables.h
// TransferBucketLinear) -- the taps are combined in linear light, and the result is sRGB-encoded and quantised through the
// LINEAR-mode composite, held as a log-bucket table (transfer_tables.h TransferTable::buckets_log: 645 buckets, index by one
// fma and one shift).  The persistent 2:1 kernel keeps a uniform table with a non-power-of-two bucket count (index by ONE fma).
#pragma once
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

}  // namespace

hipError_t prepare_scaled_kernels();  // bt709_rescale_scaled.hip; called by prepare_rescale_kernels (bt709_rescale_half.hip)

}  // namespace bt709
