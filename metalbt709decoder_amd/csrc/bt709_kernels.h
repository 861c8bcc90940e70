// Kernel parameter block and launchers shared by bt709_kernels.hip and the C-ABI shim.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "transfer_tables.h"

namespace bt709 {

constexpr int kBlockThreads = 256;     // general-path workgroup: 4 waves of 64
// (tools/decode_lab overrides these two with -D to A/B other tile shapes)
#ifndef BT709_MAX_BLOCK_THREADS
#define BT709_MAX_BLOCK_THREADS 512
#endif
#ifndef BT709_QUADS_PER_LANE
#define BT709_QUADS_PER_LANE 2
#endif
constexpr int kMaxBlockThreads = BT709_MAX_BLOCK_THREADS;  // fast-path workgroup is sized per frame width, up to 8 waves
constexpr int kQuadsPerLane = BT709_QUADS_PER_LANE;        // 4x2-pixel quads a fast-path lane owns per row pair
constexpr int kMaxBatch = 32;          // == BT709HIP_MAX_BATCH: frames in the kernarg table
// XCD-aware work map (bt709_kernels.hip decode_nv12_quads): used for launches of a multiple of 8 frames from this many on.
// Measured (round 3, same call, plain vs banded): decode 32 frames 0.756 / 0.741-0.761, 64 0.741 / 0.745-0.760, 128 0.72 / 0.77,
// 256 0.70 / 0.76-0.81; encoder 32 pictures 0.70 / 0.67, 256 0.69 / 0.73: a band needs ~8 frames to pay.
#ifndef BT709_XCD_BAND_MIN_FRAMES
#define BT709_XCD_BAND_MIN_FRAMES 64
#endif
constexpr int kXcdBandMinFrames = BT709_XCD_BAND_MIN_FRAMES;
constexpr int kMaxUniformBatch = 65535;  // evenly spaced frames per launch (grid.z limit)

enum KernelVariant : int {
  kVariantQuads = 0,   // aligned fast path, 4x2 pixels per lane
  kVariantBlocks = 1,  // general path, 2x2 pixels per lane
};

struct FramePlanes {
  const uint8_t *y;
  const uint8_t *cbcr;
  const uint8_t *alpha;  // nullptr unless the decoder has an alpha channel
  uint8_t *out;
};

// Passed by value in the kernarg segment (32 frames x 32 B + ~150 B).
struct DecodeParams {
  FramePlanes frames[kMaxBatch];
  // decode kernels: TransferBucket[N + 1], edges in x units (transfer_tables.h buckets_unit)
  const void *table_unit;
  // rescale kernels: TransferBucketLinear[N + 1] (edges in x units) and the sRGB-encode table in LOG-bucket form
  // (transfer_tables.h TransferTable::buckets_log of the LINEAR composite): bucket of a mean v = (bits(v + encode_log_add) >> 16) - encode_log_first
  const void *table_linear;
  const void *table_encode;
  // persistent 2:1 kernel, encode side: the same composite as a uniform table with a non-power-of-two
  // bucket count (transfer_tables.h UniformTable): index by two multiplies and an add
  const void *table_encode_u;
  uint32_t table_encode_u_bytes;
  float encode_u_n;
  uint32_t table_unit_bytes;
  uint32_t table_linear_bytes;
  uint32_t table_encode_bytes;
  float encode_log_add;
  uint32_t encode_log_first;
  float unit_magic;    // 2^23 / N: floats in [M, 2M) have ulp 1/N (bt709_device.h magic_index12): table_linear's index
  // table_unit's index (the 1:1 kernels): q = (bits(x + unit1_magic) >> unit1_shift) - unit1_first.  Uniform form: the same M,
  // shift 0, first = bits(M); log-bucket form (transfer_tables.h TransferTable::buckets_log): log_add, 16, log_first.
  float unit1_magic;
  uint32_t unit1_first, unit1_shift;
  uint32_t width;      // luma (source) dimensions
  uint32_t height;
  uint32_t y_stride;
  uint32_t cbcr_stride;
  uint32_t alpha_stride;
  uint32_t out_stride;
  uint32_t alpha_word;  // alpha_fill << 24
  // decode_nv12_scaled only: output size and source-per-output-pixel ratios W/OW, H/OH (float)
  uint32_t out_width, out_height;
  float scale_x, scale_y;
  uint32_t scaled_rows;  // output rows of a strip (filled by launch_decode_scaled)
  // uniform != 0: frame i = frames[0] + i * step_* (bytes); lets one launch cover any number of frames
  uint32_t uniform;
  int64_t step_y, step_cbcr, step_alpha, step_out;
  // decode_nv12_half_rep only (persistent workgroups, replicated LDS tables): table_linear is held
  // in 2^rep_dec_log2 interleaved copies (<= 16: one per lane of a ds_read_b128 lane group),
  // table_encode in 2^rep_enc_log2; a tile row is one tile of one row pair of one frame, tile_rows
  // of them in the launch, walked gridDim.x at a time; cursor_* = gridDim.x decomposed into
  // (tiles, row pairs, frames).  Filled by launch_decode_half_rep.
  uint32_t rep_dec_log2, rep_enc_log2;
  uint32_t tiles_x, tile_rows;
  uint32_t cursor_tx, cursor_rp, cursor_f;
  // XCD-aware work map of the short-lived kernels (filled by the launchers): grid.x = 8 x tiles, x & 7 = the workgroup's
  // place in the round-robin over the XCDs, which owns frames [(x & 7) * frames_per_band, ...) of the launch
  uint32_t xcd_bands, frames_per_band;
};

// Pass 1 into an RGBA16Float target (bt709_rgba16f.hip): the threshold table of transfer_tables.h
// HalfTable and the constants of the candidate.  Travels beside DecodeParams (frames, pitches).
struct HalfParams {
  const void *table;     // float T[]: T[i] = smallest x with H(x) >= h_min + i, then the candidate entries; nullptr: no curve (LINEAR)
  uint32_t table_bytes;  // 0 without a table; else thresholds + candidates
  uint32_t cand_offset;  // byte offset of the candidate entries {intercept, slope} (transfer_tables.h HalfTable::cand) in `table` = bytes of the thresholds
  uint32_t h_min, h_max; // codes the table covers
  float split, low_scale;
  float index_scale;     // HalfTable::index_scale: the split point lands on the bucket boundary 2^-4
  uint32_t row_pairs_per_block;  // filled by the launcher
  uint32_t wide_store;           // 16-byte stores (target 16-byte aligned), else 8-byte
};
// LDS plan of decode_nv12_rgba16f: the thresholds from byte 0, the candidate entries from this FIXED byte on -- so that the
// address of a bucket's entry is its binary16 bits plus a compile-time constant (the ds_read's immediate offset).  Above the
// largest threshold image (sRGB: 8 602 floats), and low enough that thresholds + candidates stay under 40 KiB: four
// workgroups per CU.
constexpr uint32_t kHalfCandLds = 34560;
const char *launch_decode_rgba16f(const DecodeParams &p, const HalfParams &hp, int frames, bool has_alpha,
                                  uint32_t in_align, uint32_t out_align, uint32_t compute_units, bool xcd_bands, hipStream_t stream);
hipError_t prepare_rgba16f_kernels();  // bt709_rgba16f.hip

// Pass 2 alone (bt709_rescale_scaled.hip render_scaled): an intermediate surface -> a BGRA8 sRGB surface of any size.
struct RenderParams {
  const uint8_t *in;   // BGRA8 sRGB words or RGBA16Float texels
  uint8_t *out;        // BGRA8 sRGB
  uint32_t in_stride, out_stride;
  uint32_t width, height, out_width, out_height;
  float scale_x, scale_y;
  uint32_t rows;       // output rows a workgroup walks (filled by the launcher)
  // tables (device): log-bucket sRGB-encode buckets and lin[256] = sRGB_nonLinearNormToLinear(byteNorm(b))
  // (the alpha channel is filtered and quantised in arithmetic)
  const void *table_encode, *table_lin;
  uint32_t table_encode_bytes;
  float encode_log_add;        // table_encode is the log-bucket form (as DecodeParams)
  uint32_t encode_log_first;
  int64_t in_step, out_step;  // batched launches: surface i of the launch = surface 0 + i * step (evenly spaced, as in a ring)
};
// grid = (column tiles, strips of `rows` output rows, frames)
const char *launch_render_scaled(const RenderParams &p, int frames, bool in_rgba16f, uint32_t compute_units, hipStream_t stream);

// BGRA -> NV12 encoder (bt709_encode.hip).  One frame per launch.
struct EncodeFrame {
  const uint8_t *bgra;  // W x H words (A<<24)|(R<<16)|(G<<8)|B, alpha ignored
  uint8_t *y;
  uint8_t *cbcr;
};
struct EncodeParams {
  EncodeFrame frames[kMaxBatch];
  // uniform != 0: frame i = frames[0] + i * step_* (bytes), as in DecodeParams
  uint32_t uniform;
  int64_t step_bgra, step_y, step_cbcr;
  const EncodeByteEntry *per_byte;  // 256 entries for the (input gamma, output gamma) pair
  const TransferBucket *from_linear;  // two-resolution BT709_from_linear(., output gamma) table (SplitTable)
  uint32_t from_linear_bytes;
  float from_linear_scale;    // its n_fine
  float from_linear_split;    // fine buckets below this xs ...
  float from_linear_coarse;   // ... coarse ones above: q = (uint)(xs * coarse) + offset
  uint32_t from_linear_offset;
  uint32_t row_pairs_per_block;  // consecutive row pairs a fast-path workgroup walks; 0 = encode_row_pairs_per_block()
  uint32_t block_threads;        // fast-path workgroup size (one quad per lane); 0 = encode_block_threads(width)
  uint32_t width, height;
  uint32_t bgra_stride, y_stride, cbcr_stride;
  uint32_t xcd_bands, frames_per_band;  // XCD-aware work map, as in DecodeParams (filled by launch_encode)
};
const char *launch_encode(const EncodeParams &p, int frames, bool fast, bool xcd_bands, hipStream_t stream);
hipError_t prepare_encode_kernels();
// Encoder fast-path geometry, measured on 4K (tools/encode_shapes.sh sweeps):
// one quad per lane, equal tiles of <= 320 lanes rounded up to whole waves (3840 -> 3 x 320,
// 1920 -> 2 x 256); a workgroup walks 3 to 9 consecutive row pairs (amortising its 8 KiB of
// table staging): the most that still leaves the launch >= 4096 workgroups.
inline uint32_t encode_block_threads(uint32_t width) {
  const uint32_t quads = width / 4, tiles = quads == 0 ? 1 : (quads + 319) / 320;
  uint32_t t = ((quads + tiles - 1) / tiles + 63) / 64 * 64;
  return t < 64 ? 64 : t;
}
inline uint32_t encode_row_pairs_per_block(uint32_t width, uint32_t height, uint32_t frames) {
  const uint32_t threads = encode_block_threads(width);
  const uint64_t tiles = (width / 4 + threads - 1) / threads;
  for (uint32_t rp = 9; rp > 3; --rp)
    if (tiles * ((height / 2 + rp - 1) / rp) * frames >= 4096) return rp;
  return 3;
}

// Planar U,V <-> interleaved CbCr (bt709_planes.hip).  The kernel that reads a plane never writes it.
struct PlaneParams {
  const uint8_t *u;  // chroma_width x chroma_height bytes (written by deinterleave_cbcr)
  const uint8_t *v;
  uint8_t *cbcr;     // chroma_width byte pairs per row (read by deinterleave_cbcr)
  uint32_t u_stride, v_stride, cbcr_stride;
  uint32_t chroma_width, chroma_height;
  uint32_t wide;     // 1: 8 samples per lane with 8/16-byte accesses
};
const char *launch_planes(const PlaneParams &p, bool interleave, hipStream_t stream);
// 16-byte-per-lane non-temporal streaming copy (bt709_planes.hip): the same-box copy ceiling benchmarks report.
const char *launch_copy_probe(void *dst, const void *src, size_t bytes, hipStream_t stream);

// Shape of the kernel launches the last bt709hip_decode[_batch] call of this thread made (tests assert the XCD-aware map
// through it: bt709hip_last_launch_info): grid and block of its FIRST launch, how many launches it took.
struct LaunchShape {
  uint32_t grid[3], block[3];
  int32_t launches, xcd_bands;
};
LaunchShape &last_launch_shape();  // thread-local (bt709_kernels.hip)

// Launchers return the kernel's name (static string) for profiling; launch errors
// are read by the caller with hipGetLastError().
// decode: variant kVariantQuads -> grid = (grid_x tiles, H/2, frames) x block_threads;
//         variant kVariantBlocks -> grid = (grid_x, frames) x kBlockThreads, grid-strided.
// quantiser: the decoder's mode is sRGB (arithmetic transfer step, no table); implied by has_alpha
// xcd_bands: use the XCD-aware work map when the launch allows it (fast path, frames a multiple of 8)
const char *launch_decode(const DecodeParams &p, int frames, int variant, bool has_alpha, bool quantiser, bool nontemporal,
                          int xcd_bands, uint32_t grid_x, uint32_t block_threads, hipStream_t stream);
// +unconvert: on packed 4:4:4 words (bt709_kernels.hip unconvert_packed444); `tables` supplies table_unit*, unit_magic, alpha_word.
// `count` frames of one geometry in one launch (grid.z): evenly spaced (in[0] + i * in_step), or through a pointer table of up to kMaxBatch
struct UnconvertBatch {
  int count;
  bool uniform;
  const void *const *in;  // count input pointers (uniform: only in[0] is read)
  void *const *out;
  int64_t in_step, out_step;
};
const char *launch_unconvert(const DecodeParams &tables, const UnconvertBatch &batch, size_t in_stride, size_t out_stride, uint32_t width,
                             uint32_t height, bool vec, bool quantiser, hipStream_t stream);
// half: grid = (grid_x, H/2 output rows, frames) x block_threads.
const char *launch_decode_half(const DecodeParams &p, int frames, bool wide, bool has_alpha, bool nontemporal,
                               uint32_t grid_x, uint32_t block_threads, hipStream_t stream);

// half, conflict-free form: grid = workgroups (<= compute units, one resident per CU) x block_threads,
// each walking tile rows workgroup, workgroup + grid, ...; fills the rep_*, tiles_x, tile_rows and cursor_*
// fields of its copy of `p`.  Returns nullptr when the tables do not fit LDS (caller falls back).
// lds_budget: LDS bytes one workgroup may take (kRepLdsBytes = one workgroup per CU).
const char *launch_decode_half_rep(const DecodeParams &p, int frames, bool has_alpha, bool nontemporal, uint32_t workgroups,
                                   uint32_t lds_budget, hipStream_t stream);
constexpr int kRepBlockThreads = 1024;      // one workgroup per CU: 16 waves
constexpr uint32_t kRepLdsBytes = 160 * 1024;

// scaled: grid = (ceil(OW / kBlockThreads), ceil(OH / rows), frames) x kBlockThreads, rows chosen from the CU count.
// in_align: largest power of two dividing every input plane pointer, pitch and frame spacing (picks the tap fetch width).
// nullptr: a plane of 2 GiB or more (row offsets are formed in 32 bits)
const char *launch_decode_scaled(const DecodeParams &p, int frames, bool has_alpha, uint32_t in_align,
                                 uint32_t compute_units, hipStream_t stream);

// Fast-path launch geometry for a frame width: tiles (workgroups) per row pair and the
// workgroup size -- ceil(quads per tile / kQuadsPerLane) rounded up to a whole wave.
//   3840 -> 1 tile x 512 threads (480 rounded up to whole waves), 1920 -> 1 x 256, 7680 -> 2 x 512.
inline uint32_t quads_tiles(uint32_t width) {
  const uint32_t quads = width / 4, cap = kMaxBlockThreads * kQuadsPerLane;
  return quads == 0 ? 1 : (quads + cap - 1) / cap;
}
inline uint32_t quads_block_threads(uint32_t width) {
  const uint32_t tiles = quads_tiles(width);
  const uint32_t per_tile = (width / 4 + tiles - 1) / tiles;
  uint32_t t = (per_tile + kQuadsPerLane - 1) / kQuadsPerLane;
  t = (t + 63) / 64 * 64;
  if (t < 64) t = 64;
  if (t > static_cast<uint32_t>(kMaxBlockThreads)) t = kMaxBlockThreads;
  return t;
}

// Row pairs stacked in one workgroup (blockDim.y): only when one tile spans the row and the
// row needs few threads, so the workgroup still has up to kMaxBlockThreads threads.
inline uint32_t quads_rows_per_block(uint32_t block_threads, uint32_t tiles) {
  if (tiles != 1 || block_threads == 0) return 1;
  const uint32_t by = kMaxBlockThreads / block_threads;
  return by < 1 ? 1 : by;
}

// Raise the dynamic-LDS cap of the kernels (tables can exceed the 64 KiB default).
hipError_t prepare_kernels();          // bt709_kernels.hip
hipError_t prepare_rescale_kernels();  // bt709_rescale_half.hip (+ bt709_rescale_scaled.hip's)

}  // namespace bt709
