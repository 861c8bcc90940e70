// Kernel parameter block and launchers shared by bt709_kernels.hip and the C-ABI shim.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "transfer_tables.h"

namespace bt709 {

constexpr int kBlockThreads = 256;  // 4 waves of 64
constexpr int kMaxBatch = 32;       // == BT709HIP_MAX_BATCH

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

// Passed by value in the kernarg segment (32 frames x 32 B + 64 B).
struct DecodeParams {
  FramePlanes frames[kMaxBatch];
  const void *table;   // TransferBucket[] (decode) or TransferBucketLinear[] (half)
  const void *table2;  // half only: LINEAR-mode TransferBucket[] used as the sRGB encoder
  uint32_t table_bytes;
  uint32_t table2_bytes;
  float table_scale;   // N of `table`
  float table2_scale;  // N of `table2`
  uint32_t width;      // luma (source) dimensions
  uint32_t height;
  uint32_t y_stride;
  uint32_t cbcr_stride;
  uint32_t alpha_stride;
  uint32_t out_stride;
  uint32_t alpha_word;  // alpha_fill << 24
};

// Launchers return the kernel's name (static string) for profiling; launch errors
// are read by the caller with hipGetLastError().
const char *launch_decode(const DecodeParams &p, int frames, int variant, bool has_alpha, bool nontemporal,
                          uint32_t grid_x, hipStream_t stream);
const char *launch_decode_half(const DecodeParams &p, int frames, bool wide, bool nontemporal, uint32_t grid_x,
                               hipStream_t stream);

// Raises the dynamic-LDS cap of the half kernels (two tables can exceed 64 KiB).
hipError_t prepare_kernels();

}  // namespace bt709
