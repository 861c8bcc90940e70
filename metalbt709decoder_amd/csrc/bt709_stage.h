// LDS staging helper shared by every kernel file (device code only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace bt709 {

// LDS staging of n elements, d[i] = src_of(i): a lane issues up to kBatch of its loads before its first write.  Written as
// `for (i = tid; i < n; i += nthreads) d[i] = s[i]` hipcc makes every round load -> wait -> ds_write, i.e. one L2 round
// trip per round inside the workgroup's lifetime (stage_table below has the measurement).
template <typename T, typename SrcOf>
__device__ __forceinline__ void stage_batched(T *d, uint32_t n, uint32_t tid, uint32_t nthreads, SrcOf src_of) {
  constexpr int kBatch = 4;
  for (uint32_t base = tid; base < n; base += nthreads * kBatch) {
    T v[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const uint32_t i = base + static_cast<uint32_t>(k) * nthreads;
      if (i < n) v[k] = src_of(i);
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const uint32_t i = base + static_cast<uint32_t>(k) * nthreads;
      if (i < n) d[i] = v[k];
    }
  }
}

}  // namespace bt709
