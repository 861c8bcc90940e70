// Chroma plane shuffles between the reference's on-disk 4:2:0 format and NV12.
//
// The reference writes YUV4MPEG2 "C420jpeg" files: per frame a Y plane, then a U (Cb) plane,
// then a V (Cr) plane (Renderer/y4m_writer.h:194-241), while the decoder consumes NV12, whose
// second plane interleaves Cb,Cr byte pairs (Renderer/BGRAToBT709Converter.m:1083).  These two
// kernels move between the layouts on the device: pure byte traffic (1 B read + 1 B written per
// chroma byte), HBM-bound, 16 bytes per lane on the interleaved side.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bt709_kernels.h"

namespace bt709 {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// bytes a0 a1 a2 a3 / b0 b1 b2 b3 -> a0 b0 a1 b1 | a2 b2 a3 b3
__device__ __forceinline__ void zip4(uint32_t a, uint32_t b, uint32_t &lo, uint32_t &hi) {
  lo = __builtin_amdgcn_perm(b, a, 0x05010400u);  // {b1,a1,b0,a0} MSB..LSB
  hi = __builtin_amdgcn_perm(b, a, 0x07030602u);
}
__device__ __forceinline__ void unzip4(uint32_t lo, uint32_t hi, uint32_t &a, uint32_t &b) {
  a = __builtin_amdgcn_perm(hi, lo, 0x06040200u);  // even bytes
  b = __builtin_amdgcn_perm(hi, lo, 0x07050301u);  // odd bytes
}

}  // namespace

// wide: chroma width % 8 == 0, planes 8-byte aligned, cbcr 16-byte aligned (strides too)
__global__ void __launch_bounds__(kBlockThreads)
interleave_cbcr(const PlaneParams p) {
  const uint32_t row = blockIdx.y;
  const uint8_t *u = p.u + static_cast<size_t>(row) * p.u_stride;
  const uint8_t *v = p.v + static_cast<size_t>(row) * p.v_stride;
  uint8_t *c = p.cbcr + static_cast<size_t>(row) * p.cbcr_stride;
  if (p.wide) {
    const uint32_t groups = p.chroma_width >> 3;  // 8 chroma samples per lane
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
      const u32x2 a = *reinterpret_cast<const u32x2 *>(u + 8 * g);
      const u32x2 b = *reinterpret_cast<const u32x2 *>(v + 8 * g);
      uint32_t o0, o1, o2, o3;
      zip4(a.x, b.x, o0, o1);
      zip4(a.y, b.y, o2, o3);
      const u32x4 o = {o0, o1, o2, o3};
      __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(c + 16 * g));
    }
  } else {
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < p.chroma_width; x += gridDim.x * blockDim.x) {
      c[2 * x] = u[x];
      c[2 * x + 1] = v[x];
    }
  }
}

__global__ void __launch_bounds__(kBlockThreads)
deinterleave_cbcr(const PlaneParams p) {
  const uint32_t row = blockIdx.y;
  uint8_t *u = const_cast<uint8_t *>(p.u) + static_cast<size_t>(row) * p.u_stride;
  uint8_t *v = const_cast<uint8_t *>(p.v) + static_cast<size_t>(row) * p.v_stride;
  const uint8_t *c = p.cbcr + static_cast<size_t>(row) * p.cbcr_stride;
  if (p.wide) {
    const uint32_t groups = p.chroma_width >> 3;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
      const u32x4 i = *reinterpret_cast<const u32x4 *>(c + 16 * g);
      uint32_t a0, b0, a1, b1;
      unzip4(i.x, i.y, a0, b0);
      unzip4(i.z, i.w, a1, b1);
      *reinterpret_cast<u32x2 *>(u + 8 * g) = u32x2{a0, a1};
      *reinterpret_cast<u32x2 *>(v + 8 * g) = u32x2{b0, b1};
    }
  } else {
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < p.chroma_width; x += gridDim.x * blockDim.x) {
      u[x] = c[2 * x];
      v[x] = c[2 * x + 1];
    }
  }
}

// Streaming copy, 16 bytes per lane, non-temporal both ways, one short-lived workgroup per 16 KiB
// dispatched in address order (the launch shape the decode kernels use): what a plain copy reaches
// on this device.  n16 = number of 16-byte units.
__global__ void __launch_bounds__(512)
copy_probe(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t n16) {
  const size_t i0 = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x, i1 = i0 + 512;
  u32x4 a = {}, b = {};
  if (i0 < n16) a = __builtin_nontemporal_load(src + i0);
  if (i1 < n16) b = __builtin_nontemporal_load(src + i1);
  if (i0 < n16) __builtin_nontemporal_store(a, dst + i0);
  if (i1 < n16) __builtin_nontemporal_store(b, dst + i1);
}

const char *launch_copy_probe(void *dst, const void *src, size_t bytes, hipStream_t stream) {
  const size_t n16 = bytes / 16;
  const size_t blocks = (n16 + 1023) / 1024;
  hipLaunchKernelGGL(copy_probe, dim3(static_cast<uint32_t>(blocks)), dim3(512), 0, stream, static_cast<u32x4 *>(dst),
                     static_cast<const u32x4 *>(src), n16);
  return "copy_probe";
}

const char *launch_planes(const PlaneParams &p, bool interleave, hipStream_t stream) {
  const uint32_t items = p.wide ? p.chroma_width / 8 : p.chroma_width;
  uint32_t gx = (items + kBlockThreads - 1) / kBlockThreads;
  if (gx < 1) gx = 1;
  const dim3 grid(gx, p.chroma_height, 1);
  if (interleave) {
    hipLaunchKernelGGL(interleave_cbcr, grid, dim3(kBlockThreads), 0, stream, p);
    return "interleave_cbcr";
  }
  hipLaunchKernelGGL(deinterleave_cbcr, grid, dim3(kBlockThreads), 0, stream, p);
  return "deinterleave_cbcr";
}

}  // namespace bt709
