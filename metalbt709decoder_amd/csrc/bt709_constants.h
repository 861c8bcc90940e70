// Constants of the BT.709 video-range YCbCr -> RGB step, shared by the HIP kernels
// and the host-side table builder.
//
// Follows the reference's CPU path (paths relative to the reference repo), NOT its
// Metal shader: the shader uses 4-decimal constants and a different operation
// order (Renderer/AAPLShaders.metal:194-227) and is numerically different; the
// parity target is Renderer/BT709.h.
//
//   Renderer/BT709.h:40-59    Kr, Kg, Kb, chroma spans, video ranges
//   Renderer/BT709.h:386-397  matrix built in float, in this expression order
//   Renderer/BT709.h:494-500  (Y-16)*(1/255f), (C-128)*(1/255f)
//
// Everything is a constant expression evaluated in IEEE binary32 by the compiler,
// exactly as the reference's `const float` initialisers are; the resulting bit
// patterns are asserted in tests/test_host_cpu.py.
#pragma once

namespace bt709 {

constexpr float kKr = 0.2126f;
constexpr float kKg = 0.7152f;
constexpr float kKb = 0.0722f;
constexpr float kCrSpan = 1.5748f;  // BT709_Er_minus_Ey_Range
constexpr float kCbSpan = 1.8556f;  // BT709_Eb_minus_Ey_Range
constexpr float kKrOverKg = kKr / kKg;
constexpr float kKbOverKg = kKb / kKg;

constexpr int kYMin = 16, kYMax = 235, kCMin = 16, kCMax = 240;

constexpr float kInv255 = 1.0f / 255.0f;
constexpr float kYScale = 255.0f / (kYMax - kYMin);
constexpr float kCScale = 255.0f / (kCMax - kCMin);

// row-major 3x3 of BT709.h:389-397 without the two zero entries
constexpr float kMY = kYScale;                                    // [0],[3],[6]
constexpr float kMCrR = (kCScale * kCrSpan);                      // [2]
constexpr float kMCbG = (-1.0f * kCScale * kCbSpan * kKbOverKg);  // [4]
constexpr float kMCrG = (-1.0f * kCScale * kCrSpan * kKrOverKg);  // [5]
constexpr float kMCbB = (kCScale * kCbSpan);                      // [7]

}  // namespace bt709
