// C++ twin of EmptyiOSTests/MetalBT709DecoderTests.m:189-277 over host/MetalBT709Decoder.hpp:
// a 2x2 frame of one (Y,Cb,Cr), tagged BT.709, decoded with the default Apple gamma, read
// back and compared with the expectation the reference test asserts.  Vectors are passed
// on the command line (tests/test_gpu_parity.py feeds tests/golden/vectors.json), so this
// file holds no oracle and no reference data.
//
//   g++ -std=c++17 host/decoder_selftest.cpp -Lmetalbt709decoder_amd -lbt709hip -o selftest
//   ./selftest Y Cb Cr R G B [Y Cb Cr R G B ...]
#include <cstdio>
#include <cstdlib>

#include "MetalBT709Decoder.hpp"

using namespace bt709;

int main(int argc, char **argv) {
  if (argc < 7 || (argc - 1) % 6 != 0) {
    std::fprintf(stderr, "usage: %s Y Cb Cr R G B [...]\n", argv[0]);
    return 2;
  }
  MetalRenderContext metalRenderContext;
  MetalBT709Decoder metalDecoder;
  metalDecoder.metalRenderContext = &metalRenderContext;
  if (!metalDecoder.setupMetal()) return 3;

  int failures = 0;
  for (int i = 1; i + 5 < argc; i += 6) {
    const uint32_t Y = std::atoi(argv[i]), Cb = std::atoi(argv[i + 1]), Cr = std::atoi(argv[i + 2]);
    const uint32_t R = std::atoi(argv[i + 3]), G = std::atoi(argv[i + 4]), B = std::atoi(argv[i + 5]);
    const int width = 2, height = 2;
    uint32_t outBT709[width * height];
    for (uint32_t &p : outBT709) p = (Cr << 16) | (Cb << 8) | Y;

    BGRATexture bgraSRGBTexture(metalRenderContext, width, height);
    CVPixelBuffer yCbCrBuffer(metalRenderContext, width, height);
    yCbCrBuffer.setBT709Attributes();
    yCbCrBuffer.copyBT709ToCoreVideo(outBT709);

    const bool worked = metalDecoder.decodeBT709(&yCbCrBuffer, nullptr, &bgraSRGBTexture, nullptr, nullptr, width,
                                                 height, true);
    if (!worked) {
      ++failures;
      continue;
    }
    for (uint32_t px : bgraSRGBTexture.getBGRATexturePixels()) {
      if (px != (0xFF000000u | (R << 16) | (G << 8) | B)) {
        std::fprintf(stderr, "(%u %u %u): got %08x, want (%u %u %u)\n", Y, Cb, Cr, px, R, G, B);
        ++failures;
      }
    }
  }
  // validation behaviour: render size mismatch returns false (MetalBT709Decoder.m:284-290)
  {
    BGRATexture tex(metalRenderContext, 2, 2);
    CVPixelBuffer buf(metalRenderContext, 2, 2);
    buf.setBT709Attributes();
    const uint32_t px[4] = {0x808010, 0x808010, 0x808010, 0x808010};
    buf.copyBT709ToCoreVideo(px);
    if (metalDecoder.decodeBT709(&buf, nullptr, &tex, nullptr, nullptr, 4, 4, true) ||
        metalDecoder.lastStatus() != BT709HIP_ERR_SIZE_MISMATCH)
      ++failures;
  }
  std::printf("%s: %d vectors, %d failures\n", failures ? "FAIL" : "ok", (argc - 1) / 6, failures);
  return failures ? 1 : 0;
}
