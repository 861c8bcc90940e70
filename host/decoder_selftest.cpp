// C++ twin of EmptyiOSTests/MetalBT709DecoderTests.m:189-277 over host/MetalBT709Decoder.hpp:
// a 2x2 frame of one (Y,Cb,Cr), tagged BT.709, decoded with the default Apple gamma, read
// back and compared with the expectation the reference test asserts.  Vectors are passed
// on the command line (tests/test_gpu_parity.py feeds tests/golden/vectors.json), so this
// file holds no oracle and no reference data.
//
//   g++ -std=c++17 host/decoder_selftest.cpp -Lmetalbt709decoder_amd -lbt709hip -o selftest
//   ./selftest [--host | --ring] Y Cb Cr R G B [Y Cb Cr R G B ...]
// --host: the same vectors with the frame and the texture in HOST memory, through the unchanged
// 8-argument selector's host overload (in-flight frame pool), three frames in flight.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "MetalBT709Decoder.hpp"

using namespace bt709;

// Host-memory flow: every vector is one 6x4 frame and one texture in plain host memory; frames are
// submitted without waiting (three in flight), then finished and compared.
static int run_host_vectors(MetalBT709Decoder &metalDecoder, int argc, char **argv) {
  const int width = 6, height = 4, n = (argc - 1) / 6;
  metalDecoder.deferredCompletion = true;  // keep three frames in flight; finishHostFrames() below
  std::vector<std::vector<uint8_t>> ys(n), cs(n), outs(n);
  int failures = 0;
  for (int v = 0; v < n; ++v) {
    const int i = 1 + 6 * v;
    ys[v].assign(static_cast<size_t>(width + 3) * height, static_cast<uint8_t>(std::atoi(argv[i])));  // pitch > width
    cs[v].resize(static_cast<size_t>(width) * (height / 2));
    for (size_t k = 0; k < cs[v].size(); k += 2) {
      cs[v][k] = static_cast<uint8_t>(std::atoi(argv[i + 1]));
      cs[v][k + 1] = static_cast<uint8_t>(std::atoi(argv[i + 2]));
    }
    outs[v].assign(static_cast<size_t>(width * 4 + 8) * height, 0x5A);
    HostPixelBuffer yCbCrBuffer;
    yCbCrBuffer.y = ys[v].data();
    yCbCrBuffer.yStride = width + 3;
    yCbCrBuffer.cbcr = cs[v].data();
    yCbCrBuffer.cbcrStride = width;
    yCbCrBuffer.width = width;
    yCbCrBuffer.height = height;
    HostTexture bgraSRGBTexture{outs[v].data(), static_cast<size_t>(width * 4 + 8), width, height};
    if (!metalDecoder.decodeBT709(yCbCrBuffer, nullptr, bgraSRGBTexture, nullptr, nullptr, width, height, v == n - 1))
      ++failures;
  }
  if (!metalDecoder.finishHostFrames()) ++failures;
  for (int v = 0; v < n; ++v) {
    const int i = 1 + 6 * v;
    const uint32_t want = 0xFF000000u | (static_cast<uint32_t>(std::atoi(argv[i + 3])) << 16) |
                          (static_cast<uint32_t>(std::atoi(argv[i + 4])) << 8) | static_cast<uint32_t>(std::atoi(argv[i + 5]));
    for (int row = 0; row < height; ++row) {
      const uint8_t *r = outs[v].data() + static_cast<size_t>(row) * (width * 4 + 8);
      for (int col = 0; col < width; ++col) {
        uint32_t px;
        std::memcpy(&px, r + 4 * col, 4);
        if (px != want) ++failures;
      }
      for (int k = width * 4; k < width * 4 + 8; ++k)
        if (r[k] != 0x5A) ++failures;  // row padding untouched
    }
  }
  // the overload validates like the reference: a BT.601-tagged buffer is refused
  HostPixelBuffer bad;
  bad.y = ys[0].data();
  bad.yStride = width + 3;
  bad.cbcr = cs[0].data();
  bad.cbcrStride = width;
  bad.width = width;
  bad.height = height;
  bad.matrix = BT709HIP_MATRIX_ITU_R_601_4;
  HostTexture tex{outs[0].data(), static_cast<size_t>(width * 4 + 8), width, height};
  if (metalDecoder.decodeBT709(bad, nullptr, tex, nullptr, nullptr, width, height, true) ||
      metalDecoder.lastStatus() != BT709HIP_ERR_MATRIX)
    ++failures;
  // The one-pass route of the renderer (AAPLRenderer.m:927-934): bgraSRGBTexture nil, the target is the render pass
  // descriptor's colour attachment -- here a drawable LARGER than the frame, the frame in its top-left viewport --
  // and waitUntilCompleted FALSE with the default (non-deferred) completion: the pixels are there on return.
  metalDecoder.deferredCompletion = false;
  {
    const int dw = width + 4, dh = height + 2, i = 1;
    std::vector<uint8_t> drawable(static_cast<size_t>(dw) * 4 * dh, 0x5A);
    HostRenderPassDescriptor rpd;
    rpd.colorAttachment0Texture = HostTexture{drawable.data(), static_cast<size_t>(dw) * 4, dw, dh};
    HostPixelBuffer buf = bad;
    buf.matrix = BT709HIP_MATRIX_ITU_R_709_2;
    if (!metalDecoder.decodeBT709(buf, nullptr, static_cast<const HostTexture *>(nullptr), nullptr, &rpd, width, height, false)) ++failures;
    const uint32_t want = 0xFF000000u | (static_cast<uint32_t>(std::atoi(argv[i + 3])) << 16) |
                          (static_cast<uint32_t>(std::atoi(argv[i + 4])) << 8) | static_cast<uint32_t>(std::atoi(argv[i + 5]));
    for (int row = 0; row < dh; ++row)
      for (int col = 0; col < dw; ++col) {
        uint32_t px;
        std::memcpy(&px, drawable.data() + (static_cast<size_t>(row) * dw + col) * 4, 4);
        if (px != (row < height && col < width ? want : 0x5A5A5A5Au)) ++failures;
      }
    // nil texture AND nil descriptor: nothing to render into
    if (metalDecoder.decodeBT709(buf, nullptr, static_cast<const HostTexture *>(nullptr), nullptr, nullptr, width, height, true) ||
        metalDecoder.lastStatus() != BT709HIP_ERR_INVALID_ARG)
      ++failures;
    // a drawable smaller than the frame cannot hold the viewport
    rpd.colorAttachment0Texture.width = width - 2;
    if (metalDecoder.decodeBT709(buf, nullptr, static_cast<const HostTexture *>(nullptr), nullptr, &rpd, width, height, true) ||
        metalDecoder.lastStatus() != BT709HIP_ERR_SIZE_MISMATCH)
      ++failures;
  }
  // The same vectors through the in-process frame sharder: three lanes, all on device 0, frame i -> lane i mod 3.
  {
    FrameSharder sharder({0, 0, 0}, width, height, MetalBT709GammaApple, false, 2);
    if (!sharder.valid() || sharder.lanes() != 3) ++failures;
    std::vector<long long> tickets(n, -1);
    for (int v = 0; v < n && sharder.valid(); ++v) {
      HostPixelBuffer b;
      b.y = ys[v].data(), b.yStride = width + 3, b.cbcr = cs[v].data(), b.cbcrStride = width, b.width = width, b.height = height;
      tickets[v] = sharder.submit(b);
      if (tickets[v] != v) ++failures;
      const int done = v - 5;  // lanes * depth = 6 frames stay valid
      for (int w = (v == n - 1 ? (done < 0 ? 0 : done) : done); w >= 0 && w <= (v == n - 1 ? v : done); ++w) {
        const int i = 1 + 6 * w;
        size_t stride = 0;
        const uint8_t *rows = sharder.wait(tickets[w], &stride);
        const uint32_t want = 0xFF000000u | (static_cast<uint32_t>(std::atoi(argv[i + 3])) << 16) |
                              (static_cast<uint32_t>(std::atoi(argv[i + 4])) << 8) | static_cast<uint32_t>(std::atoi(argv[i + 5]));
        if (rows == nullptr) {
          ++failures;
          continue;
        }
        for (int row = 0; row < height; ++row)
          for (int col = 0; col < width; ++col) {
            uint32_t px;
            std::memcpy(&px, rows + static_cast<size_t>(row) * stride + 4 * col, 4);
            if (px != want) ++failures;
          }
      }
    }
  }
  std::printf("%s: %d host vectors, %d failures\n", failures ? "FAIL" : "ok", n, failures);
  return failures ? 1 : 0;
}

// The reference's literal two passes (-decodeBT709 into an intermediate, then -renderScaled:) against the
// fused call: through a BGRA8 intermediate the pixels must be identical; through RGBA16Float they must run.
static int run_two_pass(MetalRenderContext &ctx, MetalBT709Decoder &dec) {
  const int w = 32, h = 16, ow = 20, oh = 10;
  std::vector<uint32_t> packed(static_cast<size_t>(w) * h);
  uint32_t st = 709;
  for (uint32_t &p : packed) p = (st = st * 1664525u + 1013904223u) >> 8;  // Y | Cb << 8 | Cr << 16, full range
  CVPixelBuffer buf(ctx, w, h);
  buf.setBT709Attributes();
  buf.copyBT709ToCoreVideo(packed.data());
  BGRATexture fused(ctx, ow, oh), view(ctx, ow, oh), inter8(ctx, w, h), inter16(ctx, w, h, BT709HIP_FORMAT_RGBA16F);
  MetalScaleRenderContext scale;
  int failures = 0;
  if (!scale.setupRenderPipelines(ctx)) ++failures;
  if (!dec.decodeBT709Scaled(&buf, &fused, nullptr, true)) ++failures;
  if (!dec.decodeBT709(&buf, nullptr, &inter8, nullptr, nullptr, w, h, false)) ++failures;
  if (!scale.renderScaled(ctx, view, ow, oh, nullptr, nullptr, inter8, true)) ++failures;
  if (fused.getBGRATexturePixels() != view.getBGRATexturePixels()) ++failures;
  if (!dec.decodeBT709(&buf, nullptr, &inter16, nullptr, nullptr, w, h, false)) ++failures;
  if (!scale.renderScaled(ctx, view, ow, oh, nullptr, nullptr, inter16, true)) ++failures;
  if (scale.renderScaled(ctx, view, ow + 2, oh, nullptr, nullptr, inter16, true) ||
      scale.lastStatus() != BT709HIP_ERR_SIZE_MISMATCH)
    ++failures;
  std::printf("%s: two-pass pipeline, %d failures\n", failures ? "FAIL" : "ok", failures);
  return failures ? 1 : 0;
}

// --ring: every vector is one 8 x 4 frame of a device-resident FrameRing (bt709hip_ring_*).  First the whole ring in one
// launch; then the reference's cadence -- one decodeBT709 call per frame, waitUntilCompleted FALSE -- with the coalescing
// submit on (BT709HIP_OPT_COALESCE = 4): calls are queued, the read-back of the LAST frame issues what is left.
static int run_ring_vectors(MetalBT709Decoder &metalDecoder, int argc, char **argv) {
  const int n = (argc - 1) / 6, width = 8, height = 4;
  if (n < 1) return 2;
  FrameRing ring(metalDecoder, width, height, n, false, 1);
  if (!ring.valid() || ring.frames() != n) return 3;
  std::vector<uint32_t> want(static_cast<size_t>(n));
  for (int v = 0; v < n; ++v) {
    const int i = 1 + 6 * v;
    std::vector<uint8_t> y(static_cast<size_t>(width) * height, static_cast<uint8_t>(std::atoi(argv[i])));
    std::vector<uint8_t> c(static_cast<size_t>(width) * height / 2);
    for (size_t k = 0; k < c.size(); k += 2) c[k] = static_cast<uint8_t>(std::atoi(argv[i + 1])), c[k + 1] = static_cast<uint8_t>(std::atoi(argv[i + 2]));
    if (!ring.upload(v, y.data(), c.data())) return 3;
    want[static_cast<size_t>(v)] = 0xFF000000u | (static_cast<uint32_t>(std::atoi(argv[i + 3])) << 16) |
                                   (static_cast<uint32_t>(std::atoi(argv[i + 4])) << 8) | static_cast<uint32_t>(std::atoi(argv[i + 5]));
  }
  int failures = 0;
  auto check_all = [&](const char *what) {
    for (int v = n - 1; v >= 0; --v)  // last frame first: with queued frames, its read-back is what issues them
      for (uint32_t px : ring.pixels(v))
        if (px != want[static_cast<size_t>(v)]) {
          std::fprintf(stderr, "%s, frame %d: got %08x, want %08x\n", what, v, px, want[static_cast<size_t>(v)]);
          ++failures;
          break;
        }
  };
  if (!ring.decode(0, n, true)) return 3;
  check_all("one launch");
  if (!ring.clearOutputs()) return 3;  // the second pass must write every pixel again
  if (!metalDecoder.setCoalescing(4, 500)) return 3;
  for (int v = 0; v < n; ++v)
    if (!ring.decodeFrame(v, false)) ++failures;
  check_all("coalesced one-frame calls");
  if (!metalDecoder.setOption(BT709HIP_OPT_COALESCE, 0)) return 3;
  std::printf("%s: %d ring frames, %d failures\n", failures ? "FAIL" : "ok", n, failures);
  return failures ? 1 : 0;
}

// --devices all | N: every vector is one 8 x 4 frame; a FrameRingSet (bt709hip_ringset_*: ONE process, a ring per device) over every
// visible device -- or over N lanes, wrapping onto the visible devices -- decodes all of them on every lane with one launch per
// lane, and every lane's frames are read back and compared.  Lane l's ring holds the vectors rotated by l, so a lane that
// served another lane's pixels cannot pass.  With one GPU visible this is one lane on device 0; on an 8-GPU node it is eight.
static int run_ringset_vectors(const char *devices_arg, int argc, char **argv) {
  const int n = (argc - 1) / 6, width = 8, height = 4;
  if (n < 1) return 2;
  const int visible = bt709hip_device_count();
  if (visible <= 0) return 3;
  const int lanes = std::strcmp(devices_arg, "all") == 0 ? visible : std::atoi(devices_arg);
  if (lanes < 1 || lanes > 64) return 2;
  std::vector<int> devices;
  for (int l = 0; l < lanes; ++l) devices.push_back(l % visible);
  FrameRingSet set(devices, width, height, n, MetalBT709GammaApple, false, false, 1);
  if (!set.valid() || set.lanes() != lanes) return 3;
  int failures = 0;
  std::vector<std::string> seen;
  for (int l = 0; l < lanes; ++l) {
    const bt709hip_device_info info = set.laneDevice(l);
    if (info.device_ordinal != devices[static_cast<size_t>(l)]) ++failures;
    if (l < visible) {  // the first `visible` lanes sit on DISTINCT physical devices
      for (const std::string &s : seen)
        if (s == info.pci_bus_id) ++failures;
      seen.push_back(info.pci_bus_id);
    }
    std::printf("lane %d: device %d, pci %s, uuid %s\n", l, info.device_ordinal, info.pci_bus_id, info.uuid);
  }
  std::vector<uint32_t> want(static_cast<size_t>(n));
  for (int l = 0; l < lanes; ++l)
    for (int v = 0; v < n; ++v) {
      const int i = 1 + 6 * ((v + l) % n);  // lane l holds vector (v + l) mod n in frame v
      std::vector<uint8_t> y(static_cast<size_t>(width) * height, static_cast<uint8_t>(std::atoi(argv[i])));
      std::vector<uint8_t> c(static_cast<size_t>(width) * height / 2);
      for (size_t k = 0; k < c.size(); k += 2) c[k] = static_cast<uint8_t>(std::atoi(argv[i + 1])), c[k + 1] = static_cast<uint8_t>(std::atoi(argv[i + 2]));
      if (!set.upload(l, v, y.data(), c.data())) return 3;
    }
  if (!set.decode(0, n, false) || !set.synchronize()) return 3;
  for (int l = 0; l < lanes; ++l)
    for (int v = 0; v < n; ++v) {
      const int i = 1 + 6 * ((v + l) % n);
      const uint32_t w = 0xFF000000u | (static_cast<uint32_t>(std::atoi(argv[i + 3])) << 16) | (static_cast<uint32_t>(std::atoi(argv[i + 4])) << 8) |
                         static_cast<uint32_t>(std::atoi(argv[i + 5]));
      for (uint32_t px : set.pixels(l, v))
        if (px != w) {
          std::fprintf(stderr, "lane %d, frame %d: got %08x, want %08x\n", l, v, px, w);
          ++failures;
          break;
        }
    }
  std::printf("%s: %d lanes on %d visible device(s), %d ring frames each, %d failures\n", failures ? "FAIL" : "ok", lanes, visible, n, failures);
  return failures ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc > 2 && std::strcmp(argv[1], "--devices") == 0) {
    if ((argc - 3) < 6 || (argc - 3) % 6 != 0) return 2;
    return run_ringset_vectors(argv[2], argc - 2, argv + 2);
  }
  if (argc > 1 && std::strcmp(argv[1], "--two-pass") == 0) {
    MetalRenderContext ctx;
    MetalBT709Decoder dec;
    dec.metalRenderContext = &ctx;
    if (!dec.setupMetal()) return 3;
    return run_two_pass(ctx, dec);
  }
  const bool ring = argc > 1 && std::strcmp(argv[1], "--ring") == 0;
  if (ring) {
    --argc;
    ++argv;
  }
  const bool host = argc > 1 && std::strcmp(argv[1], "--host") == 0;
  if (host) {
    --argc;
    ++argv;
  }
  if (argc < 7 || (argc - 1) % 6 != 0) {
    std::fprintf(stderr, "usage: %s [--host] Y Cb Cr R G B [...]\n", argv[0]);
    return 2;
  }
  MetalRenderContext metalRenderContext;
  MetalBT709Decoder metalDecoder;
  metalDecoder.metalRenderContext = &metalRenderContext;
  if (!metalDecoder.setupMetal()) return 3;
  if (host) return run_host_vectors(metalDecoder, argc, argv);
  if (ring) return run_ring_vectors(metalDecoder, argc, argv);

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
