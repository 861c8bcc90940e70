#!/usr/bin/env python3
"""Writes a YUV4MPEG2 C420jpeg clip (the reference's on-disk format, Renderer/y4m_writer.h:61-241) for `bench.py --y4m`:
N frames of 1920x1080 made from the bundled QuickTime HD test pattern THROUGH THE GPU ENCODER, so that no media file lives
in git.

    python tools/make_y4m_clip.py OUT.y4m [--frames 64] [--fps 60]

The picture: tests/golden/patterns_full.npz holds the pattern (Renderer/QuickTime_Test_Pattern_HD_sRGB.png) as the NV12 frame
the reference's own encoder makes of it (tests/golden/make_golden.py).  That frame is decoded on the GPU to 8-bit sRGB BGRA
(MetalBT709Decoder, default gamma); frame i of the clip is the picture panned 4 i pixels to the right and 2 i down
(wrap-around), encoded by bt709hip_encode (BGRAToBT709Converter +convertIntoCoreVideoBuffer:, sRGB in, Apple gamma out --
the reference's linear-light 2x2 chroma averaging), de-interleaved on the device (bt709hip_deinterleave_cbcr) and written
with Y4MWriter.  Needs a GPU (the product has no CPU path)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def make_clip(path, frames=64, fps=60, ctx=None):
    import metalbt709decoder_amd as mb
    from metalbt709decoder_amd import y4m
    if ctx is None:
        ctx = mb.MetalRenderContext(0)
        assert ctx.setupMetal()
    z = np.load(os.path.join(ROOT, "tests", "golden", "patterns_full.npz"))
    y, c = z["qt_hd_srgb_full_y"], z["qt_hd_srgb_full_uv"]
    h, w = y.shape
    dec = mb.MetalBT709Decoder()
    dec.metalRenderContext = ctx
    dec.gamma = mb.MetalBT709GammaApple
    assert dec.setupMetal(), dec.lastStatus
    src = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (w, h))
    mb.BGRAToBT709Converter.setBT709Attributes(src)
    src.upload_planes(y, c)
    tex = ctx.makeBGRATexture((w, h))
    assert dec.decodeBT709(src, None, tex, None, None, w, h, True), dec.lastStatus
    picture = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(h, w, 4)
    out = mb.BGRAToBT709Converter.createCoreVideoYCbCrBuffer(ctx, (w, h))
    t = ctx.makeBGRATexture((w, h))
    with y4m.Y4MWriter(path, w, h, fps=fps) as wr:
        for i in range(frames):
            ctx.fillBGRATexture(t, np.roll(picture, (2 * i, 4 * i), (0, 1)))
            assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(t, out, mb.MetalBT709GammaSRGB, mb.MetalBT709GammaApple,
                                                                      waitUntilCompleted=True)
            wr.write_pixel_buffer(out)
    return w, h


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--fps", type=int, default=60)
    a = ap.parse_args()
    w, h = make_clip(a.out, a.frames, a.fps)
    print("wrote %s: %d frames of %dx%d, %d bytes" % (a.out, a.frames, w, h, os.path.getsize(a.out)))
