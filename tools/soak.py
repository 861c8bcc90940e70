"""Random soak of every GPU path against the oracle (runs on the GPU box):
    python tools/soak.py [seed] [cases]
decode (opaque / alpha, 8-bit and RGBA16Float targets), exact 2:1 (both kernels, with alpha), any-ratio
(with alpha), the reference's two passes through both intermediate formats, the encoder, the frame ring and the
coalescing submit (round 4); round 5: ring sets on every visible device, batched RGBA16Float launches of both kernel
shapes, +unconvert: batches, the LINEAR mode's log-bucket kernel at many batch sizes.
Fresh seeds every time it is used; the committed tests hold the fixed-seed fuzz."""
import os
import sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_helpers as gh
from oracle_lib import Oracle
import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
oracle = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = gh.context()
scale = mb.MetalScaleRenderContext(); assert scale.setupRenderPipelines(ctx)
bad, counts = 0, {}
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    kind = int(rng.integers(0, 12))
    counts[kind] = counts.get(kind, 0) + 1
    gamma = int(rng.integers(0, 4))
    w = 4 * int(rng.integers(1, 600)); h = 4 * int(rng.integers(1, 40))
    y, c = gh.random_nv12(w, h, seed=int(rng.integers(0, 1 << 30)), legal=bool(rng.integers(0, 4) == 0))
    a = rng.integers(0, 256, (h, w), dtype=np.uint8) if rng.integers(0, 3) == 0 else None
    g = gamma if a is None else mb.MetalBT709GammaSRGB
    if kind == 0:
        got = gh.gpu_decode(y, c, g, alpha=a); want = oracle.decode_nv12(g, y, c, alpha=a)
    elif kind == 1:
        dec = gh.make_decoder(g, has_alpha=a is not None, options={_capi.OPT_HALF_KERNEL: int(rng.integers(-1, 2)),
                                                                    _capi.OPT_HALF_WORKGROUPS: int(rng.integers(1, 400))})
        got = gh.gpu_decode_half(y, c, g, decoder=dec, alpha=a); want = oracle.decode_nv12_half(g, y, c, alpha=a)
    elif kind == 2:
        ow, oh = int(rng.integers(1, 2 * w)), int(rng.integers(1, 2 * h))
        if rng.integers(0, 4) == 0:  # tall, narrow views: several output rows per strip, ragged last wave
            ow, oh = int(rng.integers(1, 140)), int(rng.integers(4100, 12000))
        got = gh.gpu_decode_scaled(y, c, (ow, oh), g, alpha=a); want = oracle.decode_nv12_scaled(g, y, c, ow, oh, alpha=a)
    elif kind == 3:  # RGBA16Float target
        dec = gh.make_decoder(g, has_alpha=a is not None)
        tex = ctx.makeBGRATexture((w, h), pixelFormat=mb.MTLPixelFormatRGBA16Float)
        assert dec.decodeBT709(gh.make_buffer(y, c, dec.gamma), gh.make_alpha_buffer(a) if a is not None else None, tex, None, None, w, h, True)
        got = ctx.getBGRATexturePixels(tex).view(np.uint16); want = oracle.decode_nv12_rgba16f(g, y, c, alpha=a).view(np.uint16)
    elif kind == 4:  # two passes, either intermediate
        fmt = mb.MTLPixelFormatRGBA16Float if rng.integers(0, 2) else mb.MTLPixelFormatBGRA8Unorm_sRGB
        ow, oh = int(rng.integers(1, 2 * w)), int(rng.integers(1, 2 * h))
        if rng.integers(0, 4) == 0:
            ow, oh = int(rng.integers(1, 140)), int(rng.integers(4100, 12000))
        dec = gh.make_decoder(g, has_alpha=a is not None)
        inter, view = ctx.makeBGRATexture((w, h), pixelFormat=fmt), ctx.makeBGRATexture((ow, oh))
        assert dec.decodeBT709(gh.make_buffer(y, c, dec.gamma), gh.make_alpha_buffer(a) if a is not None else None, inter, None, None, w, h, False)
        assert scale.renderScaled(ctx, view, ow, oh, None, None, inter, True)
        got = ctx.getBGRATexturePixels(view).view(np.uint8).reshape(oh, ow * 4)
        src = oracle.decode_nv12_rgba16f(g, y, c, alpha=a) if fmt == mb.MTLPixelFormatRGBA16Float else oracle.decode_nv12(g, y, c, alpha=a)
        want = oracle.render_scaled(src, ow, oh)
    elif kind == 6:  # round 4: a device-resident ring (bt709hip_ring_*), a random sub-range in one launch, 1:1 or exact 2:1
        n, half = int(rng.integers(1, 12)), bool(rng.integers(0, 3) == 0)
        dec = gh.make_decoder(g, has_alpha=a is not None)
        ring = mb.FrameRing(dec, (w, h), n, halfScale=half, tries=1)
        frs = [gh.random_nv12(w, h, seed=int(rng.integers(0, 1 << 30))) for _ in range(n)]
        for i, (fy, fc) in enumerate(frs):
            ring.pixelBuffer(i).upload_planes(fy, fc)
            if a is not None:
                ab = ring.alphaPixelBuffer(i); ctx._upload(ab.y_ptr, ab.y_stride, a, None); ctx._sync(None)
        first = int(rng.integers(0, n)); count = int(rng.integers(1, n - first + 1))
        assert ring.decode(first, count, waitUntilCompleted=True)
        ow, oh = (w // 2, h // 2) if half else (w, h)
        got = np.concatenate([ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(-1) for i in range(first, first + count)])
        want = np.concatenate([(oracle.decode_nv12_half(g, fy, fc, alpha=a) if half else oracle.decode_nv12(g, fy, fc, alpha=a)).reshape(-1)
                               for fy, fc in frs[first:first + count]])
        ring.release()
    elif kind == 7:  # round 4: the coalescing submit, a random window and a random number of one-frame calls, read back without a sync
        n, window = int(rng.integers(1, 20)), int(rng.integers(2, 33))
        dec = gh.make_decoder(g, has_alpha=a is not None, options={_capi.OPT_COALESCE: window})
        frs = [gh.random_nv12(w, h, seed=int(rng.integers(0, 1 << 30))) for _ in range(n)]
        bufs = [gh.make_buffer(fy, fc, dec.gamma) for fy, fc in frs]
        abuf = gh.make_alpha_buffer(a) if a is not None else None
        texs = [ctx.makeBGRATexture((w, h)) for _ in range(n)]
        for b, t in zip(bufs, texs):
            assert dec.decodeBT709(b, abuf, t, None, None, w, h, False)
        got = np.concatenate([ctx.getBGRATexturePixels(t).view(np.uint8).reshape(-1) for t in reversed(texs)])
        want = np.concatenate([oracle.decode_nv12(g, fy, fc, alpha=a).reshape(-1) for fy, fc in reversed(frs)])
    elif kind == 8:  # round 5: a ring set over every visible device (twice around), a random sub-range, one launch per lane
        ndev = mb.load_library().bt709hip_device_count()
        devices = list(range(ndev)) * 2
        n, half = int(rng.integers(1, 9)), bool(rng.integers(0, 3) == 0)
        f16 = not half and bool(rng.integers(0, 3) == 0)  # RGBA16Float render targets (bt709hip_ring_options.format)
        rs = mb.FrameRingSet(devices, (w, h), n, gamma=g, hasAlphaChannel=a is not None, halfScale=half, tries=1,
                             pixelFormat=mb.MTLPixelFormatRGBA16Float if f16 else mb.MTLPixelFormatBGRA8Unorm_sRGB)
        assert rs.handle, rs.lastStatus
        frs = {}
        for lane, ring in enumerate(rs.lanes):
            for i in range(n):
                frs[lane, i] = gh.random_nv12(w, h, seed=int(rng.integers(0, 1 << 30)))
                ring.pixelBuffer(i).upload_planes(*frs[lane, i])
                if a is not None:
                    ab = ring.alphaPixelBuffer(i); ring.ctx._upload(ab.y_ptr, ab.y_stride, a, None); ring.ctx._sync(None)
        first = int(rng.integers(0, n)); count = int(rng.integers(1, n - first + 1))
        assert rs.decode(first, count) and rs.synchronize()
        got = np.concatenate([ring.ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(-1)
                              for ring in rs.lanes for i in range(first, first + count)])
        want = np.concatenate([(oracle.decode_nv12_half(g, *frs[lane, i], alpha=a) if half else
                                oracle.decode_nv12_rgba16f(g, *frs[lane, i], alpha=a).view(np.uint8) if f16 else
                                oracle.decode_nv12(g, *frs[lane, i], alpha=a).view(np.uint8)).reshape(-1)
                               for lane in range(len(devices)) for i in range(first, first + count)])
        rs.release()
    elif kind == 9:  # round 5: batched RGBA16Float launches (small frames: the small shape; many frames: the large one)
        n = int(rng.integers(1, 33))
        hh = h if rng.integers(0, 3) == 0 else min(h, 24)  # tall ones: the large shape, with slices when the rows are narrow
        if hh == h and rng.integers(0, 2) == 0:
            w = 4 * int(rng.integers(1, 120))
        dec = gh.make_decoder(g)
        frs = [gh.random_nv12(w, hh, seed=int(rng.integers(0, 1 << 30))) for _ in range(n)]
        bufs = [gh.make_buffer(fy, fc, dec.gamma) for fy, fc in frs]
        texs = [ctx.makeBGRATexture((w, hh), pixelFormat=mb.MTLPixelFormatRGBA16Float) for _ in range(n)]
        assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True), dec.lastStatus
        got = np.concatenate([ctx.getBGRATexturePixels(t).view(np.uint16).reshape(-1) for t in texs])
        want = np.concatenate([oracle.decode_nv12_rgba16f(g, fy, fc).view(np.uint16).reshape(-1) for fy, fc in frs])
    elif kind == 10:  # round 5: +unconvert: over a batch of packed 4:4:4 frames
        n = int(rng.integers(1, 33))
        ww, hh = (w if rng.integers(0, 2) else w + 2), min(h, 16)
        dec = gh.make_decoder(g, alpha_fill=int(rng.integers(0, 256)))
        frs = [rng.integers(0, 1 << 24, (hh, ww), dtype=np.uint32) for _ in range(n)]
        texs = [ctx.makeBGRATexture((ww, hh)) for _ in range(n)]
        assert mb.BGRAToBT709Converter.unconvertBatch(dec, frs, texs, ww, hh), dec.lastStatus
        got = np.concatenate([ctx.getBGRATexturePixels(t).reshape(-1) for t in texs])
        want = np.concatenate([oracle.unconvert_packed(g, f, ww, hh) | np.uint32(dec.alphaFill << 24) for f in frs])
    elif kind == 11:  # round 5: the LINEAR mode (its log-bucket table, decode_nv12_quads_log) at batch sizes on both sides of the work map's thresholds, ragged heights
        n = int(rng.integers(1, 24))
        hh = 2 * int(rng.integers(1, 12))
        dec = gh.make_decoder(mb.MetalBT709GammaLinear)
        frs = [gh.random_nv12(w, hh, seed=int(rng.integers(0, 1 << 30))) for _ in range(n)]
        bufs = [gh.make_buffer(fy, fc, dec.gamma) for fy, fc in frs]
        texs = [ctx.makeBGRATexture((w, hh)) for _ in range(n)]
        assert dec.decodeBT709Batch(bufs, texs, waitUntilCompleted=True), dec.lastStatus
        got = np.concatenate([ctx.getBGRATexturePixels(t).view(np.uint8).reshape(-1) for t in texs])
        want = np.concatenate([oracle.decode_nv12(2, fy, fc).reshape(-1) for fy, fc in frs])
    else:  # encoder: BGRA -> NV12, then compare planes
        ig, og = [(1, 0), (1, 1), (2, 2), (0, 0), (1, 2)][int(rng.integers(0, 5))]
        bgra = rng.integers(0, 1 << 32, w * h, dtype=np.uint32)
        tex = ctx.makeBGRATexture((w, h), pixels=bgra); buf = mb.CVPixelBuffer(ctx, w, h)
        assert mb.BGRAToBT709Converter.convertIntoCoreVideoBuffer(tex, buf, ig, og)
        gy, gc = buf.download_planes(); wy, wc = oracle.encode_nv12(bgra, w, h, ig, og)
        got, want = np.concatenate([gy.reshape(-1), gc.reshape(-1)]), np.concatenate([wy.reshape(-1), wc.reshape(-1)])
    if got is None or not np.array_equal(got, want):
        bad += 1; print("MISMATCH", case, kind, g, w, h, a is not None)
print("soak done:", sum(counts.values()), "cases by kind", dict(sorted(counts.items())), "mismatches:", bad)
