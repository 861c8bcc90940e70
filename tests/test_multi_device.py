"""Round 5 GPU tests: everything that is about WHICH device runs and about the order work reaches a stream.

The multi-device tests size themselves by bt709hip_device_count(): with one GPU visible (the builder's boxes) they run their
one-device form, on an 8-GPU node the same tests put a lane / a ring on every device by themselves -- nothing to edit, nothing
to remember.  Reference shape: one process that drives everything (Renderer/AAPLRenderer.m:874-985), one device
(Renderer/MetalRenderContext.m:59-63); its test flow reads back what it decoded (EmptyiOSTests/MetalBT709DecoderTests.m:189-277).
"""
import ctypes as C
import re
import subprocess
import time

import numpy as np
import pytest

import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    gpu_helpers.context()
    return gpu_helpers


def visible():
    n = mb.load_library().bt709hip_device_count()
    assert n >= 1
    return n


def lane_sets():
    """[every visible device once], [every visible device twice (ordinals may repeat)]"""
    n = visible()
    return [list(range(n)), [d for d in range(n)] * 2]


def test_device_identity_through_the_c_abi(gh):
    """bt709hip_context_info carries the PCI bus id and the UUID of the context's device: distinct per visible device, the
    same for two contexts on one device."""
    lib = mb.load_library()
    seen = {}
    for d in range(visible()):
        ctxs = [mb.MetalRenderContext(d) for _ in range(2)]
        infos = []
        for c in ctxs:
            assert c.setupMetal()
            infos.append(c.info())
        a, b = infos
        assert a.device_ordinal == d and a.pci_bus_id == b.pci_bus_id and a.uuid == b.uuid
        bus = a.pci_bus_id.decode()
        assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-7]", bus), bus
        assert a.uuid.decode() == "" or re.fullmatch(r"[0-9a-f]{32}|[\x21-\x7e]{16}", a.uuid.decode())
        assert bus not in seen, "devices %d and %d report the same PCI bus id" % (seen.get(bus, -1), d)
        seen[bus] = d
        for c in ctxs:
            c.release()
    assert lib.bt709hip_context_info(None, None) == _capi.ERR_INVALID_ARG


@pytest.mark.parametrize("double", [False, True])
def test_ring_per_visible_device(gh, oracle, double):
    """bt709hip_ringset_*: ONE process, a ring per lane on range(bt709hip_device_count()) (and twice around), one launch per lane
    per step from this thread.  Every lane's ring holds frames of its own; every frame of every lane is byte-compared with the
    oracle; the lanes' devices are the ordinals asked for, the first round of them physically distinct."""
    devices = lane_sets()[1 if double else 0]
    w, h, n = 1920, 16, 8
    rs = mb.FrameRingSet(devices, (w, h), n, tries=1)
    assert rs.handle, rs.lastStatus
    assert len(rs.lanes) == len(devices) == rs.lib.bt709hip_ringset_lanes(rs.handle)
    buses = [ring.ctx.info().pci_bus_id for ring in rs.lanes]
    assert [ring.ctx.info().device_ordinal for ring in rs.lanes] == devices
    assert len(set(buses[:visible()])) == visible()
    frames = {}
    for lane, ring in enumerate(rs.lanes):
        for i in range(n):
            frames[lane, i] = gh.random_nv12(w, h, seed=7000 + 100 * lane + i)
            ring.pixelBuffer(i).upload_planes(*frames[lane, i])
    assert rs.decode() and rs.synchronize()
    for lane, ring in enumerate(rs.lanes):
        for i in range(n):
            got = ring.ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(h, w * 4)
            assert np.array_equal(got, oracle.decode_nv12(0, *frames[lane, i])), (lane, i)
    # a sub-range, waited for inside the call; argument errors
    for ring in rs.lanes:
        _capi.check(rs.lib.bt709hip_memset(ring.ctx.handle, ring.texture(0).ptr, 0, n * w * h * 4, None))
    assert rs.decode(2, 3, waitUntilCompleted=True)
    for lane, ring in enumerate(rs.lanes):
        for i in (1, 2, 4, 5):
            got = ring.ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(h, w * 4)
            assert np.array_equal(got, oracle.decode_nv12(0, *frames[lane, i])) == (2 <= i < 5), (lane, i)
    assert not rs.decode(6, 5) and rs.lastStatus == _capi.ERR_INVALID_ARG
    rs.release()
    lib, h_ = mb.load_library(), C.c_void_p()
    one = (C.c_int * 1)(visible())  # no such device
    assert lib.bt709hip_ringset_create(one, 1, 0, 0, 64, 16, 4, 0, 1, None, C.byref(h_)) == _capi.ERR_NO_DEVICE and not h_.value
    assert lib.bt709hip_ringset_create(None, 1, 0, 0, 64, 16, 4, 0, 1, None, C.byref(h_)) == _capi.ERR_INVALID_ARG
    assert lib.bt709hip_ringset_decode(None, 0, 1, 0) == _capi.ERR_INVALID_ARG and lib.bt709hip_ringset_destroy(None) == _capi.OK


def test_ring_set_alpha_half_scale(gh, oracle):
    """A ring set of alpha decoders with 2:1 outputs: the set passes gamma / alpha / half-scale through to every lane."""
    devices = lane_sets()[0]
    rs = mb.FrameRingSet(devices, (64, 16), 3, hasAlphaChannel=True, halfScale=True, tries=1)
    assert rs.handle, rs.lastStatus
    for lane, ring in enumerate(rs.lanes):
        assert ring.decoder.gamma == mb.MetalBT709GammaSRGB
        fr = [gh.random_nv12(64, 16, seed=300 + 10 * lane + i) for i in range(3)]
        al = [np.random.default_rng(350 + 10 * lane + i).integers(0, 256, (16, 64), dtype=np.uint8) for i in range(3)]
        for i in range(3):
            ring.pixelBuffer(i).upload_planes(*fr[i])
            ab = ring.alphaPixelBuffer(i)
            ring.ctx._upload(ab.y_ptr, ab.y_stride, al[i], None)
            ring.ctx._sync(None)
        assert rs.decode(waitUntilCompleted=True)
        for i in range(3):
            got = ring.ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(8, 32 * 4)
            assert np.array_equal(got, oracle.decode_nv12_half(1, fr[i][0], fr[i][1], alpha=al[i])), (lane, i)
    rs.release()


def test_ring_set_rgba16f_targets(gh, oracle):
    """A ring set whose lanes render into RGBA16Float targets (bt709hip_ring_options.format through bt709hip_ringset_create): an
    alpha decoder on every visible device, every texel of every lane against the oracle (linear-light halves, alpha included)."""
    devices = lane_sets()[0]
    w, h, n = 192, 24, 4
    rs = mb.FrameRingSet(devices, (w, h), n, hasAlphaChannel=True, tries=1, pixelFormat=mb.MTLPixelFormatRGBA16Float)
    assert rs.handle, rs.lastStatus
    for lane, ring in enumerate(rs.lanes):
        fr = [gh.random_nv12(w, h, seed=400 + 10 * lane + i) for i in range(n)]
        al = [np.random.default_rng(450 + 10 * lane + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
        for i in range(n):
            ring.pixelBuffer(i).upload_planes(*fr[i])
            ab = ring.alphaPixelBuffer(i)
            ring.ctx._upload(ab.y_ptr, ab.y_stride, al[i], None)
        assert rs.decode(waitUntilCompleted=True)
        for i in range(n):
            tex = ring.texture(i)
            assert tex.bytesPerPixel == 8 and tex.stride == w * 8
            got = ring.ctx.getBGRATexturePixels(tex).view(np.uint16)
            assert np.array_equal(got, oracle.decode_nv12_rgba16f(1, fr[i][0], fr[i][1], alpha=al[i]).view(np.uint16)), (lane, i)
    rs.release()


@pytest.mark.parametrize("double", [False, True])
def test_frame_sharder_on_every_visible_device(gh, oracle, double):
    """bt709hip_shard_*: a lane (context + decoder + in-flight pool) on every visible device -- bind(ctx) / hipSetDevice across
    devices, the whole point of the sharder -- frame i -> lane i mod n, every frame byte-compared with the oracle."""
    devices = lane_sets()[1 if double else 0]
    w, h, depth = 320, 64, 2
    sh = mb.FrameSharder(devices, (w, h), gamma=mb.MetalBT709GammaApple, depth=depth)
    assert sh.handle, sh.lastStatus
    assert [sh.laneDevice(i) for i in range(len(devices))] == devices
    frames = [gh.random_nv12(w, h, seed=8000 + i) for i in range(6 * len(devices))]
    window = len(devices) * depth
    for i, (y, c) in enumerate(frames):
        assert sh.submit(y, c) == i, sh.lastStatus
        if i >= window - 1:
            j = i - (window - 1)
            assert np.array_equal(sh.wait(j), oracle.decode_nv12(0, *frames[j])), j
    for j in range(len(frames) - (window - 1), len(frames)):
        assert np.array_equal(sh.wait(j), oracle.decode_nv12(0, *frames[j])), j
    sh.release()


def test_cpp_selftest_on_every_visible_device(gh, vectors, tmp_path):
    """host/decoder_selftest.cpp --devices all | N: the C++ mirror's FrameRingSet over every visible device (and over three
    lanes that wrap), the 28 reference vectors on every lane, lane l holding them rotated by l."""
    from test_host_cpu import build_cpp_selftest
    exe = build_cpp_selftest(tmp_path)
    args = []
    for r in vectors["metal_decode"]:
        args += [str(v) for v in r["ycbcr"] + r["rgb_out"]]
    for spec, lanes in (("all", visible()), ("3", 3)):
        res = subprocess.run([exe, "--devices", spec] + args, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr
        assert "%d lanes on %d visible device(s), 28 ring frames each, 0 failures" % (lanes, visible()) in res.stdout
        assert sum(1 for line in res.stdout.splitlines() if line.startswith("lane ")) == lanes


def test_ring_hunt_under_a_budget(gh, oracle):
    """bt709hip_ring_create_ex: the placement hunt under a byte budget and a time budget.  frugal = the incumbent pair + one
    candidate pair: the peak footprint stays under twice the ring and slabs are freed on the way; a 1 ms time budget ends the
    hunt after its first probe; a byte budget that cannot hold a second output slab means no hunt.  The ring kept decodes to the
    oracle's bytes every time, and the report says what happened."""
    ctx = gh.context()
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    n, w, h = 64, 1920, 1080
    ring_bytes = n * (w * h * 3 // 2 + w * h * 4)
    tiles = [gh.random_nv12(240, 8, seed=43000 + i) for i in range(n)]

    def check(ring):
        for i in (0, 17, n - 1):
            ty, tc = tiles[i]
            ring.pixelBuffer(i).upload_planes(np.tile(ty, (h // 8, w // 240)), np.tile(tc, (h // 8, w // 240)))
        assert ring.decode(waitUntilCompleted=True)
        for i in (0, 17, n - 1):
            got = ctx.getBGRATexturePixels(ring.texture(i)).view(np.uint8).reshape(h, w * 4)
            assert np.array_equal(got, np.tile(oracle.decode_nv12(0, *tiles[i]), (h // 8, w // 240))), i

    free0 = C.c_size_t()
    _capi.check(ctx.lib.bt709hip_mem_info(ctx.handle, C.byref(free0), None))
    frugal = mb.FrameRing(dec, (w, h), n, tries=3, frugal=True)
    p = frugal.placement()
    assert p.tries == 3 and p.hunt_ms > 0 and p.budget_bytes <= 2 * ring_bytes + (1 << 20)
    assert ring_bytes <= p.peak_bytes <= p.budget_bytes
    assert p.in_candidates <= 2 and p.out_candidates >= 3 and p.evicted == p.out_candidates - 2 and p.stopped_by == 1
    assert 0 <= p.chosen_out < p.out_candidates and p.out_prescan_GBps[p.chosen_out] > 100.0
    check(frugal)
    frugal.release()
    default = mb.FrameRing(dec, (w, h), n, tries=3)  # round 6: the default IS the frugal hunt (twice the ring)
    p = default.placement()
    assert abs(p.budget_bytes - 2 * ring_bytes) < (1 << 20) and ring_bytes < p.peak_bytes <= p.budget_bytes
    assert p.in_candidates <= 2 and p.stopped_by == 1 and p.evicted == p.out_candidates - 2 and p.out_candidates >= 3
    check(default)
    default.release()
    wide = mb.FrameRing(dec, (w, h), n, tries=3, maxBytes=4 * ring_bytes)  # round 5's default, by name
    p = wide.placement()
    assert abs(p.budget_bytes - 4 * ring_bytes) < (1 << 20) and ring_bytes < p.peak_bytes <= p.budget_bytes
    assert p.stopped_by in (0, 1) and p.evicted >= 0 and p.out_candidates >= 3
    check(wide)
    wide.release()
    roomy = mb.FrameRing(dec, (w, h), n, tries=3, maxBytes=free0.value // 2)  # round 4's behaviour: every candidate stays alive
    q = roomy.placement()
    assert q.evicted == 0 and q.stopped_by == 0 and q.peak_bytes >= ring_bytes + 2 * n * w * h * 4  # at least three outputs alive at once
    check(roomy)
    roomy.release()
    hurried = mb.FrameRing(dec, (w, h), n, tries=3, maxMilliseconds=1)
    p = hurried.placement()
    assert p.stopped_by == 2 and p.out_candidates == 1 and (p.chosen_in, p.chosen_out) == (0, 0) and p.hunt_ms < 2000
    check(hurried)
    hurried.release()
    tight = mb.FrameRing(dec, (w, h), n, tries=3, maxBytes=ring_bytes + (64 << 20))
    p = tight.placement()
    assert p.tries == 1 and p.stopped_by == 1 and p.probes == 0 and p.hunt_ms == 0.0
    check(tight)
    tight.release()
    free1 = C.c_size_t()
    _capi.check(ctx.lib.bt709hip_mem_info(ctx.handle, C.byref(free1), None))
    assert abs(free1.value - free0.value) < (64 << 20)  # nothing leaked by any of the four hunts


def _ring_of(gh, dec, w, h, n, seed):
    ring = mb.FrameRing(dec, (w, h), n, tries=1)
    frames = [gh.random_nv12(w, h, seed=seed + i) for i in range(n)]
    for i, f in enumerate(frames):
        ring.pixelBuffer(i).upload_planes(*f)
    return ring, frames


@pytest.mark.parametrize("second_coalesces", [False, True])
def test_two_decoders_on_one_stream_keep_their_submission_order(gh, oracle, second_coalesces):
    """Decoder A (coalescing submit on) queues frames that write surface S on stream s; decoder B then decodes OTHER frames into
    the same surfaces on the same stream.  B was submitted later, so S must hold B's pixels: B's call issues A's queue first
    (round 4's shim launched A's queued frames AFTER B's -- a write-after-write inversion)."""
    ctx = gh.context()
    w, h, n = 640, 16, 3
    a = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 8})
    b = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 8} if second_coalesces else None)
    ring_a, frames_a = _ring_of(gh, a, w, h, n, 61000)
    ring_b, frames_b = _ring_of(gh, b, w, h, n, 62000)
    cb = ctx.commandQueue.commandBuffer(new_stream=True)
    for i in range(n):  # A: queued (validated, not launched)
        assert a.decodeBT709(ring_a.pixelBuffer(i), None, ring_a.texture(i), commandBuffer=cb, renderWidth=w, renderHeight=h)
    assert ctx.lib.bt709hip_last_kernel_name() == b"(queued: coalescing submit)"
    for i in range(n):  # B: into A's surfaces, same stream
        assert b.decodeBT709(ring_b.pixelBuffer(i), None, ring_a.texture(i), commandBuffer=cb, renderWidth=w, renderHeight=h)
    for i in range(n):
        got = ctx.getBGRATexturePixels(ring_a.texture(i), commandBuffer=cb).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, *frames_b[i])), i
    cb.release()
    ring_a.release()
    ring_b.release()


def test_coalescing_queue_age_limit(gh, oracle):
    """BT709HIP_OPT_COALESCE_MAX_AGE_US: frames queued on stream s, then the caller goes quiet on s.  Without the option they stay
    queued until something touches s; with it (1 ms) the context's next call on ANY stream issues them."""
    ctx = gh.context()
    lib = ctx.lib
    w, h, n = 640, 16, 2
    for age_us in (0, 1000):
        dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 32, _capi.OPT_COALESCE_MAX_AGE_US: age_us})
        got_age = C.c_int(-1)
        _capi.check(lib.bt709hip_decoder_get_option(dec._handle, _capi.OPT_COALESCE_MAX_AGE_US, C.byref(got_age)))
        assert got_age.value == age_us
        ring, frames = _ring_of(gh, dec, w, h, n, 63000)
        _capi.check(lib.bt709hip_memset(ctx.handle, ring.texture(0).ptr, 0x5A, n * w * h * 4, None))
        ctx._sync(None)
        s1, other = ctx.commandQueue.commandBuffer(new_stream=True), ctx.commandQueue.commandBuffer(new_stream=True)
        for i in range(n):
            assert dec.decodeBT709(ring.pixelBuffer(i), None, ring.texture(i), commandBuffer=s1, renderWidth=w, renderHeight=h)
        time.sleep(0.02)
        _capi.check(lib.bt709hip_stream_synchronize(ctx.handle, other.stream))  # a call on ANOTHER stream of the context
        time.sleep(0.05)  # an aged queue has been launched on s1 by now and has long finished (two 640 x 16 frames)
        got = ctx.getBGRATexturePixels(ring.texture(n - 1), commandBuffer=other).view(np.uint8).reshape(h, w * 4)
        if age_us:
            assert np.array_equal(got, oracle.decode_nv12(0, *frames[n - 1]))
        else:
            assert (got == 0x5A).all()  # still "encoded, not committed"
        assert dec.flush(commandBuffer=s1)
        got = ctx.getBGRATexturePixels(ring.texture(0), commandBuffer=s1).view(np.uint8).reshape(h, w * 4)
        assert np.array_equal(got, oracle.decode_nv12(0, *frames[0]))
        s1.release()
        other.release()
        ring.release()


def test_hunt_probes_do_not_queue_on_a_coalescing_decoder(gh, oracle):
    """A ring shorter than the decoder's coalescing count whose slabs are large enough to hunt (8 x 4K = 365 MB): the probes
    time the ring's OWN launch, so the hunt turns the option off while it runs and puts it back."""
    dec = gh.make_decoder(mb.MetalBT709GammaApple, options={_capi.OPT_COALESCE: 32})
    ring = mb.FrameRing(dec, (3840, 2160), 8, tries=2)
    p = ring.placement()
    assert p.tries == 2 and p.probes >= 1 and p.first_GBps > 500.0, (p.probes, p.first_GBps)
    v = C.c_int()
    _capi.check(ring.lib.bt709hip_decoder_get_option(dec._handle, _capi.OPT_COALESCE, C.byref(v)))
    assert v.value == 32
    ring.release()


def _bench(args, timeout=900):
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[0]) if len(lines) == 1 else None)


@pytest.mark.parametrize("launcher", ["processes", "threads"])
def test_bench_n_greater_than_one_on_the_gpu(gh, launcher):
    """bench.py --gpus N end to end on hardware, both launchers: N = twice the visible devices with --allow-shared-devices (on a
    one-GPU box two ranks / lanes on it; on the 8-GPU node sixteen over eight) -- refused without the flag -- and, when more
    than one device is visible, N = the device count without it.  The line counts the distinct devices, spot-checks every
    rank's own ring and names every rank's device."""
    ndev = visible()
    common = ["--ring", "48", "--steps", "3", "--warmup", "1", "--placement-tries", "1", "--no-cpu-baseline", "--launcher", launcher]
    r, d = _bench(["--gpus", str(2 * ndev)] + common)
    assert r.returncode == 2 and d is None and "--allow-shared-devices" in r.stderr
    r, d = _bench(["--gpus", str(2 * ndev), "--allow-shared-devices"] + common)
    assert r.returncode == 0 and d is not None, r.stderr[-2000:]
    assert d["n_gpus"] == ndev and d["ranks"] == 2 * ndev and d["shared_devices"] is True
    assert d["parity_spot_check"] == "ok" and d["parity_spot_check_ranks"] == 2 * ndev and d["value"] > 100.0
    assert [x["rank"] for x in d["config"]["devices"]] == list(range(2 * ndev))
    assert len({x["pci_bus_id"] for x in d["config"]["devices"]}) == ndev
    assert all(p["parity_spot_check"] == "ok" and p["avg_launch_us"] > 0 for p in d["per_rank"])
    assert ("threads (ONE process" in d["config"]["launcher"]) == (launcher == "threads")
    if ndev > 1:
        r, d = _bench(["--gpus", str(ndev)] + common)
        assert r.returncode == 0 and d is not None, r.stderr[-2000:]
        assert d["n_gpus"] == ndev and d["ranks"] == ndev and d["shared_devices"] is False and d["parity_spot_check_ranks"] == ndev
