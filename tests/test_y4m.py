"""Y4M (YUV4MPEG2 C420jpeg) reader/writer and the GPU planar<->NV12 chroma shuffles
(format = Renderer/y4m_writer.h:61-241)."""
import numpy as np
import pytest

import metalbt709decoder_amd as mb
from metalbt709decoder_amd import y4m


def _frame(w, h, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8),
            rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8))


def test_writer_emits_the_reference_byte_layout(tmp_path):
    p = tmp_path / "a.y4m"
    y, u, v = _frame(6, 4, 1)
    with y4m.Y4MWriter(str(p), 6, 4, fps=29.97) as w:
        w.write_frame(y, u, v)
        w.write_frame(y, u, v)
    raw = p.read_bytes()
    header = b"YUV4MPEG2 W6 H4 F30000:1001 Ip A1:1 C420jpeg\nXYSCSS=420JPEG\n"  # y4m_writer.h:61-183
    assert raw.startswith(header)
    body = raw[len(header):]
    one = b"FRAME\n" + y.tobytes() + u.tobytes() + v.tobytes()                      # :194-241
    assert body == one + one
    assert {f: s for f, s in y4m.FPS.items()} == {1: "1:1", 15: "15:1", 24: "24:1", 25: "25:1", 29.97: "30000:1001",
                                                  30: "30:1", 60: "60:1"}           # Y4MHeaderFPS :22-30, 98-139
    with pytest.raises(ValueError):
        y4m.Y4MWriter(str(p), 5, 4)
    with pytest.raises(ValueError):
        y4m.Y4MWriter(str(p), 6, 4, fps=50)


def test_reader_round_trip_and_tolerance(tmp_path):
    p = tmp_path / "b.y4m"
    frames = [_frame(16, 8, s) for s in range(3)]
    with y4m.Y4MWriter(str(p), 16, 8, fps=60) as w:
        for f in frames:
            w.write_frame(*f)
    with y4m.Y4MReader(str(p)) as r:
        assert (r.width, r.height, r.fps) == (16, 8, (60, 1))
        got = list(r)
    assert len(got) == 3
    for a, b in zip(frames, got):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # a file from another writer: no XYSCSS line, extra tags
    q = tmp_path / "c.y4m"
    y, u, v = frames[0]
    q.write_bytes(b"YUV4MPEG2 W16 H8 F25:1 Ip A1:1 C420mpeg2 XFOO=bar\nFRAME\n" + y.tobytes() + u.tobytes() + v.tobytes())
    with y4m.Y4MReader(str(q)) as r:
        (y2, u2, v2), = list(r)
    assert np.array_equal(y2, y) and np.array_equal(u2, u) and np.array_equal(v2, v)
    bad = tmp_path / "d.y4m"
    bad.write_bytes(b"YUV4MPEG2 W16 H8 F25:1 C444\n")
    with pytest.raises(ValueError):
        y4m.Y4MReader(str(bad))
    trunc = tmp_path / "e.y4m"
    trunc.write_bytes(b"YUV4MPEG2 W16 H8 F25:1 C420jpeg\nFRAME\n" + b"\0" * 10)
    with pytest.raises(ValueError):
        list(y4m.Y4MReader(str(trunc)))


def test_write_nv12_host_planes(tmp_path):
    p = tmp_path / "n.y4m"
    y, u, v = _frame(8, 4, 7)
    c = np.empty((2, 8), np.uint8)
    c[:, 0::2], c[:, 1::2] = u, v
    with y4m.Y4MWriter(str(p), 8, 4) as w:
        w.write_nv12(y, c)
    (y2, u2, v2), = list(y4m.Y4MReader(str(p)))
    assert np.array_equal(y2, y) and np.array_equal(u2, u) and np.array_equal(v2, v)


def test_bench_takes_a_clip_on_the_dry_run_path(tmp_path):
    """bench.py --y4m FILE without a GPU (--dry-run): the workload is the clip -- geometry from its header, the ring sized like
    the headline's (256 x 4K worth of pixels, a multiple of 8 frames), `data` says "y4m", config.workload names the file's sha256,
    and the memory plan counts the clip's frames held in host memory."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clip = tmp_path / "tiny.y4m"
    with y4m.Y4MWriter(str(clip), 640, 360, fps=30) as w:
        for i in range(5):
            w.write_frame(*_frame(640, 360, 70 + i))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--y4m", str(clip), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.splitlines()[0])
    ring = 256 * 3840 * 2160 // (640 * 360) // 8 * 8
    assert d["data"] == "y4m" and d["metric"].startswith("Gpixel/s, y4m")
    assert "640x360" in d["config"]["workload"] and "a ring of %d frames = the 5 frames of the YUV4MPEG2" % ring in d["config"]["workload"]
    assert hashlib.sha256(clip.read_bytes()).hexdigest() in d["config"]["workload"]
    plan = d["config"]["memory_plan"]
    assert plan["ring_bytes"] == ring * ((640 * 360 * 3 // 2 + 255) // 256 * 256 + 640 * 360 * 4) and plan["host_bytes_per_rank"] > 5 * 640 * 360 * 3 // 2
    # not a YUV4MPEG2 file: refused before anything is allocated
    bad = tmp_path / "bad.y4m"
    bad.write_bytes(b"RIFF....")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--y4m", str(bad)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "YUV4MPEG2" in r.stderr


# ------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    gpu_helpers.context()
    return gpu_helpers


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(2, 2), (16, 4), (18, 6), (64, 32), (1920, 1080), (3840, 2160)])
def test_gpu_interleave_and_back(gh, size):
    w, h = size
    y, u, v = _frame(w, h, w + h)
    buf = y4m.i420_to_pixel_buffer(gh.context(), y, u, v)
    y2, c2 = buf.download_planes()
    assert np.array_equal(y2, y)
    assert np.array_equal(c2[:, 0::2], u) and np.array_equal(c2[:, 1::2], v)
    y3, u3, v3 = y4m.pixel_buffer_to_i420(buf)
    assert np.array_equal(y3, y) and np.array_equal(u3, u) and np.array_equal(v3, v)


@pytest.mark.gpu
def test_y4m_file_through_the_decoder(gh, oracle, tmp_path):
    """File -> planar frame -> GPU interleave -> decode == oracle on the same NV12."""
    ctx = gh.context()
    w, h = 64, 16
    y, u, v = _frame(w, h, 99)
    p = tmp_path / "clip.y4m"
    with y4m.Y4MWriter(str(p), w, h) as wr:
        wr.write_frame(y, u, v)
    (yr, ur, vr), = list(y4m.Y4MReader(str(p)))
    buf = y4m.i420_to_pixel_buffer(ctx, yr, ur, vr)
    tex = ctx.makeBGRATexture((w, h))
    dec = gh.make_decoder(mb.MetalBT709GammaApple)
    assert dec.decodeBT709(buf, None, tex, None, None, w, h, True)
    got = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(h, w * 4)
    c = np.empty((h // 2, w), np.uint8)
    c[:, 0::2], c[:, 1::2] = u, v
    assert np.array_equal(got, oracle.decode_nv12(0, y, c))
    # and back out to a file from the device buffer
    q = tmp_path / "out.y4m"
    with y4m.Y4MWriter(str(q), w, h) as wr:
        wr.write_pixel_buffer(buf)
    assert q.read_bytes() == p.read_bytes()


@pytest.mark.gpu
def test_bench_path_decodes_a_generated_clip(gh, tmp_path):
    """bench.py --y4m (round 6): a 64-frame 1080p clip written by tools/make_y4m_clip.py -- the bundled QuickTime pattern panned,
    through the GPU ENCODER, in the reference's file layout (Renderer/y4m_writer.h:61-241) -- fills the bench's ring (planar
    chroma interleaved on the device, frames repeated to the ring length) and is decoded by the same launch the headline times;
    bench.py byte-compares 48 rows of one ring frame per XCD band with the oracle's decode of the CLIP'S OWN planes and
    reports it in the line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from make_y4m_clip import make_clip
    clip = tmp_path / "qt_pattern_1080p.y4m"
    assert make_clip(str(clip), frames=64, fps=60, ctx=gh.context()) == (1920, 1080)
    with y4m.Y4MReader(str(clip)) as r:
        assert (r.width, r.height, r.fps) == (1920, 1080, (60, 1))
        frames = list(r)
    assert len(frames) == 64 and not np.array_equal(frames[0][0], frames[1][0])  # the pan moves the picture
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--y4m", str(clip), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--placement-tries", "1"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout)
    assert d["data"] == "y4m" and d["value"] and d["parity_spot_check"] == "ok" and len(d["parity_spot_frames"]) == 8
    assert "1920x1080" in d["config"]["workload"] and "a ring of 1024 frames = the 64 frames of the YUV4MPEG2" in d["config"]["workload"]
    import hashlib
    assert hashlib.sha256(clip.read_bytes()).hexdigest() in d["config"]["workload"]
    assert d["roofline"]["kernel"].startswith("decode_nv12") and d["roofline"]["frac"] > 0.3
