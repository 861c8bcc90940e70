"""CPU tests of the host shim on a FAKE HIP runtime (tests/native/fake_hip/): GPU sanitizers do not exist on this pool, so
the code that holds mutexes, thread-locals and per-stream queues -- csrc/shim_*.cpp, csrc/bt709_ring.cpp -- is compiled with
g++ against a stand-in <hip/hip_runtime.h> (streams = FIFOs with worker threads, launches = log entries, device memory = host
memory / address reservations) and

  * run under ASan + UBSan + LeakSanitizer and under TSan (tests/native/shim_stress.cpp: coalescing from two threads on two streams
    with a third flipping options, pool / sharder churn, ring hunts with refused allocations and failing launches, ring sets,
    graphs, decoder destruction with frames queued);
  * checked STRUCTURALLY: every export of include/bt709hip.h / bt709hip_ext.h that takes a `void *stream` must issue the frames a coalescing
    decoder has queued on that stream before its own work -- the header is parsed, so an export added later without that
    property (or without an entry below) fails here.

Nothing of this is shipped or loaded by the product; the product library (libbt709hip.so) is not involved at all.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "metalbt709decoder_amd", "csrc")
FAKE = os.path.join(HERE, "native", "fake_hip")
SHIM_SOURCES = [os.path.join(CSRC, f) for f in ("shim_core.cpp", "shim_decode.cpp", "shim_convert.cpp", "shim_coalesce.cpp", "shim_pool_shard.cpp",
                                                 "shim_introspect.cpp", "bt709_ring.cpp", "transfer_tables.cpp")] + [os.path.join(FAKE, "fake_hip.cpp")]
CXX = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-Wall", "-Wno-format-truncation", "-I" + FAKE, "-I" + os.path.join(HERE, "native")]

sys.path.insert(0, ROOT)
from metalbt709decoder_amd import _capi  # noqa: E402  (signatures only: the product library is never loaded here)


def build(out, extra, sources):
    r = subprocess.run(CXX + extra + sources + ["-o", out, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


@pytest.mark.parametrize("leg", ["asan", "tsan"])
def test_shim_stress_under_sanitizers(tmp_path, leg):
    """tests/native/shim_stress.cpp, sanitizer clean.  (tools/sanitize.sh runs the same two builds.)"""
    flags = {"asan": ["-fsanitize=address,undefined"], "tsan": ["-fsanitize=thread"]}[leg]
    exe = build(str(tmp_path / ("shim_stress_" + leg)), flags, SHIM_SOURCES + [os.path.join(HERE, "native", "shim_stress.cpp")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    assert "ok: shim stress on the fake HIP runtime, 0 failures" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr


class FakeOp(C.Structure):
    _fields_ = [("seq", C.c_uint64), ("stream", C.c_void_p), ("device", C.c_int32), ("frames", C.c_int32), ("op", C.c_char * 48),
                ("first_in", C.c_void_p), ("first_out", C.c_void_p)]


@pytest.fixture(scope="module")
def fake(tmp_path_factory):
    """The shim built against the fake runtime as a shared library, bound with the product's own ctypes signatures."""
    so = build(str(tmp_path_factory.mktemp("fake") / "libbt709hip_fake.so"), ["-shared", "-fPIC"], SHIM_SOURCES)
    lib = C.CDLL(so)
    for name, (res, args) in _capi.SYMBOLS.items():
        fn = getattr(lib, name)  # the fake build exports every symbol the header declares
        fn.restype, fn.argtypes = res, args
    assert lib.bt709hip_abi_version() == _capi.ABI_VERSION
    lib.fake_hip_log_size.restype = C.c_uint64
    lib.fake_hip_log_get.argtypes = [C.c_uint64, C.POINTER(FakeOp)]
    lib.fake_hip_set_device_count(2)
    return lib


def log(lib, start=0):
    out, op, i = [], FakeOp(), start
    while lib.fake_hip_log_get(i, C.byref(op)) == 0:
        out.append((op.op.decode(), op.stream, op.frames, op.first_out))
        i += 1
    return out


def stream_exports():
    """Names of the C-ABI exports with a `void *stream` parameter, parsed from the public header."""
    import abi_headers
    hdr = abi_headers.text()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    names = []
    for m in re.finditer(r"\b(?:int|const char \*|bt709hip_\w+ \*)\s*(bt709hip_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        if re.search(r"void\s*\*\s*stream\b", m.group(2)):
            names.append(m.group(1))
    return sorted(set(names))


def test_every_stream_taking_export_issues_the_queued_frames_first(fake):
    """For EVERY export with a `void *stream` parameter: decoder A (coalescing on) has one frame queued on stream S -- validated,
    not launched; then the export is called on S.  A's frame must be launched on S before anything the export itself puts there
    (and must have been launched at all when the export returns).  The list of exports comes from the header."""
    lib = fake
    lib.fake_hip_reset()
    ctx = C.c_void_p()
    assert lib.bt709hip_context_create(0, C.byref(ctx)) == 0
    A, B = C.c_void_p(), C.c_void_p()
    assert lib.bt709hip_decoder_create(ctx, 0, 0, C.byref(A)) == 0 and lib.bt709hip_decoder_create(ctx, 0, 0, C.byref(B)) == 0
    assert lib.bt709hip_decoder_set_option(A, _capi.OPT_COALESCE, 8) == 0
    assert lib.bt709hip_decoder_setup(B) == 0 and lib.bt709hip_render_scaled_prepare(ctx) == 0 and lib.bt709hip_encoder_prepare(ctx, 1, 0) == 0
    w, h = 64, 16
    mem = {}
    for k, n in (("in_a", w * h * 3 // 2), ("out_a", w * h * 4), ("in_b", 4 * w * h * 3 // 2), ("out_b", 4 * w * h * 4), ("aux", 1 << 16)):
        p = C.c_void_p()
        assert lib.bt709hip_malloc(ctx, n, C.byref(p)) == 0
        mem[k] = p.value
    host = (C.c_uint8 * (1 << 16))()

    def frame(base, i=0, ww=w, hh=h):
        b = base + i * ww * hh * 3 // 2
        return _capi.Frame(b, ww, b + ww * hh, ww, ww, hh, 1, 1)

    def surf(base, i=0, ww=w, hh=h, fmt=0):
        return _capi.Surface(base + i * ww * hh * 4, ww * 4, ww, hh, fmt, 0)

    fa, sa = frame(mem["in_a"]), surf(mem["out_a"])
    fb = (_capi.Frame * 4)(*[frame(mem["in_b"], i) for i in range(4)])
    sb = (_capi.Surface * 4)(*[surf(mem["out_b"], i) for i in range(4)])
    half = (_capi.Surface * 4)(*[surf(mem["out_b"], i, w // 2, h // 2) for i in range(4)])
    ring = C.c_void_p()
    assert lib.bt709hip_ring_create(B, w, h, 4, 0, 1, C.byref(ring)) == 0
    ev = C.c_void_p()
    assert lib.bt709hip_event_create(ctx, C.byref(ev)) == 0

    def new_stream():
        s = C.c_void_p()
        assert lib.bt709hip_stream_create(ctx, C.byref(s)) == 0
        return s.value

    def empty_graph():  # recorded on a stream of its own
        s, g = new_stream(), C.c_void_p()
        assert lib.bt709hip_graph_begin_capture(ctx, s) == 0 and lib.bt709hip_memset(ctx, mem["aux"], 0, 64, s) == 0
        assert lib.bt709hip_graph_end_capture(ctx, s, C.byref(g)) == 0 and lib.bt709hip_stream_destroy(ctx, s) == 0
        return g

    u, v = mem["aux"], mem["aux"] + 4096
    # name -> call on stream S (returns the status); "after" hooks undo what the call started
    recipes = {
        "bt709hip_stream_destroy": lambda S: lib.bt709hip_stream_destroy(ctx, S),
        "bt709hip_stream_synchronize": lambda S: lib.bt709hip_stream_synchronize(ctx, S),
        "bt709hip_event_record": lambda S: lib.bt709hip_event_record(ctx, ev, S),
        "bt709hip_stream_wait_event": lambda S: lib.bt709hip_stream_wait_event(ctx, S, ev),
        "bt709hip_graph_begin_capture": lambda S: lib.bt709hip_graph_begin_capture(ctx, S),
        "bt709hip_graph_end_capture": None,  # special: the frame is queued DURING the recording, see below
        "bt709hip_graph_launch": lambda S: lib.bt709hip_graph_launch(ctx, empty_graph(), S),
        "bt709hip_memset": lambda S: lib.bt709hip_memset(ctx, mem["aux"], 0, 256, S),
        "bt709hip_upload": lambda S: lib.bt709hip_upload(ctx, mem["aux"], 256, host, 256, 256, 4, S),
        "bt709hip_download": lambda S: lib.bt709hip_download(ctx, host, 256, mem["aux"], 256, 256, 4, S),
        "bt709hip_decode": lambda S: lib.bt709hip_decode(B, fb, None, sb, w, h, S, 0),
        "bt709hip_decoder_flush": lambda S: lib.bt709hip_decoder_flush(A, S),
        "bt709hip_decode_batch": lambda S: lib.bt709hip_decode_batch(B, 4, fb, None, sb, S, 0),
        "bt709hip_unconvert": lambda S: lib.bt709hip_unconvert(B, mem["in_b"], w * 4, w, h // 2, C.byref(surf(mem["out_b"], 0, w, h // 2)), S, 0),
        "bt709hip_unconvert_batch": lambda S: lib.bt709hip_unconvert_batch(B, 2, (C.c_void_p * 2)(mem["in_b"], mem["in_b"] + w * (h // 2) * 4), w * 4, w, h // 2,
                                                                            (_capi.Surface * 2)(surf(mem["out_b"], 0, w, h // 2), surf(mem["out_b"], 1, w, h // 2)), S, 0),
        "bt709hip_decode_half": lambda S: lib.bt709hip_decode_half(B, fb, None, half, S, 0),
        "bt709hip_decode_half_batch": lambda S: lib.bt709hip_decode_half_batch(B, 4, fb, None, half, S, 0),
        "bt709hip_decode_scaled": lambda S: lib.bt709hip_decode_scaled(B, fb, None, half, S, 0),
        "bt709hip_decode_scaled_batch": lambda S: lib.bt709hip_decode_scaled_batch(B, 4, fb, None, half, S, 0),
        "bt709hip_render_scaled": lambda S: lib.bt709hip_render_scaled(ctx, sb, half, S, 0),
        "bt709hip_render_scaled_batch": lambda S: lib.bt709hip_render_scaled_batch(ctx, 4, sb, half, S, 0),
        "bt709hip_ring_decode": lambda S: lib.bt709hip_ring_decode(ring, 0, 4, S, 0),
        "bt709hip_encode": lambda S: lib.bt709hip_encode(ctx, sb, fb, 1, 0, S, 0),
        "bt709hip_encode_batch": lambda S: lib.bt709hip_encode_batch(ctx, 4, sb, fb, 1, 0, S, 0),
        "bt709hip_interleave_cbcr": lambda S: lib.bt709hip_interleave_cbcr(ctx, u, 32, v, 32, mem["aux"] + 8192, 64, 32, 8, S, 0),
        "bt709hip_deinterleave_cbcr": lambda S: lib.bt709hip_deinterleave_cbcr(ctx, mem["aux"] + 8192, 64, u, 32, v, 32, 32, 8, S, 0),
        "bt709hip_copy_probe": lambda S: lib.bt709hip_copy_probe(ctx, mem["aux"] + 8192, mem["aux"], 4096, S),
    }
    exports = stream_exports()
    assert len(exports) >= 27, exports
    assert sorted(recipes) == exports, "the headers and this test disagree about the stream-taking exports: %s" % sorted(set(recipes) ^ set(exports))

    for name in exports:
        S = new_stream()
        if name == "bt709hip_stream_wait_event":  # the event it waits for must have been recorded (on another stream)
            other = new_stream()
            assert lib.bt709hip_event_record(ctx, ev, other) == 0
        mark = lib.fake_hip_log_size()
        if name == "bt709hip_graph_end_capture":
            g = C.c_void_p()
            assert lib.bt709hip_graph_begin_capture(ctx, S) == 0
            assert lib.bt709hip_decode(A, C.byref(fa), None, C.byref(sa), w, h, S, 0) == 0  # queued while S records
            assert lib.bt709hip_graph_end_capture(ctx, S, C.byref(g)) == 0                 # ... and recorded by the end of it
            assert log(lib, mark) == []                                                     # nothing ran: it is IN the graph
            assert lib.bt709hip_graph_launch(ctx, g, S) == 0 and lib.bt709hip_stream_synchronize(ctx, S) == 0
            ops = [o for o in log(lib, mark) if o[1] == S]
            assert [o[0] for o in ops] == ["graph_launch", "kernel:decode_nv12_quads<nt>"] and ops[1][3] == mem["out_a"], (name, ops)
            assert lib.bt709hip_graph_destroy(ctx, g) == 0 and lib.bt709hip_stream_destroy(ctx, S) == 0
            continue
        assert lib.bt709hip_decode(A, C.byref(fa), None, C.byref(sa), w, h, S, 0) == 0  # validated and QUEUED
        assert log(lib, mark) == [], name
        rc = recipes[name](S)
        assert rc == 0, (name, rc)
        ops = [o for o in log(lib, mark) if o[1] == S]
        assert ops, "%s returned without issuing the frame queued on its stream" % name
        assert ops[0][0].startswith("kernel:decode_nv12") and ops[0][2] == 1 and ops[0][3] == mem["out_a"], \
            "%s put %r on the stream before the queued frame" % (name, ops[0])
        assert sum(1 for o in ops if o[3] == mem["out_a"] and o[0].startswith("kernel:decode")) == 1, (name, ops)
        if name == "bt709hip_graph_begin_capture":
            g = C.c_void_p()
            assert lib.bt709hip_graph_end_capture(ctx, S, C.byref(g)) == 0 and lib.bt709hip_graph_destroy(ctx, g) == 0
        if name != "bt709hip_stream_destroy":
            assert lib.bt709hip_stream_synchronize(ctx, S) == 0 and lib.bt709hip_stream_destroy(ctx, S) == 0
    assert lib.bt709hip_ring_destroy(ring) == 0 and lib.bt709hip_event_destroy(ctx, ev) == 0
    assert lib.bt709hip_decoder_destroy(A) == 0 and lib.bt709hip_decoder_destroy(B) == 0


def test_fake_runtime_is_not_part_of_the_product():
    """The fake runtime lives under tests/ only: the package, the public header and the build script never mention it."""
    for path in [os.path.join(ROOT, "include", "bt709hip.h"), os.path.join(ROOT, "include", "bt709hip_ext.h"), os.path.join(ROOT, "metalbt709decoder_amd", "build.py"),
                 os.path.join(ROOT, "metalbt709decoder_amd", "_capi.py"), os.path.join(ROOT, "metalbt709decoder_amd", "decoder.py")]:
        assert "fake_hip" not in open(path).read(), path
    for f in os.listdir(CSRC):
        assert "fake_hip" not in open(os.path.join(CSRC, f)).read(), f
