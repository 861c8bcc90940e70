"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been
built, the reference's own headers (oracle/_ref/libbt709ref.so).

Test infrastructure only: imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from metalbt709decoder_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libbt709ref.so")

GAMMA_APPLE, GAMMA_SRGB, GAMMA_LINEAR, GAMMA_ITU709 = 0, 1, 2, 3
GAMMA_NAMES = {0: "apple", 1: "srgb", 2: "linear", 3: "itu709"}

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int)
_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)


def build_oracle(force=False):
    """Compile oracle/ (and oracle/_ref when /root/reference exists)."""
    src = os.path.join(ORACLE_DIR, "bt709_oracle.c")
    stale = (not os.path.exists(ORACLE_SO)
             or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src))
    if force or stale or (os.path.isdir("/root/reference") and not os.path.exists(REF_SO)):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class Oracle:
    def __init__(self):
        build_oracle()
        L = self.lib = C.CDLL(ORACLE_SO)
        for name in ("srgb_to_linear", "linear_to_srgb", "itu709_to_linear",
                     "linear_to_itu709", "apple196_to_linear", "linear_to_apple196"):
            f = getattr(L, "bt709o_" + name)
            f.restype, f.argtypes = C.c_float, [C.c_float]
        L.bt709o_quantize.restype, L.bt709o_quantize.argtypes = C.c_int, [C.c_float]
        L.bt709o_transfer_to_byte.restype = C.c_int
        L.bt709o_transfer_to_byte.argtypes = [C.c_int, C.c_float]
        L.bt709o_ycbcr_to_rgbn.argtypes = [C.c_int] * 3 + [_f32p]
        L.bt709o_decode_pixel.argtypes = [C.c_int] * 4 + [_i32p]
        L.bt709o_decode_alpha.restype, L.bt709o_decode_alpha.argtypes = C.c_int, [C.c_int]
        L.bt709o_encode_pixel.argtypes = [C.c_int] * 4 + [_i32p]
        L.bt709o_encode_linear_pixel.argtypes = [C.c_int] * 4 + [_i32p]
        L.bt709o_decode_to_linear_pixel.argtypes = [C.c_int] * 4 + [_i32p]
        L.bt709o_decode_nv12_rows.restype = C.c_int
        L.bt709o_decode_nv12_rows.argtypes = [
            C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t,
            C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t, C.c_int]
        L.bt709o_decode_nv12_half.restype = C.c_int
        L.bt709o_decode_nv12_half.argtypes = [
            C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int,
            _u8p, C.c_size_t, C.c_int]
        L.bt709o_unconvert_packed.restype = C.c_int
        L.bt709o_unconvert_packed.argtypes = [C.c_int, _u32p, _u32p, C.c_int, C.c_int]
        L.bt709o_convert_packed.restype = C.c_int
        L.bt709o_convert_packed.argtypes = [C.c_int, _u32p, _u32p, C.c_int, C.c_int]
        L.bt709o_packed_to_nv12.argtypes = [_u32p, C.c_int, C.c_int, _u8p, C.c_size_t,
                                            _u8p, C.c_size_t]
        L.bt709o_nv12_to_packed.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t,
                                            C.c_int, C.c_int, _u32p]
        L.bt709o_subsample_block.argtypes = [_i32p, C.c_int, C.c_int, _i32p, _i32p, _i32p]
        L.bt709o_encode_nv12.restype = C.c_int
        L.bt709o_encode_nv12.argtypes = [_u32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _u8p, C.c_size_t, _u8p, C.c_size_t]
        L.bt709o_decode_table.argtypes = [C.c_int, _u8p, C.c_int]
        L.bt709o_roundtrip_histogram.argtypes = [C.c_int, _u64p, C.c_int]
        L.bt709o_thresholds.argtypes = [C.c_int, _f32p]
        L.bt709o_check_thresholds.restype = C.c_uint64
        L.bt709o_check_thresholds.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int]

    # -- scalars
    def transfer_to_byte(self, gamma, v):
        return self.lib.bt709o_transfer_to_byte(gamma, float(v))

    def decode_pixel(self, gamma, Y, Cb, Cr):
        out = (C.c_int * 3)()
        self.lib.bt709o_decode_pixel(gamma, Y, Cb, Cr, out)
        return tuple(out)

    def encode_pixel(self, gamma, R, G, B):
        out = (C.c_int * 3)()
        self.lib.bt709o_encode_pixel(gamma, R, G, B, out)
        return tuple(out)

    def decode_alpha(self, A):
        return self.lib.bt709o_decode_alpha(A)

    def encode_linear_pixel(self, apply_curve, R, G, B):
        out = (C.c_int * 3)()
        self.lib.bt709o_encode_linear_pixel(apply_curve, R, G, B, out)
        return tuple(out)

    def decode_to_linear_pixel(self, apply_curve, Y, Cb, Cr):
        out = (C.c_int * 3)()
        self.lib.bt709o_decode_to_linear_pixel(apply_curve, Y, Cb, Cr, out)
        return tuple(out)

    def ycbcr_to_rgbn(self, Y, Cb, Cr):
        out = (C.c_float * 3)()
        self.lib.bt709o_ycbcr_to_rgbn(Y, Cb, Cr, out)
        return np.array(list(out), dtype=np.float32)

    # -- frames (numpy in, numpy out; tight or explicit strides)
    def decode_nv12(self, gamma, y, uv, alpha=None, alpha_fill=0xFF, rows=None, out=None):
        """y: (H, y_stride>=W) u8 view of a plane; uv: (H/2, uv_stride>=W) u8.
        Arrays must be C-contiguous 2-D; the row pitch is shape[1]; width is
        given by `y.shape[1]` unless the array carries a `.width` attribute via
        the (array, width) tuple form."""
        y, width = _plane(y)
        uv, _ = _plane(uv)
        height = y.shape[0]
        a = None
        if alpha is not None:
            a, _ = _plane(alpha)
        if out is None:
            out = np.zeros((height, width * 4), dtype=np.uint8)
        r0, r1 = rows if rows is not None else (0, height)
        if (height & 1) or (width & 1):
            return None
        rc = self.lib.bt709o_decode_nv12_rows(
            gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
            _ptr(a, _u8p) if a is not None else None, a.shape[1] if a is not None else 0,
            width, r0, r1, _ptr(out, _u8p), out.shape[1], alpha_fill)
        if rc != 0:
            return None
        return out

    def decode_nv12_half(self, gamma, y, uv, alpha_fill=0xFF, alpha=None):
        y, width = _plane(y)
        uv, _ = _plane(uv)
        a = _plane(alpha)[0] if alpha is not None else None
        height = y.shape[0]
        out = np.zeros((height // 2, (width // 2) * 4), dtype=np.uint8)
        rc = self.lib.bt709o_decode_nv12_half(
            gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
            _ptr(a, _u8p) if a is not None else None, a.shape[1] if a is not None else 0,
            width, height, _ptr(out, _u8p), out.shape[1], alpha_fill)
        return out if rc == 0 else None

    def decode_nv12_scaled(self, gamma, y, uv, out_width, out_height, alpha_fill=0xFF, alpha=None):
        y, width = _plane(y)
        uv, _ = _plane(uv)
        a = _plane(alpha)[0] if alpha is not None else None
        height = y.shape[0]
        out = np.zeros((out_height, out_width * 4), dtype=np.uint8)
        fn = self.lib.bt709o_decode_nv12_scaled
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p,
                       C.c_size_t, C.c_int, C.c_int, C.c_int]
        rc = fn(gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
                _ptr(a, _u8p) if a is not None else None, a.shape[1] if a is not None else 0, width, height,
                _ptr(out, _u8p), out.shape[1], out_width, out_height, alpha_fill)
        return out if rc == 0 else None

    def decode_nv12_rgba16f(self, gamma, y, uv, alpha=None):
        """-> (H, W, 4) float16 R,G,B,A: pass 1 into an RGBA16Float target."""
        y, width = _plane(y)
        uv, _ = _plane(uv)
        a = _plane(alpha)[0] if alpha is not None else None
        height = y.shape[0]
        out = np.zeros((height, width * 8), dtype=np.uint8)
        fn = self.lib.bt709o_decode_nv12_rgba16f
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t]
        rc = fn(gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
                _ptr(a, _u8p) if a is not None else None, a.shape[1] if a is not None else 0, width, height,
                _ptr(out, _u8p), out.shape[1])
        return out.view(np.float16).reshape(height, width, 4) if rc == 0 else None

    def render_scaled(self, src, out_width, out_height):
        """Pass 2 alone.  src: (H, W*4) uint8 BGRA8 sRGB rows, or (H, W, 4) float16 RGBA linear."""
        rgba16f = src.dtype == np.float16
        src = np.ascontiguousarray(src)
        height = src.shape[0]
        width = src.shape[1] if rgba16f else src.shape[1] // 4
        raw = src.view(np.uint8).reshape(height, -1)
        out = np.zeros((out_height, out_width * 4), dtype=np.uint8)
        fn = self.lib.bt709o_render_scaled
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t, C.c_int, C.c_int]
        rc = fn(int(rgba16f), _ptr(raw, _u8p), raw.shape[1], width, height, _ptr(out, _u8p), out.shape[1], out_width, out_height)
        return out if rc == 0 else None

    def half_table(self, gamma, nthreads=8):
        """(2^24, 3) uint16: R,G,B half codes of every (Y,Cb,Cr), index (Y<<16)+(Cb<<8)+Cr."""
        cb, cr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
        c = np.empty((256, 512), np.uint8)
        c[:, 0::2], c[:, 1::2] = cb, cr  # one chroma row per Cb, Cr running along it
        out = np.empty((256, 65536, 3), np.uint16)
        from concurrent.futures import ThreadPoolExecutor

        def one(Y):
            # a 512 x 512 frame: block (Cb, Cr), all four luma samples = Y
            y = np.full((512, 512), Y, np.uint8)
            cc = np.repeat(c.reshape(256, 256, 2), 1, axis=0).reshape(256, 512)
            px = self.decode_nv12_rgba16f(gamma, y, cc).view(np.uint16)[::2, ::2, :3]
            out[Y] = px.reshape(65536, 3)

        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(one, range(256)))
        return out.reshape(-1, 3)

    def unconvert_packed(self, gamma, ycbcr, width, height):
        ycbcr = np.ascontiguousarray(ycbcr, dtype=np.uint32)
        out = np.zeros(width * height, dtype=np.uint32)
        rc = self.lib.bt709o_unconvert_packed(gamma, _ptr(ycbcr, _u32p), _ptr(out, _u32p),
                                              width, height)
        return out if rc == 0 else None

    def convert_packed(self, gamma, bgra, width, height):
        bgra = np.ascontiguousarray(bgra, dtype=np.uint32)
        out = np.zeros(width * height, dtype=np.uint32)
        rc = self.lib.bt709o_convert_packed(gamma, _ptr(bgra, _u32p), _ptr(out, _u32p),
                                            width, height)
        return out if rc == 0 else None

    def packed_to_nv12(self, ycbcr, width, height):
        ycbcr = np.ascontiguousarray(ycbcr, dtype=np.uint32)
        y = np.zeros((height, width), dtype=np.uint8)
        uv = np.zeros((height // 2, width), dtype=np.uint8)
        self.lib.bt709o_packed_to_nv12(_ptr(ycbcr, _u32p), width, height,
                                       _ptr(y, _u8p), width, _ptr(uv, _u8p), width)
        return y, uv

    def nv12_to_packed(self, y, uv):
        y, width = _plane(y)
        uv, _ = _plane(uv)
        height = y.shape[0]
        out = np.zeros(width * height, dtype=np.uint32)
        self.lib.bt709o_nv12_to_packed(_ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p),
                                       uv.shape[1], width, height, _ptr(out, _u32p))
        return out

    def subsample_block(self, rgb12, in_gamma, out_gamma):
        rgb = (C.c_int * 12)(*rgb12)
        y4 = (C.c_int * 4)()
        cb, cr = C.c_int(), C.c_int()
        self.lib.bt709o_subsample_block(rgb, in_gamma, out_gamma, y4, C.byref(cb), C.byref(cr))
        return tuple(y4) + (cb.value, cr.value)

    def encode_nv12(self, bgra, width, height, in_gamma=GAMMA_SRGB, out_gamma=GAMMA_APPLE):
        bgra = np.ascontiguousarray(bgra, dtype=np.uint32)
        y = np.zeros((height, width), dtype=np.uint8)
        uv = np.zeros((height // 2, width), dtype=np.uint8)
        rc = self.lib.bt709o_encode_nv12(_ptr(bgra, _u32p), width, height, in_gamma, out_gamma,
                                         _ptr(y, _u8p), width, _ptr(uv, _u8p), width)
        return (y, uv) if rc == 0 else None

    # -- exhaustive
    def decode_table(self, gamma, nthreads=8):
        t = np.zeros((1 << 24) * 3, dtype=np.uint8)
        self.lib.bt709o_decode_table(gamma, _ptr(t, _u8p), nthreads)
        return t

    def roundtrip_histogram(self, gamma, nthreads=8):
        h = np.zeros(11, dtype=np.uint64)
        self.lib.bt709o_roundtrip_histogram(gamma, _ptr(h, _u64p), nthreads)
        return [int(v) for v in h]

    def thresholds(self, gamma):
        t = np.zeros(255, dtype=np.float32)
        self.lib.bt709o_thresholds(gamma, _ptr(t, _f32p))
        return t

    def check_thresholds(self, gamma, lo_bits=0, hi_bits=0x3F800000, nthreads=8):
        return int(self.lib.bt709o_check_thresholds(gamma, lo_bits, hi_bits, nthreads))


def _plane(p):
    """Accept `array` (width = pitch) or `(array, width)`."""
    if isinstance(p, tuple):
        arr, width = p
    else:
        arr, width = p, p.shape[1]
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    assert arr.ndim == 2
    return arr, int(width)


class Reference:
    """The reference's own BT709.h / sRGB.h, compiled in place (container only)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            build_oracle()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        L = self.lib = C.CDLL(REF_SO)
        L.ref_decode_pixel.argtypes = [C.c_int] * 4 + [_i32p]
        L.ref_encode_pixel.argtypes = [C.c_int] * 4 + [_i32p]
        L.ref_ycbcr_to_rgbn.argtypes = [C.c_int] * 3 + [_f32p]
        L.ref_decode_alpha.restype, L.ref_decode_alpha.argtypes = C.c_int, [C.c_int]
        L.ref_transfer_to_byte.restype = C.c_int
        L.ref_transfer_to_byte.argtypes = [C.c_int, C.c_float]
        for name in ("srgb_to_linear", "linear_to_srgb", "itu709_to_linear",
                     "linear_to_itu709", "apple196_to_linear", "linear_to_apple196"):
            f = getattr(L, "ref_" + name)
            f.restype, f.argtypes = C.c_float, [C.c_float]
        L.ref_decode_table.argtypes = [C.c_int, _u8p, C.c_int, C.c_int]
        L.ref_roundtrip_histogram.argtypes = [C.c_int, _u64p, C.c_int, C.c_int]
        L.ref_subsample_block.argtypes = [_i32p, C.c_int, C.c_int, _i32p]
        L.ref_encode_nv12.argtypes = [_u32p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
        L.ref_decode_nv12_rows.argtypes = [C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int,
                                           C.c_int, C.c_int, _u8p, C.c_size_t, C.c_int]

    def decode_nv12(self, gamma, y, uv, alpha_fill=0xFF, rows=None, out=None):
        """Frame loop over the reference's per-pixel function (cpu_baseline "reference")."""
        y, width = _plane(y)
        uv, _ = _plane(uv)
        height = y.shape[0]
        if out is None:
            out = np.zeros((height, width * 4), dtype=np.uint8)
        r0, r1 = rows if rows is not None else (0, height)
        self.lib.ref_decode_nv12_rows(gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
                                      width, r0, r1, _ptr(out, _u8p), out.shape[1], alpha_fill)
        return out

    def decode_nv12_half(self, gamma, y, uv, alpha_fill=0xFF, alpha=None):
        """Pass 1 + exact 2:1 pass 2, composed of the reference's own inlines (oracle/ref_harness.c)."""
        y, width = _plane(y)
        uv, _ = _plane(uv)
        a = _plane(alpha)[0] if alpha is not None else None
        height = y.shape[0]
        out = np.zeros((height // 2, (width // 2) * 4), dtype=np.uint8)
        fn = self.lib.ref_decode_nv12_half
        fn.restype = None
        fn.argtypes = [C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p,
                       C.c_size_t, C.c_int]
        fn(gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
           _ptr(a, _u8p) if a is not None else None, a.shape[1] if a is not None else 0, width, height,
           _ptr(out, _u8p), out.shape[1], alpha_fill)
        return out

    def decode_nv12_scaled(self, gamma, y, uv, out_width, out_height, alpha_fill=0xFF, alpha=None):
        y, width = _plane(y)
        uv, _ = _plane(uv)
        a = _plane(alpha)[0] if alpha is not None else None
        height = y.shape[0]
        out = np.zeros((out_height, out_width * 4), dtype=np.uint8)
        fn = self.lib.ref_decode_nv12_scaled
        fn.restype = None
        fn.argtypes = [C.c_int, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p,
                       C.c_size_t, C.c_int, C.c_int, C.c_int]
        fn(gamma, _ptr(y, _u8p), y.shape[1], _ptr(uv, _u8p), uv.shape[1],
           _ptr(a, _u8p) if a is not None else None, a.shape[1] if a is not None else 0, width, height,
           _ptr(out, _u8p), out.shape[1], out_width, out_height, alpha_fill)
        return out

    def half_table(self, gamma, y0=0, y1=256):
        """(2^24, 3) uint16 half codes from the reference's own matrix and curve functions."""
        t = np.zeros(((1 << 24), 3), dtype=np.uint16)
        fn = self.lib.ref_rgba16f_table
        fn.restype = None
        fn.argtypes = [C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda r: fn(gamma, t.ctypes.data_as(C.POINTER(C.c_uint16)), r, min(r + 16, y1)), range(y0, y1, 16)))
        return t

    def alpha_half(self, A):
        self.lib.ref_alpha_half.restype = C.c_int
        return self.lib.ref_alpha_half(int(A))

    def decode_pixel(self, gamma, Y, Cb, Cr):
        out = (C.c_int * 3)()
        self.lib.ref_decode_pixel(gamma, Y, Cb, Cr, out)
        return tuple(out)

    def encode_pixel(self, gamma, R, G, B):
        out = (C.c_int * 3)()
        self.lib.ref_encode_pixel(gamma, R, G, B, out)
        return tuple(out)

    def decode_alpha(self, A):
        return self.lib.ref_decode_alpha(A)

    def transfer_to_byte(self, gamma, v):
        return self.lib.ref_transfer_to_byte(gamma, float(v))

    def decode_table(self, gamma, y0=0, y1=256):
        t = np.zeros((1 << 24) * 3, dtype=np.uint8)
        self.lib.ref_decode_table(gamma, _ptr(t, _u8p), y0, y1)
        return t

    def roundtrip_histogram(self, gamma, r0=0, r1=256):
        h = np.zeros(11, dtype=np.uint64)
        self.lib.ref_roundtrip_histogram(gamma, _ptr(h, _u64p), r0, r1)
        return [int(v) for v in h]

    def subsample_block(self, rgb12, in_gamma, out_gamma):
        rgb = (C.c_int * 12)(*rgb12)
        out = (C.c_int * 6)()
        self.lib.ref_quiet_begin()
        try:
            self.lib.ref_subsample_block(rgb, in_gamma, out_gamma, out)
        finally:
            self.lib.ref_quiet_end()
        return tuple(out)

    def encode_nv12(self, bgra, width, height, in_gamma=GAMMA_SRGB, out_gamma=GAMMA_APPLE):
        bgra = np.ascontiguousarray(bgra, dtype=np.uint32)
        y = np.zeros((height, width), dtype=np.uint8)
        uv = np.zeros((height // 2, width), dtype=np.uint8)
        self.lib.ref_encode_nv12(_ptr(bgra, _u32p), width, height, in_gamma, out_gamma,
                                 _ptr(y, _u8p), _ptr(uv, _u8p))
        return y, uv

    def thresholds(self, gamma):
        """Bisection on the float bit pattern against the reference composite."""
        t = np.zeros(255, dtype=np.float32)
        one = 0x3F800000
        for k in range(1, 256):
            lo, hi = 0, one
            if self.transfer_to_byte(gamma, _bits(hi)) < k:
                t[k - 1] = np.inf
                continue
            while lo < hi:
                mid = (lo + hi) // 2
                if self.transfer_to_byte(gamma, _bits(mid)) >= k:
                    hi = mid
                else:
                    lo = mid + 1
            t[k - 1] = _bits(lo)
        return t


def _bits(u):
    return np.array([u], dtype=np.uint32).view(np.float32)[0]
