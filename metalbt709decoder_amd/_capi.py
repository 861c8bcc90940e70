"""ctypes binding of the C ABI declared in include/bt709hip.h (the reference-twinned calls) and include/bt709hip_ext.h.

Loading never falls back to anything: if libbt709hip.so is missing and cannot be
built, importing the product raises.
"""
import ctypes as C
import os

from . import build as _build

c_void_pp = C.POINTER(C.c_void_p)


class Frame(C.Structure):  # bt709hip_frame
    _fields_ = [("y", C.c_void_p), ("y_stride", C.c_size_t),
                ("cbcr", C.c_void_p), ("cbcr_stride", C.c_size_t),
                ("width", C.c_int32), ("height", C.c_int32),
                ("matrix", C.c_int32), ("transfer", C.c_int32)]


class Surface(C.Structure):  # bt709hip_surface
    _fields_ = [("bgra", C.c_void_p), ("stride", C.c_size_t),
                ("width", C.c_int32), ("height", C.c_int32),
                ("format", C.c_int32), ("reserved", C.c_int32)]


ABI_VERSION = 502  # BT709HIP_VERSION of the two headers these bindings were written against

# bt709hip_format
FORMAT_BGRA8_SRGB = 0
FORMAT_RGBA16F = 1

# bt709hip_decoder_option / bt709hip_context_option
OPT_NONTEMPORAL = 1
OPT_HALF_KERNEL = 2
OPT_HALF_WORKGROUPS = 3
OPT_HALF_LDS_KB = 4
OPT_XCD_BANDS = 5
OPT_COALESCE = 6
OPT_COALESCE_MAX_AGE_US = 7
CTX_OPT_GRID_MULT = 1
CTX_OPT_ENCODE_ROW_PAIRS = 2
CTX_OPT_ENCODE_THREADS = 3
CTX_OPT_XCD_BANDS = 4
CTX_OPT_STREAMING_TRIES = 5


class RingPlacement(C.Structure):  # bt709hip_ring_placement
    _fields_ = [("tries", C.c_int32), ("in_candidates", C.c_int32), ("out_candidates", C.c_int32),
                ("chosen_in", C.c_int32), ("chosen_out", C.c_int32), ("probes", C.c_int32),
                ("first_GBps", C.c_float), ("chosen_GBps", C.c_float), ("best_GBps", C.c_float), ("worst_GBps", C.c_float),
                ("out_prescan_GBps", C.c_float * 18), ("out_kept", C.c_int32 * 18),
                ("hunt_ms", C.c_float), ("stopped_by", C.c_int32), ("peak_bytes", C.c_uint64), ("budget_bytes", C.c_uint64),
                ("evicted", C.c_int32), ("reserved", C.c_int32)]


class RingOptions(C.Structure):  # bt709hip_ring_options
    _fields_ = [("max_bytes", C.c_uint64), ("max_ms", C.c_uint32), ("frugal", C.c_int32), ("format", C.c_int32), ("reserved", C.c_int32)]


class LaunchInfo(C.Structure):  # bt709hip_launch_info
    _fields_ = [("grid", C.c_uint32 * 3), ("block", C.c_uint32 * 3), ("launches", C.c_int32), ("xcd_bands", C.c_int32)]


class DeviceInfo(C.Structure):  # bt709hip_device_info
    _fields_ = [("device_ordinal", C.c_int32), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("lds_bytes_per_block", C.c_int32),
                ("memory_clock_khz", C.c_int32), ("memory_bus_width_bits", C.c_int32),
                ("l2_bytes", C.c_int32), ("clock_khz", C.c_int32),
                ("total_memory_bytes", C.c_uint64),
                ("name", C.c_char * 128), ("arch", C.c_char * 64),
                ("pci_bus_id", C.c_char * 32), ("uuid", C.c_char * 40)]


# status codes (bt709hip_status)
OK = 0
ERR_INVALID_ARG = -1
ERR_NOT_SETUP = -2
ERR_SIZE_MISMATCH = -3
ERR_ODD_DIMENSIONS = -4
ERR_MATRIX = -5
ERR_TRANSFER = -6
ERR_ALPHA_TRANSFER = -7
ERR_STRIDE = -8
ERR_HIP = -9
ERR_NO_DEVICE = -10
ERR_UNSUPPORTED = -11

MAX_BATCH = 32

# every symbol the two headers declare: name -> (restype, argtypes)
_P, _I, _Z = C.c_void_p, C.c_int, C.c_size_t
_FP, _SP = C.POINTER(Frame), C.POINTER(Surface)
SYMBOLS = {
    "bt709hip_device_count": (_I, []),
    "bt709hip_abi_version": (_I, []),
    "bt709hip_context_set_option": (_I, [_P, _I, _I]),
    "bt709hip_context_create": (_I, [_I, c_void_pp]),
    "bt709hip_context_destroy": (_I, [_P]),
    "bt709hip_context_info": (_I, [_P, C.POINTER(DeviceInfo)]),
    "bt709hip_stream_create": (_I, [_P, c_void_pp]),
    "bt709hip_stream_create_with_priority": (_I, [_P, _I, c_void_pp]),
    "bt709hip_stream_destroy": (_I, [_P, _P]),
    "bt709hip_stream_synchronize": (_I, [_P, _P]),
    "bt709hip_event_create": (_I, [_P, c_void_pp]),
    "bt709hip_event_destroy": (_I, [_P, _P]),
    "bt709hip_event_record": (_I, [_P, _P, _P]),
    "bt709hip_event_synchronize": (_I, [_P, _P]),
    "bt709hip_stream_wait_event": (_I, [_P, _P, _P]),
    "bt709hip_event_elapsed_ms": (_I, [_P, _P, _P, C.POINTER(C.c_float)]),
    "bt709hip_pool_create": (_I, [_P, _I, _I, _I, C.POINTER(C.c_void_p)]),
    "bt709hip_pool_destroy": (_I, [_P]),
    "bt709hip_pool_acquire": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "bt709hip_pool_alpha_plane": (_I, [_P, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "bt709hip_pool_submit": (_I, [_P, _I]),
    "bt709hip_pool_wait": (_I, [_P, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "bt709hip_pool_release": (_I, [_P, _I]),
    "bt709hip_shard_create": (_I, [C.POINTER(C.c_int), _I, _I, _I, _I, _I, _I, C.POINTER(C.c_void_p)]),
    "bt709hip_shard_destroy": (_I, [_P]),
    "bt709hip_shard_lanes": (_I, [_P]),
    "bt709hip_shard_lane_device": (_I, [_P, _I]),
    "bt709hip_shard_lane_decoder": (_P, [_P, _I]),
    "bt709hip_shard_acquire": (_I, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t)]),
    "bt709hip_shard_commit": (_I, [_P, C.c_uint64]),
    "bt709hip_shard_cancel": (_I, [_P]),
    "bt709hip_shard_submit": (_I, [_P, _FP, _FP, C.POINTER(C.c_uint64)]),
    "bt709hip_shard_wait": (_I, [_P, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "bt709hip_graph_begin_capture": (_I, [_P, _P]),
    "bt709hip_graph_end_capture": (_I, [_P, _P, C.POINTER(C.c_void_p)]),
    "bt709hip_graph_launch": (_I, [_P, _P, _P]),
    "bt709hip_graph_destroy": (_I, [_P, _P]),
    "bt709hip_mem_info": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bt709hip_decoder_has_alpha": (_I, [_P]),
    "bt709hip_decoder_context": (_P, [_P]),
    "bt709hip_decoder_flush": (_I, [_P, _P]),
    "bt709hip_decoder_flush_all": (_I, [_P]),
    "bt709hip_ring_create": (_I, [_P, _I, _I, _I, _I, _I, c_void_pp]),
    "bt709hip_ring_create_ex": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(RingOptions), c_void_pp]),
    "bt709hip_ring_destroy": (_I, [_P]),
    "bt709hip_ringset_create": (_I, [C.POINTER(C.c_int), _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(RingOptions), c_void_pp]),
    "bt709hip_ringset_destroy": (_I, [_P]),
    "bt709hip_ringset_lanes": (_I, [_P]),
    "bt709hip_ringset_lane_context": (_P, [_P, _I]),
    "bt709hip_ringset_lane_decoder": (_P, [_P, _I]),
    "bt709hip_ringset_lane_ring": (_P, [_P, _I]),
    "bt709hip_ringset_decode": (_I, [_P, _I, _I, _I]),
    "bt709hip_ringset_synchronize": (_I, [_P]),
    "bt709hip_ring_frames": (_I, [_P]),
    "bt709hip_ring_frame": (_I, [_P, _I, _FP, _FP, _SP]),
    "bt709hip_ring_placement_info": (_I, [_P, C.POINTER(RingPlacement)]),
    "bt709hip_ring_decode": (_I, [_P, _I, _I, _P, _I]),
    "bt709hip_malloc": (_I, [_P, _Z, c_void_pp]),
    "bt709hip_free": (_I, [_P, _P]),
    "bt709hip_host_alloc": (_I, [_P, _Z, c_void_pp]),
    "bt709hip_host_free": (_I, [_P, _P]),
    "bt709hip_memset": (_I, [_P, _P, _I, _Z, _P]),
    "bt709hip_upload": (_I, [_P, _P, _Z, _P, _Z, _Z, _Z, _P]),
    "bt709hip_download": (_I, [_P, _P, _Z, _P, _Z, _Z, _Z, _P]),
    "bt709hip_decoder_create": (_I, [_P, _I, _I, c_void_pp]),
    "bt709hip_decoder_destroy": (_I, [_P]),
    "bt709hip_decoder_set_context": (_I, [_P, _P]),
    "bt709hip_decoder_set_alpha_fill": (_I, [_P, _I]),
    "bt709hip_decoder_get_gamma": (_I, [_P]),
    "bt709hip_decoder_set_option": (_I, [_P, _I, _I]),
    "bt709hip_decoder_get_option": (_I, [_P, _I, C.POINTER(C.c_int)]),
    "bt709hip_decoder_setup": (_I, [_P]),
    "bt709hip_decode": (_I, [_P, _FP, _FP, _SP, _I, _I, _P, _I]),
    "bt709hip_decode_batch": (_I, [_P, _I, _FP, _FP, _SP, _P, _I]),
    "bt709hip_unconvert": (_I, [_P, _P, _Z, _I, _I, _SP, _P, _I]),
    "bt709hip_unconvert_batch": (_I, [_P, _I, C.POINTER(C.c_void_p), _Z, _I, _I, _SP, _P, _I]),
    "bt709hip_decode_half": (_I, [_P, _FP, _FP, _SP, _P, _I]),
    "bt709hip_decode_half_batch": (_I, [_P, _I, _FP, _FP, _SP, _P, _I]),
    "bt709hip_decode_scaled": (_I, [_P, _FP, _FP, _SP, _P, _I]),
    "bt709hip_decode_scaled_batch": (_I, [_P, _I, _FP, _FP, _SP, _P, _I]),
    "bt709hip_render_scaled": (_I, [_P, _SP, _SP, _P, _I]),
    "bt709hip_render_scaled_batch": (_I, [_P, _I, _SP, _SP, _P, _I]),
    "bt709hip_render_scaled_prepare": (_I, [_P]),
    "bt709hip_decoder_prepare_format": (_I, [_P, _I]),
    "bt709hip_encoder_prepare": (_I, [_P, _I, _I]),
    "bt709hip_encode": (_I, [_P, _SP, _FP, _I, _I, _P, _I]),
    "bt709hip_encode_batch": (_I, [_P, _I, _SP, _FP, _I, _I, _P, _I]),
    "bt709hip_interleave_cbcr": (_I, [_P, _P, _Z, _P, _Z, _P, _Z, _I, _I, _P, _I]),
    "bt709hip_deinterleave_cbcr": (_I, [_P, _P, _Z, _P, _Z, _P, _Z, _I, _I, _P, _I]),
    "bt709hip_copy_probe": (_I, [_P, _P, _P, _Z, _P]),
    "bt709hip_malloc_streaming": (_I, [_P, _Z, _I, c_void_pp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "bt709hip_strerror": (C.c_char_p, [_I]),
    "bt709hip_last_hip_error": (_I, []),
    "bt709hip_last_hip_error_string": (C.c_char_p, []),
    "bt709hip_gamma_thresholds": (_I, [_I, C.POINTER(C.c_float)]),
    "bt709hip_gamma_lookup": (_I, [_I, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bt709hip_gamma_lookup_decode": (_I, [_I, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bt709hip_matrix_constants": (_I, [C.POINTER(C.c_float)]),
    "bt709hip_half_thresholds": (_I, [_I, C.POINTER(C.c_float), _I]),
    "bt709hip_half_lookup": (_I, [_I, C.c_float, _I, C.POINTER(C.c_int)]),
    "bt709hip_last_kernel_name": (C.c_char_p, []),
    "bt709hip_last_launch_info": (_I, [C.POINTER(LaunchInfo)]),
}

_lib = None


def library_path():
    return _build.LIB


def load(path=None):
    """Return the loaded C-ABI library, building it in-tree first if it is stale.  Never runs a
    library older than its sources or of another ABI version: when the library is stale and no
    compiler is present, or the loaded library reports a BT709HIP_VERSION other than these
    bindings', this raises (no fallback of any kind).
    path: load THIS build of the library instead (A/B runs of tools/build_variant.py; must be the
    first load of the process); it is not rebuilt, only ABI-checked."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.realpath(path) != os.path.realpath(_lib._name):
            raise RuntimeError("another build of the library is already loaded: %s" % _lib._name)
        return _lib
    if path is None and _build.is_stale():
        path = _build.LIB
        try:
            _build.build()  # a compile error propagates
        except OSError as exc:  # no hipcc on this machine
            raise ImportError("libbt709hip.so is %s and could not be (re)built: %s"
                              % ("stale" if os.path.exists(path) else "missing", exc))
    lib = C.CDLL(path or _build.LIB)
    path = path or _build.LIB
    try:
        lib.bt709hip_abi_version.restype = C.c_int
        have = lib.bt709hip_abi_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise ImportError("%s reports ABI version %s, these bindings need %d" % (path, have, ABI_VERSION))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def strerror(status):
    return load().bt709hip_strerror(int(status)).decode()


class Bt709Error(RuntimeError):
    def __init__(self, status, what=""):
        self.status = int(status)
        msg = "%s: %s" % (what, strerror(status)) if what else strerror(status)
        if self.status == ERR_HIP:
            msg += " (%s)" % load().bt709hip_last_hip_error_string().decode()
        super().__init__(msg)


def check(status, what=""):
    if status != OK:
        raise Bt709Error(status, what)
