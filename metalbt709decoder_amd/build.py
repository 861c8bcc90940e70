"""In-tree build of the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m metalbt709decoder_amd.build [--force] [--asm]

Produces metalbt709decoder_amd/libbt709hip.so next to this file (git-ignored, but it
travels to the GPU box with the repo snapshot).  -ffp-contract=off is part of the
contract, not an optimisation choice: the kernels must not fuse multiply-adds
(see csrc/bt709_kernels.hip).
"""
import contextlib
import fcntl
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbt709hip.so")
ASM = os.path.join(HERE, "build", "bt709_kernels.s")  # decode + rescale + encode kernels, concatenated
SOURCES = ["bt709_kernels.hip", "bt709_rescale_half.hip", "bt709_rescale_scaled.hip", "bt709_rgba16f.hip", "bt709_encode.hip", "bt709_planes.hip",
           "shim_core.cpp", "shim_decode.cpp", "shim_convert.cpp", "shim_coalesce.cpp", "shim_pool_shard.cpp", "shim_introspect.cpp",
           "bt709_ring.cpp", "transfer_tables.cpp"]
KERNEL_SOURCES = ["bt709_kernels.hip", "bt709_rescale_half.hip", "bt709_rescale_scaled.hip", "bt709_rgba16f.hip", "bt709_encode.hip"]
HEADERS = ["bt709_kernels.h", "bt709_device.h", "bt709_constants.h", "bt709_quantise.h", "bt709_stage.h", "transfer_tables.h", "shim_internal.h", "bt709_rescale.h"]
ARCH = "gfx950"
# -fno-slp-vectorize: hipcc otherwise pairs scalar f32 multiplies/adds into v_pk_* ops, which run
# at half rate on gfx950 and need v_mov shuffles to build their operand pairs (measured: 458 VALU
# instructions of which 66 packed + 55 moves, vs 427 single-rate ones without it, per 16 pixels).
FLAGS = ["-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-Wall"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _deps():
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    files.append(os.path.join(os.path.dirname(HERE), "include", "bt709hip.h"))
    files.append(os.path.join(os.path.dirname(HERE), "include", "bt709hip_ext.h"))
    files.append(os.path.abspath(__file__))
    return files


def is_stale(target=LIB):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(f) > t for f in _deps())


@contextlib.contextmanager
def _build_lock():
    """Exclusive advisory lock around the stale check and the compile.  Every rank of a
    torch.distributed.run job imports the package at the same moment and may find the library
    stale together: the first one in builds, the others wait here and then see a fresh file."""
    fd = os.open(os.path.join(HERE, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def build(force=False, verbose=False):
    """Compile libbt709hip.so if missing or older than its sources; returns its path.
    Safe to call from several processes at once (file lock; the compiler writes to a private
    temporary name that is renamed into place only when complete)."""
    with _build_lock():
        if not force and not is_stale():
            return LIB
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        cmd = [_hipcc(), "--offload-arch=" + ARCH, *FLAGS, "-shared",
               *[os.path.join(CSRC, s) for s in SOURCES], "-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


def build_variant(out_path, defines, csrc=CSRC):
    """A second build of the library with extra -D flags (tile shape and other correct-output tunables, e.g.
    BT709_QUADS_PER_LANE=1), for same-call A/B runs: bench.py --library <out_path>.  Never replaces the in-tree
    library.  `csrc`: another source directory -- tools/lab_variants.py compiles a copy of csrc/ with its experiment
    gates re-inserted (the product sources carry none)."""
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    # items starting with "-" are raw compiler flags (e.g. -fslp-vectorize), the rest -D macros
    extra = [d if d.startswith("-") else "-D" + d for d in defines]
    flags = [f for f in FLAGS if not ("-fslp-vectorize" in extra and f == "-fno-slp-vectorize")]
    cmd = [_hipcc(), "--offload-arch=" + ARCH, *flags, *extra, "-shared",
           *[os.path.join(csrc, s) for s in SOURCES], "-o", out_path]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return out_path


def emit_asm(force=False):
    """Device-only gfx950 assembly of the kernels (used by the ISA checks in tests/)."""
    if not force and os.path.exists(ASM) and not is_stale(ASM):
        return ASM
    os.makedirs(os.path.dirname(ASM), exist_ok=True)
    parts = []
    for src in KERNEL_SOURCES:
        out = "%s.%d.%s" % (ASM, os.getpid(), src)
        cmd = [_hipcc(), "--offload-arch=" + ARCH, *[f for f in FLAGS if f != "-fPIC"], "-S",
               "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed:\n" + r.stdout + r.stderr)
        parts.append(open(out).read())
        os.remove(out)
    tmp = "%s.%d.tmp" % (ASM, os.getpid())
    with open(tmp, "w") as f:
        f.write("\n".join(parts))
    os.replace(tmp, ASM)
    return ASM


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m metalbt709decoder_amd.build --variant tools/bin/libbt709hip_q1.so BT709_QUADS_PER_LANE=1
        i = sys.argv.index("--variant")
        print("built", build_variant(os.path.abspath(sys.argv[i + 1]), sys.argv[i + 2:]))
        sys.exit(0)
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
    if "--asm" in sys.argv:
        print("asm  ", emit_asm(force=True))
