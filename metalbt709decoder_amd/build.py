"""In-tree build of the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m metalbt709decoder_amd.build [--force] [--asm]

Produces metalbt709decoder_amd/libbt709hip.so next to this file (git-ignored, but it
travels to the GPU box with the repo snapshot).  -ffp-contract=off is part of the
contract, not an optimisation choice: the kernels must not fuse multiply-adds
(see csrc/bt709_kernels.hip).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbt709hip.so")
ASM = os.path.join(HERE, "build", "bt709_kernels.s")  # decode + rescale + encode kernels, concatenated
SOURCES = ["bt709_kernels.hip", "bt709_rescale.hip", "bt709_encode.hip", "bt709_planes.hip", "bt709hip.cpp",
           "transfer_tables.cpp"]
KERNEL_SOURCES = ["bt709_kernels.hip", "bt709_rescale.hip", "bt709_encode.hip"]
HEADERS = ["bt709_kernels.h", "bt709_device.h", "bt709_constants.h", "transfer_tables.h"]
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
    files.append(os.path.abspath(__file__))
    return files


def is_stale(target=LIB):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force=False, verbose=False):
    """Compile libbt709hip.so if missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=" + ARCH, *FLAGS, "-shared",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def emit_asm(force=False):
    """Device-only gfx950 assembly of the kernels (used by the ISA checks in tests/)."""
    if not force and os.path.exists(ASM) and not is_stale(ASM):
        return ASM
    os.makedirs(os.path.dirname(ASM), exist_ok=True)
    parts = []
    for src in KERNEL_SOURCES:
        out = ASM + "." + src
        cmd = [_hipcc(), "--offload-arch=" + ARCH, *[f for f in FLAGS if f != "-fPIC"], "-S",
               "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed:\n" + r.stdout + r.stderr)
        parts.append(open(out).read())
        os.remove(out)
    with open(ASM + ".tmp", "w") as f:
        f.write("\n".join(parts))
    os.replace(ASM + ".tmp", ASM)
    return ASM


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
    if "--asm" in sys.argv:
        print("asm  ", emit_asm(force=True))
