#!/usr/bin/env python3
"""Round 6: on some boxes hipMalloc of a multi-GB slab takes ~300 ms (0.3 ms on others) and bt709hip_ring_create's hunt -- twelve
output candidates -- takes 2-4 s instead of 0.9 s (profiles/r06_hunt_default.txt).  This lab times allocation patterns on
whatever box it lands on: fresh allocations, allocate-after-free (touched and untouched memory), with a spacer held, with the
freed slab's size changed, from a second thread while the first launches.  Prints one line per pattern; on a fast box every
number is under a millisecond and the lab says so."""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as gh  # noqa: E402

ctx = gh.context()
lib, h = ctx.lib, ctx.handle
GB = 1 << 30
N = int(8.5e9)


def ms(f):
    t0 = time.perf_counter()
    r = f()
    return (time.perf_counter() - t0) * 1e3, r


def malloc(n=N):
    p = C.c_void_p()
    t, _ = ms(lambda: lib.bt709hip_malloc(h, n, C.byref(p)))
    return t, p


def touch(p, n=N):
    lib.bt709hip_memset(h, p, 0, n, None)
    lib.bt709hip_stream_synchronize(h, None)


def free(p):
    return ms(lambda: lib.bt709hip_free(h, p))[0]


def pattern(free_first, n=12):
    """n candidates of 8.5 GB, each touched; the slab that goes is freed BEFORE (free_first) or AFTER the next one is allocated."""
    times, prev = [], None
    t0 = time.perf_counter()
    for i in range(n):
        if free_first and prev is not None:
            free(prev)
        t, p = malloc()
        if not free_first and prev is not None:
            free(prev)
        touch(p)
        times.append(t)
        prev = p
    free(prev)
    return (time.perf_counter() - t0) * 1e3, times


for rnd in range(3):
    for free_first in (True, False):
        total, times = pattern(free_first)
        print("%s: total %.0f ms, malloc %s" % ("free, then malloc " if free_first else "malloc, then free ", total, " ".join("%.0f" % t for t in times)))
sys.exit(0)

fresh = []
held = []
for i in range(3):
    t, p = malloc()
    fresh.append(t)
    held.append(p)
print("three fresh 8.5 GB slabs, all held: malloc %s ms" % ", ".join("%.1f" % t for t in fresh))
for p in held:
    touch(p)
tf = [free(p) for p in held]
print("touched, then freed: free %s ms" % ", ".join("%.1f" % t for t in tf))
if max(fresh) < 20.0:
    again = []
    for i in range(4):
        t, p = malloc()
        touch(p)
        again.append(t)
        free(p)
    print("malloc / touch / free x 4: malloc %s ms" % ", ".join("%.1f" % t for t in again))
    if max(again) < 20.0:
        print("FAST BOX: allocation costs nothing here; nothing to learn")
        sys.exit(0)
print("SLOW BOX: patterns")
seq = []
for i in range(4):
    t, p = malloc()
    touch(p)
    seq.append(t)
    free(p)
print("A. malloc / touch / free, same size, back to back: malloc %s ms" % ", ".join("%.1f" % t for t in seq))
seq = []
for i in range(4):
    t, p = malloc()
    seq.append(t)
    free(p)
print("B. malloc / free WITHOUT touching: malloc %s ms" % ", ".join("%.1f" % t for t in seq))
seq = []
prev = None
for i in range(5):
    t, p = malloc()
    touch(p)
    seq.append(t)
    if prev is not None:
        free(prev)
    prev = p
free(prev)
print("C. the previous slab freed only AFTER the next one is allocated: malloc %s ms" % ", ".join("%.1f" % t for t in seq))
seq = []
for i in range(4):
    t, p = malloc(N + i * (256 << 20))
    touch(p, N)
    seq.append(t)
    free(p)
print("D. sizes growing by 256 MB: malloc %s ms" % ", ".join("%.1f" % t for t in seq))
seq = []
for n in (1 * GB, 2 * GB, 4 * GB):
    t, p = malloc(n)
    touch(p, n)
    seq.append("%d GB %.1f" % (n // GB, t))
    free(p)
print("E. by size: malloc %s ms" % ", ".join(seq))
res = {}


def worker(k):
    res[k] = malloc()


ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
t0 = time.perf_counter()
for t in ths:
    t.start()
for t in ths:
    t.join()
wall = (time.perf_counter() - t0) * 1e3
print("F. three slabs from three threads at once: each %s ms, wall %.1f ms" % (", ".join("%.1f" % res[k][0] for k in range(3)), wall))
for k in range(3):
    free(res[k][1])
