#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_y4m.py -m gpu -x -q > gpurun_out/r06_3_tests.txt 2>&1; tail -5 gpurun_out/r06_3_tests.txt
python - <<'PY' > gpurun_out/r06_malloc_times.txt 2>&1
import ctypes as C, time, sys
sys.path.insert(0, "tests")
import gpu_helpers as gh
ctx = gh.context(); lib, h = ctx.lib, ctx.handle
for gb in (3.2, 8.5):
    n = int(gb * 1e9)
    for i in range(4):
        p = C.c_void_p(); t0 = time.perf_counter(); lib.bt709hip_malloc(h, n, C.byref(p)); t1 = time.perf_counter()
        lib.bt709hip_memset(h, p, 0, 1 << 20, None); lib.bt709hip_stream_synchronize(h, None); t2 = time.perf_counter()
        lib.bt709hip_free(h, p); t3 = time.perf_counter()
        print("%.1f GB: malloc %.1f ms, first touch %.1f ms, free %.1f ms" % (gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
PY
cat gpurun_out/r06_malloc_times.txt
O=gpurun_out/r06_ab_hunt.txt; rm -f $O
bash tools/ab_hunt.sh $O 4 "default(1 input)=;2 inputs=tools/bin/hunt_in2.so;warm 10 ms=tools/bin/hunt_w10.so;warm 10 ms, confirm x3=tools/bin/hunt_w10c3.so"
cut -c1-170 $O; tail -3 $O.err
