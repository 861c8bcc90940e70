#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_4k_driver_args.json 2> $O/bench_4k.err; echo "bench rc=$?"
timeout 600 python bench.py --workload 1080p --no-cpu-baseline > $O/bench_1080p.json 2>/dev/null
timeout 600 python bench.py --workload 8k-half --no-cpu-baseline > $O/bench_8k-half.json 2>/dev/null
tools/bench_paths.sh > $O/bench_paths.txt 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f.split("/")[-1], d.get("value"), r.get("frac"), "first", r.get("first_allocation_frac"), r.get("avg_launch_us"), r.get("same_run_copy_GBps"), d.get("parity_spot_check"), (r.get("smooth_content") or {}).get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/bench_paths.txt
