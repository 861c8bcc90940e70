#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite after the refactors, the new bench line, the counter list of this box
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04a; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_4k_driver_args.json 2> $O/bench_4k.err; echo "bench rc=$?"; tail -3 $O/bench_4k.err
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_4k_2ranks_one_gpu.json 2> $O/bench_4k_2ranks.err; echo "bench2 rc=$?"; tail -3 $O/bench_4k_2ranks.err
for c in 0 8 32; do timeout 300 python bench.py --workload 4k-batch8 --share 1 --streams 1 --coalesce $c --no-cpu-baseline --steps 400 > $O/bench_batch8_share1_1stream_coalesce$c.json 2>/dev/null; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > "$OLDPWD/$O/counters_avail.txt" 2>&1 )
grep -c . $O/counters_avail.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f.split("/")[-1], d.get("value"), r.get("frac"), r.get("first_allocation_frac"), r.get("avg_launch_us"), r.get("same_run_copy_GBps"), d.get("parity_spot_check"), d.get("parity_spot_frames"))
        print("   placement", json.dumps(d["config"].get("placement"))[:600])
    except Exception as e: print(f, "ERR", e)
PY
