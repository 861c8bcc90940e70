#!/bin/bash
# round 4, second GPU call: placement counters (item 3 i), 8k-half clock question (item 4), sentinel-cut rocprof of 4k / 1080p / 8k-half (item 5)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04b; mkdir -p $O
timeout 900 tools/placement_pmc_r04.sh > $O/placement_pmc.txt 2>&1; tail -40 $O/placement_pmc.txt
timeout 900 tools/rounds/r04_half_clock.sh > $O/half_clock.txt 2>&1; cat $O/half_clock.txt
for t in 4k 1080p 8k-half; do
  if [ $t = 4k ]; then timeout 900 tools/profile_gpu.sh $t > /dev/null 2>&1; else timeout 900 tools/profile_gpu.sh $t --workload $t > /dev/null 2>&1; fi
  grep '^{' gpurun_out/prof_$t/trace.log | tail -1 > $O/profiled_run_$t.json
  python tools/pmc_summary.py gpurun_out/prof_$t r04 $t > /dev/null
done
mkdir -p $O/profiles; cp profiles/r04_* profiles/pmc_traffic.json $O/profiles/
rm -rf gpurun_out/prof_* gpurun_out/pmcq_*
python - <<'PY'
import json,csv,glob
for t in ("4k","1080p","8k-half"):
    try:
        d=json.loads(open("gpurun_out/r04b/profiled_run_%s.json"%t).read()); r=d["roofline"]
        rows=list(csv.DictReader(open("profiles/r04_%s_kernel_stats.csv"%t)))
        print(t, "bench avg_launch_us", r["avg_launch_us"], "ms_per_step", d["ms_per_step"], "| rocprof timed-only:", [(x["Name"][:40], x["Calls"], x["AverageNs"], x["MedianNs"]) for x in rows])
    except Exception as e: print(t, "ERR", e)
PY
