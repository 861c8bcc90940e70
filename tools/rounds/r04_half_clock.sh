#!/bin/bash
# Round 4: is the 7 % between rocprof's per-launch time (384 us) and the sustained region time (413.5 us) of
# decode_nv12_half_rep dead time between launches, or a clock that sags when the VALU-bound kernel is sustained?
# 16 / 32 / 64 frames per launch on a ring of 64 8K frames, regions of >= 1 s, the shader clock sampled from sysfs meanwhile;
# then short bursts (the placement probe's ~15 ms) of the same launch for the burst rate.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_half; mkdir -p $O
for fpl in 16 32 64; do
  python tools/clock_sampler.py $O/clk_$fpl.csv 0.05 & S=$!
  python bench.py --workload 8k-half --ring 64 --frames-per-launch $fpl --steps 700 --warmup 50 --repeats 5 --no-cpu-baseline --placement-tries 1 > $O/bench_fpl$fpl.json 2>/dev/null
  kill $S; wait $S 2>/dev/null
  python - $O/bench_fpl$fpl.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
fpl = r["algorithmic_bytes_per_launch"] / 82944000
print("frames per launch %d: region %.0f ms  per 8K frame %.3f us  launch %.1f us  frac %.4f  value %.1f Gpx/s" % (
  fpl, d["ms_per_step"]*d["region_steps"], r["avg_launch_us"]/fpl, r["avg_launch_us"], r["frac"], d["value"]))
PY
  python tools/clock_sampler.py --summary $O/clk_$fpl.csv
done
# idle clock for reference
python tools/clock_sampler.py $O/clk_idle.csv 0.05 & S=$!; sleep 1.5; kill $S; wait $S 2>/dev/null; echo -n "idle: "; python tools/clock_sampler.py --summary $O/clk_idle.csv
rm -f $O/clk_*.csv
for f in 16 64; do python tools/clock_sampler.py $O/clk_b$f.csv 0.05 & S=$!; python tools/burst_vs_sustained.py 8k-half $f; kill $S; wait $S 2>/dev/null; python tools/clock_sampler.py --summary $O/clk_b$f.csv; done
python tools/burst_vs_sustained.py 4k 256
rm -f $O/clk_*.csv
