#!/bin/bash
# Round 5: two stubs that price work VERDICT r4 asked about (wrong output by construction, tools/lab_variants.py):
#  (1) BT709_LAB_HALF_ENCODE_B32: the persistent 2:1 kernel's encode side as 4-byte entries in twice the copies (item 5)
#  (2) BT709_LAB_SCALED_QUARTER_FEWER_TAPS: the any-ratio kernel with a quarter of its tap decodes deleted = the most a ratio-1.5
#      specialisation (3x3 source block -> 2x2 outputs per lane, 27 instead of 36 tap decodes per 4 outputs) could save (item 7)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
P='import sys,json
d=json.loads(sys.stdin.readline()); r=d["roofline"]
print("%-28s %8.2f us per 16-frame launch  frac %.4f  first allocation %.4f  spot check: %s" % (sys.argv[1], r["avg_launch_us"], r["frac"], r.get("first_allocation_frac", 0), d["parity_spot_check"][:40]))'
{
echo "# (1) BASELINE config 4 (8K -> 4K, 16 per launch, uniform random bytes): python3 bench.py --workload 8k-half, fresh process per line, shipped against the stub library"
for rep in 1 2 3; do
  python3 bench.py --workload 8k-half --steps 20 --warmup 5 --no-cpu-baseline --no-smooth-leg 2>/dev/null | python3 -c "$P" shipped
  python3 bench.py --workload 8k-half --steps 20 --warmup 5 --no-cpu-baseline --no-smooth-leg --library tools/bin/lab_half_enc_b32.so 2>/dev/null | python3 -c "$P" "4-byte entries x 2 copies"
done
} > gpurun_out/r05/ab_half_encode_b32.txt 2>&1
Q='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("%-34s %-60s %7.2f us/frame  frac %.4f" % (sys.argv[1], d["workload"], d["us_per_frame"], d["frac_of_8TBps"]))'
{
echo "# (2) any-ratio fused decode + rescale (tools/bench_scaled.py --path scaled), fresh process per line, shipped against the stub with a quarter of the tap decodes deleted"
for rep in 1 2 3; do
  for lib in "" "--library tools/bin/lab_scaled_quarter.so"; do
    n=shipped; [ -n "$lib" ] && n="quarter of the taps deleted"
    python tools/bench_scaled.py --path scaled --frames-per-launch 8 $lib 2>/dev/null | python -c "$Q" "$n"
    python tools/bench_scaled.py --path scaled --frames-per-launch 1 $lib 2>/dev/null | python -c "$Q" "$n"
  done
done
} > gpurun_out/r05/ab_scaled_r15_bound.txt 2>&1
cat gpurun_out/r05/ab_half_encode_b32.txt gpurun_out/r05/ab_scaled_r15_bound.txt
