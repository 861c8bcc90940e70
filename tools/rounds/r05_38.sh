#!/bin/bash
# the rescale kernels' encode side through the log-bucket table (fma + shift) instead of the two-resolution table (multiply, convert, shift, add, min)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_38.txt
timeout 1500 python -m pytest tests/test_rescale.py tests/test_rgba16f.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 >> $O
for round in 1 2; do
for l in "" tools/bin/libbt709hip_head3.so; do
  echo "## ${l:-shipped} (round $round)" >> $O
  timeout 600 bash tools/bench_scaled_set.sh $l >> $O 2>&1
  L=""; [ -n "$l" ] && L="--library $l"
  for a in "--path render8" "--path render8 --frames-per-launch 16" "--path render16" "--path render16 --frames-per-launch 16"; do
    python tools/bench_scaled.py $a $L | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%-62s %8.2f us  %7.1f Gpx/s out  %.3f  %s" % (d["workload"], d["us_per_frame"], d["out_gpixel_per_s"], d["frac_of_8TBps"], d["kernel"]))' >> $O
  done
done
done
cat $O
