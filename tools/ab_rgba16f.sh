#!/bin/bash
# RGBA16F target: one-sided settlement (shipping) against round 2's first form, same call (runs on the GPU box).
#   the two-sided form is commit 8e386e3's kernel: build that tree's library as tools/bin/lib_twosided.so
cd "${GRAFT_REPO_ROOT:-.}"
P='import sys,json; d=json.loads(sys.stdin.read()); print("%-62s %8.2f us  %7.1f Gpx/s  %.4f" % (d["workload"], d["us_per_frame"], d["out_gpixel_per_s"], d["frac_of_8TBps"]))'
for rep in 1 2 3; do
  for lib in "" tools/bin/lib_twosided.so; do
    echo -n "${lib:-shipping}: "; python tools/bench_scaled.py --path rgba16f --frames-per-launch 16 ${lib:+--library $lib} | python -c "$P"
  done
done
