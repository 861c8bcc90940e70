#!/bin/bash
# the four shapes the any-ratio kernel is tuned on (runs on the GPU box); $1 = optional variant library
cd "${GRAFT_REPO_ROOT:-.}"
L=""; [ -n "${1:-}" ] && L="--library $1"
P='import sys,json; d=json.loads(sys.stdin.read()); print("%-62s %8.2f us  %7.1f Gpx/s out  %.3f" % (d["workload"], d["us_per_frame"], d["out_gpixel_per_s"], d["frac_of_8TBps"]))'
B="python tools/bench_scaled.py --path scaled $L"
$B | python -c "$P"
$B --frames-per-launch 8 | python -c "$P"
$B --width 1920 --height 1080 --out-width 3840 --out-height 2160 --frames-per-launch 8 | python -c "$P"
$B --width 7680 --height 4320 --out-width 3840 --out-height 2160 --frames-per-launch 4 --ring 8 | python -c "$P"
$B --width 1920 --height 1080 --out-width 1366 --out-height 768 --ring 32 --frames-per-launch 16 | python -c "$P"
