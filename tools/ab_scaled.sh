#!/bin/bash
# The any-ratio kernel (tools/bench_scaled.py --path scaled) over a list of shapes x a list of libraries, fresh process per line (runs on the GPU box).
#   [BENCH_PATH=render8|render16] tools/ab_scaled.sh OUT "W H OW OH FPL RING;..." "label=library.so;..."     (library empty = the shipped one)
cd "${GRAFT_REPO_ROOT:-.}"
O=$1; SHAPES=$2; LIBS=$3; ROUNDS=${ROUNDS:-2}
P='import sys,json; d=json.loads(sys.stdin.read()); print("%-28s %-58s %8.2f us/frame  %7.1f Gpx/s out  frac %.4f" % (sys.argv[1], d["workload"], d["us_per_frame"], d["out_gpixel_per_s"], d["frac_of_8TBps"]))'
IFS=';' read -ra SH <<< "$SHAPES"; IFS=';' read -ra LB <<< "$LIBS"
for s in "${SH[@]}"; do
  set -- $s
  for r in $(seq $ROUNDS); do
    for l in "${LB[@]}"; do
      name=${l%%=*}; lib=${l#*=}; L=""; [ -n "$lib" ] && L="--library $lib"
      timeout 300 python tools/bench_scaled.py --path ${BENCH_PATH:-scaled} $L --width $1 --height $2 --out-width $3 --out-height $4 --frames-per-launch $5 --ring $6 2>>$O.err | python -c "$P" "$name" >> $O
    done
  done
done
