#!/bin/bash
# The placement hunt of bt709hip_ring_create under variant builds (python -m metalbt709decoder_amd.build --variant X.so BT709_HUNT_...): N fresh bench.py processes per library,
# one line each: what the hunt cost and what the headline then measured on the ring it chose (runs on the GPU box).
#   tools/ab_hunt.sh OUT N "label=library.so;..."  [extra bench.py args]
cd "${GRAFT_REPO_ROOT:-.}"
O=$1; N=$2; LIBS=$3; shift 3
IFS=';' read -ra LB <<< "$LIBS"
for i in $(seq $N); do
  for l in "${LB[@]}"; do
    name=${l%%=*}; lib=${l#*=}; L=""; [ -n "$lib" ] && L="--library $lib"
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-smooth-leg $L "$@" 2>>$O.err | python -c '
import sys, json
d = json.loads(sys.stdin.read()); p = d["config"]["placement"]; r = d["roofline"]
print("%-34s frac %.4f first_allocation %.4f hunt_ms %5.0f peak %.1f GB candidates %s chosen %s probe_chosen %.0f prescan %s" % (sys.argv[1], r["frac"], r["first_allocation_frac"],
      p["hunt_ms"], p["peak_bytes"] / 1e9, p["candidates"], p["chosen"], p["probe_GBps"]["chosen_confirmed"], p["output_prescan_GBps"]))' "$name" >> $O
  done
done
