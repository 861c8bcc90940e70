#!/bin/bash
# allocation order / grouping of the two slabs against the placement regime of a fresh process's FIRST ring (no hunt): five fresh processes per mode, interleaved
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_41.txt
for round in 1 2 3 4 5; do
  for m in in_out out_in one spaced_kept spaced_freed; do
    timeout 300 python tools/alloc_order_lab.py $m 2>&1 | tail -1 >> $O
  done
done
sort $O | awk '{print}'; 
