#!/bin/bash
# a LARGE spacer between the input and the output slab (freed afterwards): does distance in the allocator's sequence decide the regime?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_42.txt
for round in 1 2 3 4; do
  for m in in_out far32 far96 far160; do
    timeout 300 python tools/alloc_order_lab.py $m 2>&1 | tail -1 >> $O
  done
done
sort $O
