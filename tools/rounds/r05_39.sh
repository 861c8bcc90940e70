#!/bin/bash
# any-ratio kernel: 2 / 4 interleaved copies of the (now 5 KiB) encode-side table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_39.txt
for round in 1 2; do
for l in "" tools/bin/scaled_enc1.so tools/bin/scaled_enc2.so; do
  echo "## ${l:-shipped} (round $round)" >> $O
  timeout 600 bash tools/bench_scaled_set.sh $l >> $O 2>&1
done
done
cat $O
