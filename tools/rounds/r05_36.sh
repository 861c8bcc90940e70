#!/bin/bash
# any-ratio kernel: conflict-free decode-side copies WITH large workgroups (strips x copies), which round 3 measured only separately
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_36.txt
for l in "" tools/bin/scaled_s4c4.so tools/bin/scaled_s2c3.so tools/bin/scaled_s4c3.so tools/bin/scaled_s3c4.so tools/bin/scaled_s4c0.so; do
  echo "## ${l:-shipped}" >> $O
  timeout 600 bash tools/bench_scaled_set.sh $l >> $O 2>&1
done
cat $O
