#!/bin/bash
# config 4 (8K -> 4K, decode_nv12_half_rep): encode side through the log-bucket table in 4 copies against the 26 KiB uniform table (head)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_31.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_rescale.py -m gpu -x -q 2>&1 | tail -4 >> $O
one() { python bench.py --workload 8k-half --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-70s %7.1f Gpx/s frac %.4f  %8.2f us/launch  %s' % (' '.join(sys.argv[1:]), d['value'], r['frac'], r['avg_launch_us'], d['parity_spot_check']))" "$@"; }
for round in 1 2 3; do
  one >> $O
  one --library tools/bin/libbt709hip_head.so >> $O
done
one --content flat >> $O
one --content flat --library tools/bin/libbt709hip_head.so >> $O
one --content smooth >> $O
one --content smooth --library tools/bin/libbt709hip_head.so >> $O
cat $O
