#!/bin/bash
# config 4: centre_norm of two bytes per v_pk_fma_f32 in the persistent 2:1 kernel (12 instructions fewer per 4 output pixels)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_32.txt
:
one() { python bench.py --workload 8k-half --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s %7.1f Gpx/s frac %.4f  %8.2f us/launch  %s' % (' '.join(sys.argv[1:]), d['value'], r['frac'], r['avg_launch_us'], d['parity_spot_check']))" "$@"; }
for round in 1 2 3; do one >> $O; one --library tools/bin/libbt709hip_head2.so >> $O; done
one --content flat >> $O; one --content flat --library tools/bin/libbt709hip_head2.so >> $O
one --content smooth >> $O
cat $O
