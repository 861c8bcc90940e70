#!/bin/bash
# persistent 2:1 kernel: tile rows per step (BT709_REP_STEP).  Variant libraries are built in the container
# (python -m metalbt709decoder_amd.build --variant tools/bin/libbt709hip_u$u.so BT709_REP_STEP=$u) and
# timed against each other on the GPU box through bench.py --library; the in-tree library is never replaced.
cd "${GRAFT_REPO_ROOT:-.}"
for u in 1 2 3 4; do
  [ -f tools/bin/libbt709hip_u$u.so ] || python -m metalbt709decoder_amd.build --variant tools/bin/libbt709hip_u$u.so BT709_REP_STEP=$u >/dev/null
  python bench.py --library tools/bin/libbt709hip_u$u.so --workload 8k-half --no-cpu-baseline --steps 100 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step $u', d['value'], d['roofline']['avg_launch_us'])"
done
