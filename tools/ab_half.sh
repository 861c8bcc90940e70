#!/bin/bash
# Same-call A/B of the persistent 2:1 kernel (runs on the GPU box): in-tree library vs a variant build.
# usage: tools/ab_half.sh tools/bin/libbt709hip_<variant>.so [rounds]
cd "${GRAFT_REPO_ROOT:-.}"
half() { python bench.py --workload 8k-half --no-cpu-baseline --steps 40 "$@" 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Gpx/s', d['roofline']['avg_launch_us'], 'us', d['roofline']['frac'], d['parity_spot_check'])"; }
for round in $(seq 1 ${2:-3}); do
  echo "== half in-tree (round $round)"; half
  echo "== half $1 (round $round)"; half --library $1
done
