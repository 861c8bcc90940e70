#!/bin/bash
# 2:1 rescale: persistent (rep) vs per-tile kernel at small launch sizes (runs on the GPU box)
for fpl in 1 2 4 16; do for r in 0 1; do
  python bench.py --decoder-option 2=$r --workload 8k-half --frames-per-launch $fpl --no-cpu-baseline --steps 60 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fpl $fpl rep $r', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
done; done
