#!/bin/bash
# encoder workgroup shape sweep (threads per tile x row pairs per workgroup), 4K, 32 pictures per launch
# usage: tools/encode_shapes.sh "192 256 320" "3 6 9"
for t in ${1:-192 256 320 384 512}; do for rp in ${2:-3 6 9 12}; do
  python tools/bench_encode.py --threads $t --row-pairs $rp --frames-per-launch 32 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('threads $t rowpairs $rp', d['frac_of_8TBps'], d['gpixel_per_s'])"
done; done
