#!/bin/bash
# encoder, ONE 4K picture per launch (the shape a caller of +convertIntoCoreVideoBuffer: gets): threads per tile x row pairs per workgroup
cd "${GRAFT_REPO_ROOT:-.}"
for t in ${1:-128 192 256 320 512}; do for rp in ${2:-1 2 3 4 6}; do
  python tools/bench_encode.py --threads $t --row-pairs $rp --frames-per-launch 1 --steps 200 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('threads $t rowpairs $rp', d['frac_of_8TBps'], d['gpixel_per_s'])"
done; done
python tools/bench_encode.py --frames-per-launch 1 --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('default', d['frac_of_8TBps'], d['gpixel_per_s'])"
