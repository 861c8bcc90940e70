#!/bin/bash
# The XCD-aware work map of the 1:1 kernel (decoder option 5; used from 64 frames per launch) against the plain map, by frames
# per launch; same call, no placement hunt (every line a fresh process on whatever placement it gets).
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --no-cpu-baseline --no-smooth-leg --placement-tries 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-64s %8.1f Gpx/s (%7.1f .. %7.1f) frac %.4f  %8.2f us/launch  copy %s  %s' % (' '.join(sys.argv[1:]), d['value'], d['value_min'], d['value_max'], r['frac'], r['avg_launch_us'], r.get('same_run_copy_GBps'), d['parity_spot_check']))" "$@"; }
for round in 1 2; do
for f in ${FRAMES:-32 64 128 256}; do
  one --ring 256 --frames-per-launch $f --decoder-option 5=0
  one --ring 256 --frames-per-launch $f --decoder-option 5=1
done
done
