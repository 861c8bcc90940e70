#!/bin/bash
# BASELINE config 5's per-GPU unit (SHARE frames per step, default 1 = the n = 8 point): HIP streams the steps rotate over,
# and the same fork / join pattern recorded into one graph with parallel branches (--graph --streams N).
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --workload 4k-batch8 --share ${SHARE:-1} --no-cpu-baseline --steps ${STEPS:-400} "$@" 2>&1 | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('%-40s %8.1f Gpx/s  (%7.1f .. %7.1f)  %7.3f us/step  %s  %s' % (' '.join(sys.argv[1:]), d['value'], d['value_min'], d['value_max'], d['ms_per_step']*1e3, r['kernel'], d['parity_spot_check']))
except Exception as e: print(' '.join(sys.argv[1:]), 'ERR', t[-3:])" "$@"; }
for r in 1 2; do
for n in ${STREAMS:-1 2 3 4 5 6 8}; do one --streams $n; done
done
for n in ${GRAPH_STREAMS:-1 2 3 4}; do one --graph --streams $n; done
