#!/bin/bash
# NOTE: the BT709_LAB_PRIO_* gates were removed from bt709_kernels.hip at the end of round 3 (both variants lost); check out commit f9775f7 to re-run this.
# Same-call A/B of s_setprio placements in the short-lived 1:1 kernel (variant libraries built with
# python -m metalbt709decoder_amd.build --variant tools/bin/libbt709hip_prio_*.so BT709_LAB_PRIO_LOADS | BT709_LAB_PRIO_STORES=n)
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --no-cpu-baseline --no-smooth-leg "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-60s %8.1f Gpx/s  frac %.4f  %7.2f us  %s  %s' % (' '.join(sys.argv[1:]), d['value'], r['frac'], r['avg_launch_us'], r['kernel'], d['parity_spot_check']))" "$@"; }
for round in 1 2; do
one
one --library tools/bin/libbt709hip_prio_loads.so
one --library tools/bin/libbt709hip_prio_stores1.so
one --library tools/bin/libbt709hip_prio_stores3.so
done
one
