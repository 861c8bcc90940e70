#!/bin/bash
# Ceilings for the persistent 2:1 kernel (review item 3 of round 2), same call: stub builds with WRONG output that remove
# work outright, so that the time they save bounds every exact formulation of the same idea from above.
#   python tools/lab_variants.py tools/bin/libbt709hip_bound_<X>.so BT709_LAB_BOUND_<X>   (round 4: the gates live in tools/lab_variants.py, not in csrc/)
# The parity spot check of a stub build fails by construction (bench.py then prints value null and exits 1); the launch time is what is read.
cd "${GRAFT_REPO_ROOT:-.}"
half() { python bench.py --workload 8k-half --no-cpu-baseline --steps 40 "$@" 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('%-58s %8.2f us per 16-frame launch  %6.1f GB/s  frac %.4f  spot check: %s' % (' '.join(sys.argv[1:]) or 'shipped', r['avg_launch_us'], r['achieved'], r['frac'], d['parity_spot_check']))" "$@"; }
for round in 1 2; do
  half
  half --library tools/bin/libbt709hip_bound_SHARED_INDEX.so
  half --library tools/bin/libbt709hip_bound_ONE_ENCODE.so
  half --library tools/bin/libbt709hip_bound_both.so
done
half
