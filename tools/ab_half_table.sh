#!/bin/bash
# Lab: what the persistent 2:1 kernel would gain from decode-side table entries of 8 bytes instead of 16 (16 copies then fit twice
# into a CU's LDS = two workgroups per CU, 8 waves per SIMD).  The stub build halves the bucket count instead (WRONG OUTPUT, same
# accesses, same instructions).  args of run: library, workgroups, LDS KiB per workgroup (decoder options 3 and 4)
cd "${GRAFT_REPO_ROOT:-.}"
run() { printf "%-40s workgroups %4s lds_kb %3s: " "$1" "$2" "$3"; python bench.py ${1:+--library $1} --decoder-option 3=$2 --decoder-option 4=$3 --workload 8k-half --no-cpu-baseline --placement-tries 1 --steps 60 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('%.1f us/launch  frac %.4f  %s' % (r['avg_launch_us'], r['frac'], d['parity_spot_check'][:12]))"; }
for rep in 1 2; do
run "" 256 160
run "" 512 80
run tools/bin/libbt709hip_halftable.so 256 160
run tools/bin/libbt709hip_halftable.so 512 80
run tools/bin/libbt709hip_halftable.so 768 53
run tools/bin/libbt709hip_halftable.so 512 60
done
