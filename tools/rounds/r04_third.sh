#!/bin/bash
# round 4, third GPU call: the software-pipelined 2:1 kernel (parity, A/B against the unpipelined build, three contents), the
# memory-side placement counters, the 8k-half profile of the new kernel
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "half or config4 or rescale or scaled or pass2 or smoke or pattern" > $O/pytest_half.log 2>&1; echo "pytest rc=$?" >> $O/pytest_half.log; tail -4 $O/pytest_half.log
half() { timeout 300 python bench.py --workload 8k-half --no-cpu-baseline --steps 40 "$@" 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('%-70s %8.2f us per 16-frame launch  %6.1f GB/s  frac %.4f  smooth %s  spot check: %s' % (' '.join(sys.argv[1:]) or 'shipped (pipelined)', r['avg_launch_us'], r['achieved'], r['frac'], (r.get('smooth_content') or {}).get('frac'), d['parity_spot_check']))" "$@"; }
{ for round in 1 2; do
  half
  half --library tools/bin/libbt709hip_nopipe.so
  done
  half --content flat
  half --content flat --library tools/bin/libbt709hip_nopipe.so
  half --content smooth
  half --content smooth --library tools/bin/libbt709hip_nopipe.so
} > $O/ab_half_pipeline.txt 2>&1; cat $O/ab_half_pipeline.txt
PASS_TIMEOUT=300 timeout 1500 tools/placement_pmc_r04.sh > $O/placement_pmc_tcc.txt 2>&1; cat $O/placement_pmc_tcc.txt
timeout 900 tools/profile_gpu.sh 8k-half --workload 8k-half > /dev/null 2>&1
grep '^{' gpurun_out/prof_8k-half/trace.log | tail -1 > $O/profiled_run_8k-half.json
python tools/pmc_summary.py gpurun_out/prof_8k-half r04 8k-half > $O/pmc_8k-half.json
mkdir -p $O/profiles; cp profiles/r04_8k-half* $O/profiles/
rm -rf gpurun_out/prof_* gpurun_out/pmcq_*
