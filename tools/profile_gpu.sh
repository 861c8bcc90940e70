#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root.  Collects, for one bench workload:
#   1. rocprofv3 --kernel-trace --stats           (per-kernel durations)
#   2. separate --pmc passes (never combined with tracing domains other than kernel-trace):
#      FETCH_SIZE | WRITE_SIZE | SQ LDS/VALU counters | SQ wait counters | TCC hit/miss
# into gpurun_out/prof_<tag>/.  tools/pmc_summary.py turns those into profiles/*.
# usage: tools/profile_gpu.sh <tag> [bench.py args...]
#        PROFILE_PROG=tools/bench_encode.py tools/profile_gpu.sh encode [its args...]   (any program
#        that takes --steps/--warmup)
set -u
TAG=${1:-4k}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ -n "${PROFILE_PROG:-}" ]; then BENCH="python3 $REPO/$PROFILE_PROG $*"; else BENCH="python3 $REPO/bench.py --no-cpu-baseline $*"; fi

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH --steps 100 --warmup 10 > "$OUT/trace.log" 2>&1

pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH --steps 6 --warmup 2 > "$OUT/pmc_$name.log" 2>&1
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
pmc tcc TCC_HIT_sum TCC_MISS_sum
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
find "$OUT" -name "*.csv" | head -50 > "$OUT/files.txt"
echo done
