#!/bin/bash
# One-off counter passes for a program (runs on the GPU box):
#   tools/pmc_quick.sh <tag> <kernel substring> "<counters pass 1>" ["<counters pass 2>" ...] -- python3 prog args...
# prints the per-launch mean of every counter for the kernels whose name contains the substring.
TAG=$1; KERNEL=$2; shift 2
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
PROG=("$@"); PROG[1]="$REPO/${PROG[1]}"
cd /tmp && export TMPDIR=/tmp
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- "${PROG[@]}" > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections
out, kernel = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/p*/**/pmc_counter_collection.csv", recursive=True)):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc): print("%-40s %16.0f  (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
