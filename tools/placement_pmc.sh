#!/bin/bash
# Lab (GPU box): counters of the 256-frame launch per output slab, one rocprofv3 --pmc pass (= one process, one placement) per counter set.
#   tools/placement_pmc.sh > gpurun_out/placement_pmc.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_placement
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
N=8; K=6
i=0
for P in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
         "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum" \
         "TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_CYCLE_sum"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/tools/placement_pmc.py" $N $K > "$OUT/p$i.log" 2>&1
  echo "== pass $i: $P"
  grep "^SLAB" "$OUT/p$i.log"
  python3 - "$OUT/p$i" $K <<'PY'
import csv, glob, sys, collections
out, K = sys.argv[1], int(sys.argv[2])
for f in sorted(glob.glob(out + "/**/pmc_counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "decode_nv12_quads" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    slab_of = {d: i // K for i, d in enumerate(ids)}
    first = {d for i, d in enumerate(ids) if i % K == 0}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in first:
            continue  # the slab's warm launch
        acc[r["Counter_Name"]][slab_of[d]].append(float(r["Counter_Value"]))
    for name in sorted(acc):
        print("%-40s %s" % (name, " ".join("%12.0f" % (sum(v) / len(v)) for _, v in sorted(acc[name].items()))))
PY
  i=$((i+1))
done
rm -rf "$OUT"
