#!/bin/bash
# Round 4 (VERDICT item 3 i): DRAM-side and translation-side counters of the 256-frame launch per output slab -- N slabs in ONE
# process (both regimes usually present), K launches each, one rocprofv3 --pmc pass per counter set (a pass = a process = its own
# placement; every pass prints its own per-slab rates, the comparison is slow vs fast slabs WITHIN a pass).
#   tools/placement_pmc_r04.sh > gpurun_out/r04_placement_pmc.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_placement
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
N=${N:-8}; K=${K:-4}
i=0
# PASSES=tcc (default): the memory-side sets (fast: a few TCC / GRBM instances); PASSES=tcp: the translation sets, which take
# ~7 minutes each under rocprofv3 (per-CU counters) -- they ran once (profiles/r04_placement_pmc.txt, passes T0 / T1)
if [ "${PASSES:-tcc}" = tcp ]; then
SETS=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
      "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum")
else
SETS=("BT709_WRREQ_MAX BT709_WRREQ_MIN TCC_EA0_WRREQ_sum"
      "BT709_WRSTALL_MAX BT709_WRSTALL_MIN TCC_EA0_WRREQ_STALL_sum"
      "TCC_BUBBLE_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_64B_sum"
      "BT709_RDREQ_MAX BT709_RDREQ_MIN BT709_TCCBUSY_MAX BT709_TCCBUSY_MIN"
      "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE GRBM_TA_BUSY")
fi
for P in "${SETS[@]}"; do
  timeout ${PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc $P -E "$REPO/tools/placement_extra_counters.yaml" --output-format csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/tools/placement_pmc.py" $N $K > "$OUT/p$i.log" 2>&1
  echo "== pass $i: $P"
  grep "^SLAB" "$OUT/p$i.log" || tail -5 "$OUT/p$i.log"
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
        print("%-46s %s" % (name, " ".join("%13.0f" % (sum(v) / len(v)) for _, v in sorted(acc[name].items()))))
PY
  i=$((i+1))
done
rm -rf "$OUT"
