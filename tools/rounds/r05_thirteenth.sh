#!/bin/bash
# Round 5: does the small-table (Apple, ITU-709) 1:1 launch gain from several row pairs per workgroup too?  One process, one ring.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
{
echo "# 4K 1:1, ring 256 x 256 per launch, ONE process, ONE ring, alternating regions (tools/ab_libs.py): shipped (one row pair per workgroup for tables up to 16 KiB)"
echo "# against builds that send EVERY table mode through decode_nv12_quads_rows with 4 / 2 row pairs per workgroup under the XCD-aware map"
for g in 0 3 2; do
  echo "## gamma $g"
  python tools/ab_libs.py --gamma $g --ring 256 --per-launch 256 --rounds 5 --steps 10 --tries 4 shipped tools/bin/lab_apple_rp4.so tools/bin/lab_apple_rp2.so 2>&1 | grep -v "^input slab\|^output slab"
done
} > gpurun_out/r05/ab_small_table_rows.txt 2>&1
cat gpurun_out/r05/ab_small_table_rows.txt
