#!/bin/bash
# Round 5: the 1:1 kernel's big-table form (LINEAR mode, 33 KiB table): RP row pairs per workgroup, every load first (straight line),
# against one row pair per workgroup; parity of the whole suite first (the sweeps of gamma 2 run through the new kernel).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05/pytest_quads_rows.log
{
echo "# LINEAR mode (gamma 2, 4 096-bucket table = 33 KiB staged per workgroup), 4K 1:1, ring 256, ONE process, ONE ring, alternating regions (tools/ab_libs.py --gamma 2)"
echo "# shipped = decode_nv12_quads_rows, 2 row pairs per workgroup; lin_rp1 = one row pair per workgroup (rounds 1-4); rp3 / rp4"
for per in 256 32; do
  echo "## $per frames per launch"
  python tools/ab_libs.py --gamma 2 --ring 256 --per-launch $per --rounds 4 --steps 10 --tries 4 shipped tools/bin/lab_lin_rp1.so tools/bin/lab_lin_rp3.so tools/bin/lab_lin_rp4.so 2>&1 | grep -v "^input slab\|^output slab"
done
echo "## Apple mode for reference (4 KiB table: decode_nv12_quads in all four libraries)"
python tools/ab_libs.py --gamma 0 --ring 256 --per-launch 256 --rounds 3 --steps 10 --tries 4 shipped tools/bin/lab_lin_rp1.so 2>&1 | grep -v "^input slab\|^output slab"
} > gpurun_out/r05/ab_linear_rows.txt 2>&1
cat gpurun_out/r05/pytest_quads_rows.log gpurun_out/r05/ab_linear_rows.txt
