#!/bin/bash
# Round 5: RGBA16F straight-line shapes, wider sweep: (blocks per lane, row pairs per workgroup, lanes per tile), one process, one ring.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
{
echo "# RGBA16F target, 4K, ring 128, ONE process, ONE ring, alternating regions (tools/ab_libs.py --format rgba16f); variants = NB_RP_LANES; fraction = 9.5 B per pixel / 8 TB/s"
for per in 128 16 1; do
  echo "## $per frames per launch"
  python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $per --rounds 3 --steps 10 --tries 4 tools/bin/lab_f16_oldshape.so tools/bin/lab_f16_[A-L]_*.so 2>&1 | grep -v "^input slab\|^output slab"
done
} > gpurun_out/r05/ab_rgba16f_shapes2.txt 2>&1
cat gpurun_out/r05/ab_rgba16f_shapes2.txt
