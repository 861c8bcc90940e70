#!/bin/bash
# Round 5: the RGBA16F kernel in its straight-line shape -- parity first (exhaustive half-float sweeps, two-pass tests), then the
# shapes against each other and against round 4's walking shape in ONE process on ONE ring (tools/ab_libs.py --format rgba16f).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q -k "rgba16f or two_pass or half_lookup or render or multi_device or ring" 2>&1 | tail -5 > gpurun_out/r05/pytest_f16.log
{
echo "# RGBA16F target, 4K, ring 128, ONE process, ONE ring, alternating regions (tools/ab_libs.py --format rgba16f); fraction = 9.5 B per pixel / 8 TB/s"
for per in 128 16 1; do
  echo "## $per frames per launch"
  python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $per --rounds 3 --steps 10 --tries 4 shipped tools/bin/lab_f16_oldshape.so \
    tools/bin/lab_f16_nb2_rp1.so tools/bin/lab_f16_nb2_rp3.so tools/bin/lab_f16_nb2_rp4.so tools/bin/lab_f16_nb1_rp4.so tools/bin/lab_f16_nb4_rp1_512.so tools/bin/lab_f16_nb4_rp2_512.so tools/bin/lab_f16_nb2_rp2_b6.so \
     2>&1 | grep -v "^input slab\|^output slab"
done
echo "## other gammas, 128 per launch"
for g in 1 3 2; do python tools/ab_libs.py --format rgba16f --gamma $g --ring 128 --per-launch 128 --rounds 3 --steps 10 --tries 4 shipped tools/bin/lab_f16_oldshape.so 2>&1 | grep -v "^input slab\|^output slab"; done
} > gpurun_out/r05/ab_rgba16f_shapes.txt 2>&1
cat gpurun_out/r05/pytest_f16.log gpurun_out/r05/ab_rgba16f_shapes.txt
