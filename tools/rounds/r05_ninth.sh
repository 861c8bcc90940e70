#!/bin/bash
# Round 5: the shipped RGBA16F kernel (two straight-line shapes chosen by launch size): parity, then against round 4's walking shape and
# against its own ceilings (arithmetic deleted / conversion only), one process, one ring.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q -k "rgba16f or two_pass or half_lookup or render" 2>&1 | tail -5 > gpurun_out/r05/pytest_f16_shipped.log
{
echo "# RGBA16F target, 4K, ring 128, ONE process, ONE ring, alternating regions (tools/ab_libs.py --format rgba16f); fraction = 9.5 B per pixel / 8 TB/s"
echo "# shipped = (4 blocks x 2 row pairs, 512-lane tiles), (2 x 3, 960) for launches under 4 workgroups per CU; oldshape = round 4's walking kernel (512 lanes, up to 16 row pairs, one-ahead prefetch)"
for per in 128 32 16 8 4 2 1; do
  echo "## $per frames per launch"
  python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $per --rounds 3 --steps 10 --tries 4 shipped tools/bin/lab_f16_oldshape.so \
    tools/bin/lab_f16_noarith.so tools/bin/lab_f16_noarith_notable.so tools/bin/lab_f16_cvtonly.so 2>&1 | grep -v "^input slab\|^output slab"
done
echo "## other gammas (sRGB, ITU-709, Linear), 128 per launch"
for g in 1 3 2; do python tools/ab_libs.py --format rgba16f --gamma $g --ring 128 --per-launch 128 --rounds 3 --steps 10 --tries 4 shipped tools/bin/lab_f16_oldshape.so 2>&1 | grep -v "^input slab\|^output slab"; done
} > gpurun_out/r05/ab_rgba16f_shipped.txt 2>&1
python tools/bench_scaled.py --path rgba16f --ring 128 --frames-per-launch 128 > gpurun_out/r05/bench_rgba16f_128.json 2>/dev/null
python tools/bench_scaled.py --path rgba16f --frames-per-launch 16 > gpurun_out/r05/bench_rgba16f_16.json 2>/dev/null
python tools/bench_scaled.py --path rgba16f --frames-per-launch 1 > gpurun_out/r05/bench_rgba16f_1.json 2>/dev/null
cat gpurun_out/r05/pytest_f16_shipped.log gpurun_out/r05/ab_rgba16f_shipped.txt gpurun_out/r05/bench_rgba16f_*.json
