#!/bin/bash
# rgba16f: packed-pair candidate/settle form against HEAD's kernel, same process, same ring
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rgba16f.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_14_tests.txt
for gamma in 0 1 3 2; do
  echo "## gamma $gamma, 128 frames per launch" >> gpurun_out/r05_14_ab.txt
  timeout 600 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 128 --rounds 3 --tries 3 --gamma $gamma shipped tools/bin/libbt709hip_head.so >> gpurun_out/r05_14_ab.txt 2>&1
done
for n in 16 1; do
  echo "## gamma 0, $n frames per launch" >> gpurun_out/r05_14_ab.txt
  timeout 600 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $n --rounds 3 --tries 3 shipped tools/bin/libbt709hip_head.so >> gpurun_out/r05_14_ab.txt 2>&1
done
cat gpurun_out/r05_14_tests.txt gpurun_out/r05_14_ab.txt
