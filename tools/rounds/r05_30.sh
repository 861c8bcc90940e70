#!/bin/bash
# LINEAR mode through a log-bucket table (5 KiB) in the plain 1:1 kernel against the 33 KiB table in the rows kernel (head)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_30.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py -m gpu -x -q 2>&1 | tail -4 >> $O
for n in 256 64 32 8 1; do
  echo "## 4K gamma 2 (LINEAR), $n frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --ring 256 --per-launch $n --rounds 3 --tries 3 --gamma 2 shipped tools/bin/libbt709hip_head.so 2>&1 | grep -v slab >> $O
done
echo "## 1080p gamma 2, 1024 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --width 1920 --height 1080 --ring 1024 --per-launch 1024 --rounds 3 --tries 3 --gamma 2 shipped tools/bin/libbt709hip_head.so 2>&1 | grep -v slab >> $O
echo "## 4K gamma 0 (Apple: unchanged kernel), 256 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --ring 256 --per-launch 256 --rounds 3 --tries 3 --gamma 0 shipped tools/bin/libbt709hip_head.so 2>&1 | grep -v slab >> $O
cat $O
