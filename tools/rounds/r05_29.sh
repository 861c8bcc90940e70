#!/bin/bash
# the 1:1 kernel: XCD map below 64 frames per launch?  same process, same ring (round 3 measured it process against process)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_29.txt
L="shipped tools/bin/band_min16.so tools/bin/band_min8.so"
for n in 8 16 32 48 64; do
  echo "## 4K, $n frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --ring 192 --per-launch $n --rounds 3 --tries 3 $L 2>&1 | grep -v slab >> $O
done
for n in 32 64 128; do
  echo "## 1080p, $n frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --width 1920 --height 1080 --ring 768 --per-launch $n --rounds 3 --tries 3 $L 2>&1 | grep -v slab >> $O
done
cat $O
