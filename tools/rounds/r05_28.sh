#!/bin/bash
# the 1:1 kernel (BGRA8 target) by frame width: does it lose on narrow frames as the RGBA16F kernel did?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_28.txt
for cfg in "3840 2160 256 256" "1920 1080 1024 1024" "1280 720 2048 2048" "640 360 8192 8192" "1920 1080 1024 32" "1280 720 2048 32" "640 360 8192 64"; do
  set -- $cfg
  echo "## $1 x $2, ring $3, $4 frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --width $1 --height $2 --ring $3 --per-launch $4 --rounds 3 --tries 3 shipped 2>&1 | grep -v slab >> $O
done
cat $O
