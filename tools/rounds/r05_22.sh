#!/bin/bash
# rgba16f packed-pair form: XCD bands for shorter launches?  (and the fixed tool: uploads waited for)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_22.txt
L="shipped tools/bin/libbt709hip_head.so tools/bin/f16p_band8.so tools/bin/f16p_band8_3wg.so tools/bin/f16p_3wg.so"
for n in 64 32 16 8; do
  echo "## 4K gamma 0, $n frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
for n in 512 32 8; do
echo "## 1080p, $n frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
cat $O
