#!/bin/bash
# rgba16f packed-pair form, 1080p: row pairs per workgroup x workgroups per CU x launch length (all with bands from 8 frames)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_23.txt
L="tools/bin/libbt709hip_head.so tools/bin/f16p_band8.so tools/bin/f16p_band8_3wg.so tools/bin/f16p_band8_n2.so tools/bin/f16p_band8_n2_3wg.so tools/bin/f16p_band8_n4.so"
for n in 512 128 64 32 16 8; do
echo "## 1080p, $n frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
cat $O
