#!/bin/bash
# rgba16f packed-pair form: work shapes again (the arithmetic is cheaper now: does a workgroup that covers more pixels per staged table win?)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_16_shapes.txt
L="shipped tools/bin/f16p_4_3_512.so tools/bin/f16p_4_4_512.so tools/bin/f16p_2_4_960.so tools/bin/f16p_2_6_960.so tools/bin/f16p_8_2_256.so tools/bin/f16p_8_3_256.so tools/bin/lab_f16p_noarith.so"
for n in 128 16 1; do
  echo "## gamma 0, $n frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
echo "## 1080p, 512 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch 512 --rounds 3 --tries 3 $L >> $O 2>&1
cat $O
