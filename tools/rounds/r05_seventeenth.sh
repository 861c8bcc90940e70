#!/bin/bash
# rgba16f packed-pair form: narrow shape for 1080p rows; workgroups per CU (LDS floor) at 16 / 128 frames per launch
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_17.txt
timeout 900 python -m pytest tests/test_rgba16f.py -m gpu -x -q 2>&1 | tail -3 >> $O
L="shipped tools/bin/libbt709hip_head.so tools/bin/f16p_3wg.so tools/bin/f16p_2wg.so tools/bin/lab_f16p_noarith.so"
for n in 128 32 16 8; do
  echo "## 4K gamma 0, $n frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
echo "## 1080p, 512 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch 512 --rounds 3 --tries 3 $L tools/bin/f16p_4_3_512.so >> $O 2>&1
echo "## 1080p, 32 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch 32 --rounds 3 --tries 3 $L tools/bin/f16p_4_3_512.so >> $O 2>&1
echo "## 720p, 1024 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1280 --height 720 --ring 1024 --per-launch 1024 --rounds 3 --tries 3 $L tools/bin/f16p_4_4_512.so >> $O 2>&1
cat $O
