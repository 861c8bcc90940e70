#!/bin/bash
# rgba16f: slices (several row-pair groups per workgroup share one staged table) for narrow frames
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_25.txt
timeout 900 python -m pytest tests/test_rgba16f.py -m gpu -x -q 2>&1 | tail -3 >> $O
# shipped = slices + narrow(4,3) rule; sl_n2 = slices, no narrow shape; sl1 = no slices (the previous commit); sl1_n2 = neither; f1 / f4 = slices kept while the launch has >= 1 / 4 workgroups per slot (default 2)
L="shipped tools/bin/f16p_sl_n2.so tools/bin/f16p_sl1.so tools/bin/f16p_sl1_n2.so tools/bin/f16p_sl_n2_f1.so tools/bin/f16p_sl_n2_f4.so tools/bin/libbt709hip_head.so"
for n in 512 128 32 16 8; do
echo "## 1080p, $n frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
for n in 1024 64 8; do
echo "## 720p, $n frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1280 --height 720 --ring 1024 --per-launch $n --rounds 3 --tries 3 $L >> $O 2>&1
done
echo "## 640x360, 2048 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 640 --height 360 --ring 2048 --per-launch 2048 --rounds 3 --tries 3 $L >> $O 2>&1
cat $O
