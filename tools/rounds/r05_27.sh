#!/bin/bash
# rgba16f: bare bucket index (no product) where one line covers the split bucket: sRGB; against the previous commit (scaled everywhere) and head
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_27.txt
timeout 900 python -m pytest tests/test_rgba16f.py -m gpu -x -q 2>&1 | tail -3 >> $O
for gamma in 1 0 3; do
  echo "## 4K gamma $gamma, 128 frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 128 --rounds 3 --tries 3 --gamma $gamma shipped tools/bin/f16p_sl1.so tools/bin/libbt709hip_head.so >> $O 2>&1
done
echo "## 4K gamma 1, 16 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 16 --rounds 3 --tries 3 --gamma 1 shipped tools/bin/f16p_sl1.so tools/bin/libbt709hip_head.so >> $O 2>&1
cat $O
