#!/bin/bash
# rgba16f: table staging by LDS DMA (global_load_lds_dwordx4) against the VGPR round trip
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_37.txt
for gamma in 0 1 3; do
  echo "## 4K gamma $gamma, 128 frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 128 --rounds 3 --tries 3 --gamma $gamma shipped tools/bin/lab_f16p_dma.so 2>&1 | grep -v slab >> $O
done
echo "## 4K gamma 0, 16 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 16 --rounds 3 --tries 3 shipped tools/bin/lab_f16p_dma.so 2>&1 | grep -v slab >> $O
echo "## 1080p gamma 0, 512 frames per launch" >> $O
timeout 900 python tools/ab_libs.py --format rgba16f --width 1920 --height 1080 --ring 512 --per-launch 512 --rounds 3 --tries 3 shipped tools/bin/lab_f16p_dma.so 2>&1 | grep -v slab >> $O
cat $O
