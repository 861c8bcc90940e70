#!/bin/bash
# rgba16f: the index scale as two v_mul_f32 instead of one v_pk_mul_f32
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_33.txt
for gamma in 0 1; do
  echo "## 4K gamma $gamma, 128 frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 128 --rounds 3 --tries 3 --gamma $gamma shipped tools/bin/f16p_scalar_mul.so 2>&1 | grep -v slab >> $O
done
cat $O
