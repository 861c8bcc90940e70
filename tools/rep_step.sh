#!/bin/bash
# persistent 2:1 kernel: tile rows per step (BT709_REP_STEP), rebuilt in place on the GPU box
cd "${GRAFT_REPO_ROOT:-.}"
SRC="metalbt709decoder_amd/csrc"
for u in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -shared \
    -DBT709_REP_STEP=$u $SRC/bt709_kernels.hip $SRC/bt709_rescale.hip $SRC/bt709_encode.hip $SRC/bt709_planes.hip $SRC/bt709hip.cpp \
    $SRC/transfer_tables.cpp -o metalbt709decoder_amd/libbt709hip.so 2>/dev/null
  python bench.py --workload 8k-half --no-cpu-baseline --steps 100 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step $u', d['value'], d['roofline']['avg_launch_us'])"
done
