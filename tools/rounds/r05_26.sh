#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 5101 5102 5103; do timeout 1200 python tools/soak.py $seed 350 2>&1 | tail -2; done > gpurun_out/r05_26_soak.txt
cat gpurun_out/r05_26_soak.txt
