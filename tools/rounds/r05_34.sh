#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 5201 5202 5203 5204; do timeout 1500 python tools/soak.py $seed 600 2>&1 | tail -1; done > gpurun_out/r05_34_soak.txt
cat gpurun_out/r05_34_soak.txt
