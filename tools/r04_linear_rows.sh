#!/bin/bash
# Round 4: the LINEAR mode's 33 KiB table shared by two row pairs per workgroup (1024 lanes, shipped) against one (BT709_BIG_TABLE_ROWS=1),
# 4K x 256 / 32 per launch, product ring with 4 candidates per slab, alternating fresh processes.
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests -m gpu -x -q -k "exhaustive or sweep or xcd or geometry or random_frames or ragged or Linear or linear" 2>&1 | tail -3
for round in 1 2 3; do
  echo "shipped (2 row pairs per workgroup):"; ONLY_GAMMA=2 python tools/bench_alpha11.py - 256 4; ONLY_GAMMA=2 PER_LAUNCH=32 python tools/bench_alpha11.py - 256 4
  echo "one row pair per workgroup:";          ONLY_GAMMA=2 python tools/bench_alpha11.py tools/bin/libbt709hip_rows1.so 256 4; ONLY_GAMMA=2 PER_LAUNCH=32 python tools/bench_alpha11.py tools/bin/libbt709hip_rows1.so 256 4
done
