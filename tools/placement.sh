#!/bin/bash
# 1:1 kernel: does the placement of rows / frames in memory matter? (runs on the GPU box)
run() { echo "== $*"; env "$@" timeout 120 tools/bin/decode_lab_fma 0 5 | head -1; }
run LAB_ROW_PAD=0
run LAB_ROW_PAD=256
run LAB_ROW_PAD=1024
run LAB_ROW_PAD=4096
run LAB_IN_ROW_PAD=256
run LAB_IN_ROW_PAD=256 LAB_ROW_PAD=1024
run LAB_FRAME_PAD=4096
run LAB_FRAME_PAD=65536
run LAB_FRAME_PAD=1048576
run LAB_ROW_PAD=0
