#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rescale.py -m gpu -x -q > gpurun_out/r06_1_tests.txt 2>&1; tail -5 gpurun_out/r06_1_tests.txt
O=gpurun_out/r06_ab_scaled_share.txt; rm -f $O
ENL="shipped(once)=;per-lane(r5)=tools/bin/scaled_no_once.so;once+persistent=tools/bin/scaled_once_persistent.so;once+LDS-tile=tools/bin/scaled_once_lds.so;stub:half-taps=tools/bin/scaled_half_fewer.so;stub:pair-DPP=tools/bin/scaled_pair_dpp.so"
bash tools/ab_scaled.sh $O "1920 1080 3840 2160 8 64;1920 1080 3840 2160 1 64;1920 1080 2560 1440 8 64;1280 720 3840 2160 8 64" "$ENL"
bash tools/ab_scaled.sh $O "3840 2160 3840 2160 8 32" "shipped=;stub:half-taps=tools/bin/scaled_half_fewer.so;stub:pair-DPP=tools/bin/scaled_pair_dpp.so"
bash tools/ab_scaled.sh $O "3840 2160 2560 1440 8 64" "shipped=;stub:quarter-taps=tools/bin/scaled_quarter_fewer.so;stub:half-taps=tools/bin/scaled_half_fewer.so;stub:pair-DPP=tools/bin/scaled_pair_dpp.so"
cat $O; tail -5 $O.err
