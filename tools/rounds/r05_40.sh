#!/bin/bash
# frame ring with RGBA16F targets: parity, and the RGBA16F path timed on the product's own hunt (three fresh processes)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_40.txt
timeout 1500 python -m pytest tests/test_rgba16f.py tests/test_multi_device.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 >> $O
for i in 1 2 3; do
  python tools/bench_scaled.py --path rgba16f --ring 128 --frames-per-launch 128 2>&1 | tail -1 | cut -c1-330 >> $O
done
python tools/bench_scaled.py --path rgba16f --ring 128 --frames-per-launch 128 --placement-tries 1 2>&1 | tail -1 | cut -c1-330 >> $O
python tools/bench_scaled.py --path rgba16f --frames-per-launch 16 2>&1 | tail -1 | cut -c1-330 >> $O
python tools/bench_scaled.py --path rgba16f --frames-per-launch 1 2>&1 | tail -1 | cut -c1-330 >> $O
cat $O
