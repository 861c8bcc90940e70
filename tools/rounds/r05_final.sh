#!/bin/bash
# Round 5, last call: the whole -m gpu suite and the driver's bench line on HEAD, the per-gamma 1:1 rates (the LINEAR mode now runs
# decode_nv12_quads_rows), smoke().
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_4k_driver_args.json 2> $O/bench_4k_driver_args.err; head -c 600 $O/bench_4k_driver_args.json; echo
{ python tools/bench_alpha11.py - 256 4; PER_LAUNCH=32 python tools/bench_alpha11.py - 256 4; } > $O/bench_alpha.txt 2>&1; cat $O/bench_alpha.txt
python tools/bench_scaled.py --path rgba16f --ring 128 --frames-per-launch 128 > $O/bench_rgba16f.json 2>/dev/null; cat $O/bench_rgba16f.json
