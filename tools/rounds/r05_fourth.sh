#!/bin/bash
# Round 5: ABI 500 on hardware -- the whole -m gpu suite (device identity, ring set, hunt budget, stream order across decoders,
# coalescing age limit), then bench.py: the driver's line, two ranks wrapped onto one GPU (refused, then --allow-shared-devices),
# the one-process launcher (--launcher threads) with two lanes on one GPU, and the hunt under 64 GB / frugal budgets.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05/pytest_abi500.log
B=gpurun_out/r05/bench
mkdir -p $B
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $B/bench_4k_driver_args.json 2> $B/bench_4k_driver_args.err
python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $B/bench_4k_2ranks_refused.json 2> $B/bench_4k_2ranks_refused.err; echo "rc=$?" >> $B/bench_4k_2ranks_refused.err
python3 bench.py --gpus 2 --steps 20 --warmup 5 --allow-shared-devices --ring 64 > $B/bench_4k_2ranks_shared.json 2> $B/bench_4k_2ranks_shared.err
python3 bench.py --gpus 2 --steps 20 --warmup 5 --allow-shared-devices --ring 64 --launcher threads > $B/bench_4k_threads_2lanes_shared.json 2> $B/bench_4k_threads_2lanes_shared.err
for budget in "--hunt-max-gb 64" "--hunt-frugal" "--hunt-max-gb 32"; do
  for rep in 1 2 3; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-smooth-leg $budget 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d['config']['placement']
print('$budget', 'frac %.4f first_allocation %.4f' % (d['roofline']['frac'], d['roofline']['first_allocation_frac']), 'hunt_ms %.0f peak %.1f GB budget %.1f GB evicted %d stopped_by %d candidates %s chosen %s prescan %s' % (p['hunt_ms'], p['peak_bytes'] / 1e9, p['budget_bytes'] / 1e9, p['evicted'], p['stopped_by'], p['candidates'], p['chosen'], p['output_prescan_GBps']))"
  done
done > gpurun_out/r05/hunt_budget.txt 2>&1
tail -5 gpurun_out/r05/pytest_abi500.log; cat gpurun_out/r05/hunt_budget.txt; for f in $B/*.json; do echo "== $f"; head -c 1500 $f; echo; done; tail -3 $B/*.err
