#!/bin/bash
# eight ranks (the driver's N = 8 command shape) wrapped onto this one GPU: rendezvous, per-rank spot checks, the line -- functional only
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --allow-shared-devices --ring 48 --placement-tries 1 > gpurun_out/r05_35_8ranks.json 2> gpurun_out/r05_35_8ranks.err; echo "rc=$?" >> gpurun_out/r05_35_8ranks.err
timeout 900 python bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --allow-shared-devices --ring 48 --placement-tries 1 --launcher threads > gpurun_out/r05_35_8lanes.json 2> gpurun_out/r05_35_8lanes.err; echo "rc=$?" >> gpurun_out/r05_35_8lanes.err
tail -n 3 gpurun_out/r05_35_8ranks.err; tail -n 3 gpurun_out/r05_35_8lanes.err; python - <<'PY'
import json
for f in ("gpurun_out/r05_35_8ranks.json","gpurun_out/r05_35_8lanes.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["n_gpus"], d["ranks"], d["shared_devices"], d["parity_spot_check"], d["parity_spot_check_ranks"], len(d["per_rank"]), d["config"]["launcher"])
    except Exception as e: print(f, "ERR", e)
PY
