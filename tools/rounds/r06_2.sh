#!/bin/bash
# five fresh bench.py processes on the round-6 default hunt (frugal), then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/r06_hunt_default.jsonl; rm -f $O
for i in 1 2 3 4 5; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 >> $O 2>gpurun_out/r06_hunt_default.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r06_hunt_default.jsonl"):
    d=json.loads(l); p=d["config"]["placement"]; r=d["roofline"]
    print("frac %.4f first_allocation %.4f hunt_ms %.0f peak %.1f GB budget %.1f GB evicted %d stopped_by %d candidates %s chosen %s value %.1f cpu %s" % (
        r["frac"], r["first_allocation_frac"], p["hunt_ms"], p["peak_bytes"]/1e9, p["budget_bytes"]/1e9, p["evicted"], p["stopped_by"], p["candidates"], p["chosen"], d["value"],
        json.dumps(d.get("cpu_baseline",{}).get("frames"))))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_2_tests.txt 2>&1; tail -4 gpurun_out/r06_2_tests.txt
