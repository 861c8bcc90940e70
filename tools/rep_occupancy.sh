#!/bin/bash
# persistent 2:1 kernel: one workgroup per CU with 16 table copies vs two per CU with 8 (more waves, more conflicts)
run() { echo "== $*"; env "$@" python bench.py --workload 8k-half --no-cpu-baseline --steps 80 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])"; }
run BT709HIP_REP_WORKGROUPS=256 BT709HIP_REP_LDS_KB=160
run BT709HIP_REP_WORKGROUPS=512 BT709HIP_REP_LDS_KB=80
run BT709HIP_REP_WORKGROUPS=512 BT709HIP_REP_LDS_KB=78
run BT709HIP_REP_WORKGROUPS=256 BT709HIP_REP_LDS_KB=80
run BT709HIP_REP_WORKGROUPS=768 BT709HIP_REP_LDS_KB=52
run BT709HIP_REP_WORKGROUPS=256 BT709HIP_REP_LDS_KB=160
