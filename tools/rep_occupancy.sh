#!/bin/bash
# persistent 2:1 kernel: one workgroup per CU with 16 table copies vs two per CU with 8 (more waves, more conflicts)
# args: workgroups, LDS KiB per workgroup  (bt709hip_decoder_option 3 and 4)
run() { echo "== workgroups $1 lds_kb $2"; python bench.py --decoder-option 3=$1 --decoder-option 4=$2 --workload 8k-half --no-cpu-baseline --steps 80 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])"; }
run 256 160
run 512 80
run 512 78
run 256 80
run 768 52
run 256 160
