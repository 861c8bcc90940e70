#!/bin/bash
# Same-call A/B of the bucket-index form (runs on the GPU box): round-to-nearest add (default build)
# vs round 1's round-toward-zero add between two s_setreg (-DBT709_INDEX_RTZ), interleaved.
#   1:1 kernel: tools/bin/decode_lab_rne vs decode_lab_rtz  (built by: see tools/decode_lab.hip header, +/- -DBT709_INDEX_RTZ)
#   2:1 kernel: bench.py against the in-tree library vs tools/bin/libbt709hip_rtz.so
#               (python tools/lab_variants.py tools/bin/libbt709hip_rtz.so BT709_INDEX_RTZ)
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2 3; do
  for v in rne rtz; do echo "== decode_lab $v (round $round)"; tools/bin/decode_lab_$v 0 5 | head -2; done
done
half() { python bench.py --workload 8k-half --no-cpu-baseline --steps 40 "$@" 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Gpx/s', d['roofline']['avg_launch_us'], 'us', d['roofline']['frac'], d['parity_spot_check'])"; }
for round in 1 2 3; do
  echo "== half rne (round $round)"; half
  echo "== half rtz (round $round)"; half --library tools/bin/libbt709hip_rtz.so
done
