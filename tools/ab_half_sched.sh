cd "${GRAFT_REPO_ROOT:-.}"
half() { python bench.py --workload 8k-half --no-cpu-baseline --steps 40 "$@" 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Gpx/s', d['roofline']['avg_launch_us'], 'us', d['roofline']['frac'], d['parity_spot_check'])"; }
for round in 1 2; do
  echo "== base (batch 6, U=2)"; half
  echo "== batch 12"; half --library tools/bin/libbt709hip_b12.so
  echo "== U=3"; half --library tools/bin/libbt709hip_u3.so
  echo "== batch 12, U=1"; half --library tools/bin/libbt709hip_b12u1.so
  echo "== 2 WG/CU (uniform: 4 dec copies, 2 enc)"; half --decoder-option 3=512 --decoder-option 4=80
  echo "== 2 WG/CU split-encode lib (8 dec copies)"; half --library tools/bin/libbt709hip_split.so --decoder-option 3=512 --decoder-option 4=80
done
