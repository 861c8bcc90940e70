#!/bin/bash
# Round 4: the RGBA16Float kernel's candidate from the tangent table (shipped) against v_log_f32 / v_exp_f32 (tools/bin/libbt709hip_logexp.so =
# the library built from the commit before), tools/bench_scaled.py --path rgba16f, 4K, ring 128 / 64, alternating fresh processes.
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests -m gpu -x -q -k "rgba16f or two_pass or half_lookup or render" 2>&1 | tail -3
run() { python tools/bench_scaled.py --path rgba16f "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-90s %7.2f us/frame %6.1f Gpx/s  frac %.4f' % (' '.join(sys.argv[1:]), d['us_per_frame'], d['out_gpixel_per_s'], d['frac_of_8TBps']))" "$@"; }
for round in 1 2; do
  for lib in "" "--library tools/bin/libbt709hip_logexp.so"; do
    run --ring 128 --frames-per-launch 128 $lib
    run --frames-per-launch 16 $lib
    run --frames-per-launch 1 $lib
    run --ring 128 --frames-per-launch 128 --gamma srgb $lib; run --ring 128 --frames-per-launch 128 --gamma itu709 $lib
  done
done
