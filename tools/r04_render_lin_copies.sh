#!/bin/bash
# Round 4: pass 2 alone from a BGRA8 intermediate with lin[256] in 1 (shipped) / 8 / 16 / 32 interleaved LDS copies
# (-DBT709_RENDER_LIN_COPIES_LOG2=n builds), 4K -> 1440p and 1080p -> 4K, 1 and 16 surfaces per launch, alternating processes.
cd "${GRAFT_REPO_ROOT:-.}"
run() { python tools/bench_scaled.py --path render8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-100s %7.2f us/frame %6.1f Gpx/s out  frac %.4f' % (' '.join(sys.argv[1:]), d['us_per_frame'], d['out_gpixel_per_s'], d['frac_of_8TBps']))" "$@"; }
for round in 1 2; do
  for lib in "" "--library tools/bin/libbt709hip_lin3.so" "--library tools/bin/libbt709hip_lin4.so" "--library tools/bin/libbt709hip_lin5.so"; do
    run --frames-per-launch 16 $lib
    run --frames-per-launch 1 $lib
    run --width 1920 --height 1080 --out-width 3840 --out-height 2160 --frames-per-launch 8 $lib
  done
done
