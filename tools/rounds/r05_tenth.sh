#!/bin/bash
# Round 5: the encoder as straight-line tiles (quads per lane, row pairs per workgroup, lanes per tile) against the walking kernel:
# parity first, then tools/bench_encode.py, fresh process per line (placement: 4 candidates per slab), two rounds.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_encoder.py tests/test_y4m.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05/pytest_encoder.log
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("%-34s %7.3f us/frame %7.1f Gpx/s  frac %.4f  %s" % (sys.argv[1], d["us_per_frame"], d["gpixel_per_s"], d["frac_of_8TBps"], d["kernel"]))'
{
for fpl in 256 32 1; do
  echo "## $fpl picture(s) per launch"
  for rep in 1 2; do
    for lib in shipped enc_walk enc_1_2_320 enc_2_1_512 enc_2_3_512 enc_1_4_512 enc_4_1_256 enc_4_2_256 enc_1_1_512; do
      L=""; [ $lib != shipped ] && L="--library tools/bin/lab_$lib.so"
      python tools/bench_encode.py --ring 256 --frames-per-launch $fpl --steps 12 --placement-tries 4 $L 2>/dev/null | python -c "$P" $lib
    done
  done
done
} > gpurun_out/r05/ab_encode_shapes.txt 2>&1
cat gpurun_out/r05/pytest_encoder.log gpurun_out/r05/ab_encode_shapes.txt
