#!/bin/bash
# Lab: what the LDS gathers of encode_bgra_nv12 cost -- the same launches over random bytes (every
# gather conflicts), a smooth gradient and one colour per picture (gathers broadcast).
#   gpurun --timeout 900 -- 'bash tools/ab_encode_content.sh > gpurun_out/ab_encode_content.txt 2>&1'
for rep in 1 2; do
for fpl in 256 32; do
for c in random smooth flat; do
  python tools/bench_encode.py --ring 256 --frames-per-launch $fpl --steps 12 --placement-tries 4 --content $c
done; done; done
