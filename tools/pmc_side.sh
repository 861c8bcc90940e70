C1="SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
C2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
echo "== encode 32"; bash tools/pmc_quick.sh e32 encode_bgra "$C1" "$C2" -- python3 tools/bench_encode.py --frames-per-launch 32 --steps 4
echo "== encode 1"; bash tools/pmc_quick.sh e1 encode_bgra "$C1" "$C2" -- python3 tools/bench_encode.py --frames-per-launch 1 --steps 4
echo "== rgba16f 128"; bash tools/pmc_quick.sh r16 decode_nv12_rgba16f "$C1" "$C2" -- python3 tools/bench_scaled.py --path rgba16f --ring 128 --frames-per-launch 128 --steps 3
echo "== render8 16"; bash tools/pmc_quick.sh r8 render_scaled "$C1" "$C2" -- python3 tools/bench_scaled.py --path render8 --frames-per-launch 16 --steps 3
echo "== scaled 8"; bash tools/pmc_quick.sh s8 decode_nv12_scaled "$C1" "$C2" -- python3 tools/bench_scaled.py --path scaled --frames-per-launch 8 --steps 3
echo "== 4k apple"; bash tools/pmc_quick.sh a4k decode_nv12_quads "$C1" "$C2" -- python3 bench.py --no-cpu-baseline --no-smooth-leg --placement-tries 1 --steps 3 --warmup 1
rm -rf gpurun_out/pmcq_*
