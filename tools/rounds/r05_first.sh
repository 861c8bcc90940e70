#!/bin/bash
# Round 5, first gpurun call: the suite on the fresh box, then the traffic-pattern ceilings VERDICT r4 (Missing 4) asks for:
# RGBA16F kernel (arithmetic deleted / conversion only / no settlement / row-pairs-per-workgroup), encoder and +unconvert: with their
# arithmetic deleted.  Variant libraries: tools/lab_variants.py (built in the container, tools/bin/ travels with the snapshot).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05/pytest_first.log
{
echo "# RGBA16F target, 4K, ring 128, one process, one ring, alternating regions (tools/ab_libs.py --format rgba16f); fraction = 9.5 B per pixel / 8 TB/s"
for per in 128 16; do
  echo "## $per frames per launch"
  python tools/ab_libs.py --format rgba16f --ring 128 --per-launch $per --rounds 3 --steps 10 --tries 4 shipped tools/bin/lab_f16_noarith.so \
    tools/bin/lab_f16_noarith_notable.so tools/bin/lab_f16_noarith_rpb2.so tools/bin/lab_f16_cvtonly.so tools/bin/lab_f16_cvtonly_notable.so tools/bin/lab_f16_nosettle.so \
    tools/bin/lab_f16_rpb4.so tools/bin/lab_f16_rpb64.so 2>&1 | grep -v "^input slab\|^output slab"
done
} > gpurun_out/r05/ab_rgba16f_ceiling.txt 2>&1
{
echo "# encoder, 4K x 256 per launch, ring 256, fresh process per line (tools/bench_encode.py), shipped against the build with its lookups and arithmetic deleted"
for rep in 1 2; do
  for lib in "" "--library tools/bin/lab_enc_noarith.so"; do
    python tools/bench_encode.py --ring 256 --frames-per-launch 256 --steps 12 --placement-tries 4 $lib 2>&1 | tail -1
  done
done
for lib in "" "--library tools/bin/lab_enc_noarith.so"; do
  python tools/bench_encode.py --ring 64 --frames-per-launch 1 --steps 50 $lib 2>&1 | tail -1
done
} > gpurun_out/r05/ab_encode_ceiling.txt 2>&1
{
echo "# +unconvert: 4K, ring 64, one frame per call (tools/bench_unconvert.py), shipped against the build with its lookups deleted"
for rep in 1 2; do
  python tools/bench_unconvert.py 64 400
  python tools/bench_unconvert.py 64 400 tools/bin/lab_unc_noarith.so
done
} > gpurun_out/r05/ab_unconvert_ceiling.txt 2>&1
tail -n 40 gpurun_out/r05/*.txt gpurun_out/r05/pytest_first.log
