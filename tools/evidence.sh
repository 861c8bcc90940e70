#!/bin/bash
# One-call evidence run on the GPU box: tests, every bench line, rocprof + PMC summaries.
# usage (through gpurun, from the repo root): tools/evidence.sh r06
R=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$R; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_4k_driver_args.json 2> $O/bench_4k.err
python bench.py --no-cpu-baseline > $O/bench_4k.json 2>/dev/null
python bench.py --workload 1080p > $O/bench_1080p.json 2>/dev/null
# round 6: the reference's own on-disk format as the bench's input -- a 64-frame 1080p clip written through the GPU encoder, then bench.py --y4m
python tools/make_y4m_clip.py /tmp/qt_pattern_1080p.y4m --frames 64 > $O/make_y4m_clip.txt 2>&1
python bench.py --y4m /tmp/qt_pattern_1080p.y4m > $O/bench_y4m.json 2> $O/bench_y4m.err
python bench.py --workload 8k-half > $O/bench_8k-half.json 2>/dev/null
for sh in 8 4 2 1; do python bench.py --workload 4k-batch8 --share $sh --no-cpu-baseline --steps 400 > $O/bench_batch8_share$sh.json 2>/dev/null; done
python bench.py --workload 4k-batch8 --share 1 --streams 1 --no-cpu-baseline --steps 400 > $O/bench_batch8_share1_1stream.json 2>/dev/null
python bench.py --workload 4k-batch8 --share 1 --streams 2 --no-cpu-baseline --steps 400 > $O/bench_batch8_share1_2streams.json 2>/dev/null
python bench.py --workload 4k-batch8 --share 1 --graph --no-cpu-baseline --steps 400 > $O/bench_batch8_share1_graph.json 2>/dev/null
# round 4: the reference's cadence (one frame per call, one stream) with the coalescing submit; the plain multi-rank command on one GPU
for c in 8 32; do python bench.py --workload 4k-batch8 --share 1 --streams 1 --coalesce $c --no-cpu-baseline --steps 400 > $O/bench_batch8_share1_1stream_coalesce$c.json 2>/dev/null; done
# round 5: ranks that outnumber the visible GPUs are REFUSED (rc 2, no line) ...
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_4k_plain_command_2ranks_refused.json 2> $O/bench_4k_plain_command_2ranks_refused.txt; echo "rc=$?" >> $O/bench_4k_plain_command_2ranks_refused.txt
# ... unless asked for: the functional N > 1 run on this one-GPU box (shared_devices: true, n_gpus 1, every rank spot-checked)
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --allow-shared-devices > $O/bench_4k_plain_command_2ranks_one_gpu.json 2> $O/bench_4k_plain_command_2ranks_one_gpu.err
python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline --allow-shared-devices > $O/bench_4k_plain_command_4ranks_one_gpu.json 2> $O/bench_4k_plain_command_4ranks_one_gpu.err
# ONE process driving N GPUs (bt709hip_ringset_*): two and four lanes wrapped onto this GPU
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --allow-shared-devices --launcher threads > $O/bench_4k_threads_2lanes_one_gpu.json 2> $O/bench_4k_threads_2lanes_one_gpu.err
python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline --allow-shared-devices --launcher threads --ring 64 > $O/bench_4k_threads_4lanes_one_gpu.json 2> $O/bench_4k_threads_4lanes_one_gpu.err
# and the driver's launcher line, 2 ranks wrapped onto this one GPU
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --allow-shared-devices > $O/bench_4k_torchrun_2ranks_one_gpu.json 2> $O/bench_4k_torchrun_2ranks_one_gpu.err
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --workload 8k-half --content flat --no-cpu-baseline > $O/bench_8k-half_flat.json 2>/dev/null
{ tools/ab_batch8.sh; echo "# two frames per step"; SHARE=2 STREAMS="1 2 3 4" tools/ab_batch8.sh; } > $O/batch8_streams.txt 2>&1
tools/encode_single_shapes.sh "320 512" "1 3" > $O/encode_single_shapes.txt 2>&1
tools/ab_bands.sh > $O/ab_bands.txt 2>&1
{ python tools/bench_encode.py --ring 256 --frames-per-launch 256 --steps 10 --placement-tries 5; python tools/bench_encode.py --frames-per-launch 32; } > $O/bench_encode.json 2>/dev/null
python tools/bench_encode.py --frames-per-launch 1 > $O/bench_encode_single.json 2>/dev/null
tools/bench_paths.sh > $O/bench_paths.txt 2>&1
python tools/stream_bench.py > $O/stream_bench.txt 2>&1
{ python tools/bench_alpha11.py - 256 4; PER_LAUNCH=32 python tools/bench_alpha11.py - 256 4; python tools/bench_half_alpha.py 7680 4320 8 1; python tools/bench_half_alpha.py 7680 4320 8 0; python tools/bench_half_alpha.py 3840 2160 16 1; } > $O/bench_alpha.txt 2>&1
tools/profile_gpu.sh 4k > /dev/null 2>&1
tools/profile_gpu.sh 1080p --workload 1080p > /dev/null 2>&1
# the JSON line of the PROFILED process itself: the sentinel-cut kernel stats are compared with that line's avg_launch_us
for t in 4k 1080p; do grep '^{' gpurun_out/prof_$t/trace.log | tail -1 > $O/profiled_run_$t.json; done
# the sRGB-mode (arithmetic quantiser) and alpha variants of the 1:1 kernel: one kernel trace of tools/bench_alpha11.py holds all five decoders
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_alpha11/trace" -o trace -- python3 "$OLDPWD/tools/bench_alpha11.py" > "$OLDPWD/gpurun_out/prof_alpha11_trace.log" 2>&1 )
tools/profile_gpu.sh 8k-half --workload 8k-half > /dev/null 2>&1
grep '^{' gpurun_out/prof_8k-half/trace.log | tail -1 > $O/profiled_run_8k-half.json
PROFILE_PROG=tools/bench_encode.py tools/profile_gpu.sh encode --frames-per-launch 32 > /dev/null 2>&1
PROFILE_PROG=tools/bench_scaled.py tools/profile_gpu.sh scaled --path scaled --frames-per-launch 8 > /dev/null 2>&1
PROFILE_PROG=tools/bench_scaled.py tools/profile_gpu.sh rgba16f --path rgba16f --ring 128 --frames-per-launch 128 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/prof_4k $R 4k > /dev/null
python tools/pmc_summary.py gpurun_out/prof_1080p $R 1080p > /dev/null
cp gpurun_out/prof_alpha11/trace/trace_kernel_stats.csv profiles/${R}_alpha11_kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py gpurun_out/prof_8k-half $R 8k-half > /dev/null
python tools/pmc_summary.py gpurun_out/prof_encode $R encode encode_bgra > /dev/null
python tools/pmc_summary.py gpurun_out/prof_scaled $R scaled decode_nv12_scaled > /dev/null
python tools/pmc_summary.py gpurun_out/prof_rgba16f $R rgba16f decode_nv12_rgba16f > /dev/null
mkdir -p $O/profiles; cp profiles/${R}_*_kernel_stats*.csv profiles/${R}_*_pmc.json profiles/pmc_traffic.json $O/profiles/
# gpurun merges at most 64 MiB back: the raw traces and counter dumps have been condensed above, drop them
rm -rf gpurun_out/prof_* gpurun_out/pmcq_*
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f.split("/")[-1], d.get("value", d.get("gpixel_per_s")), r.get("frac", d.get("frac_of_8TBps")), "first", r.get("first_allocation_frac"), r.get("avg_launch_us"), r.get("same_run_copy_GBps"), d.get("parity_spot_check"))
    except Exception as e: print(f, "ERR", e)
PY
python - <<PY
import json,csv
for t in ("4k","1080p","8k-half"):
    try:
        d=json.loads(open("$O/profiled_run_%s.json"%t).read()); r=d["roofline"]
        rows=list(csv.DictReader(open("profiles/${R}_%s_kernel_stats.csv"%t)))
        print(t, "profiled process: avg_launch_us", r["avg_launch_us"], "ms_per_step", d["ms_per_step"], "| rocprof, timed regions only:", [(x["Calls"], x["AverageNs"], x["MedianNs"]) for x in rows])
    except Exception as e: print(t, "ERR", e)
PY
cat $O/bench_paths.txt; tail -5 $O/stream_bench.txt
