cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
tools/bench_paths.sh > $O/bench_paths.txt 2>&1
PROFILE_PROG=tools/bench_scaled.py tools/profile_gpu.sh rgba16f --path rgba16f --frames-per-launch 16 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/prof_rgba16f r04 rgba16f decode_nv12_rgba16f > /dev/null
mkdir -p $O/profiles; cp profiles/r04_rgba16f* $O/profiles/
rm -rf gpurun_out/prof_*
grep rgba16f $O/bench_paths.txt | cut -c1-200
python - <<'PY'
import json
d=json.load(open('profiles/r04_rgba16f_pmc.json')); m=d['per_launch_mean']; dur=d['duration_ns']['mean']; cyc=dur*2.4
print('dur', dur, 'VALU busy', m['SQ_ACTIVE_INST_VALU']/1024/cyc*4, 'LDS busy', m['SQ_LDS_IDX_ACTIVE']/256/cyc, 'valu inst per px-wave', m['SQ_INSTS_VALU']/(3840*2160*16/64))
PY
