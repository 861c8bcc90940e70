#!/bin/bash
# rgba16f packed-pair form: where the time goes (stub builds, same process, same ring) + counters of the shipped kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_15_ab.txt
for gamma in 0 1; do
  echo "## gamma $gamma, 128 frames per launch" >> $O
  timeout 900 python tools/ab_libs.py --format rgba16f --ring 128 --per-launch 128 --rounds 3 --tries 3 --gamma $gamma shipped tools/bin/libbt709hip_head.so tools/bin/lab_f16p_noscale.so tools/bin/lab_f16p_nocand.so tools/bin/lab_f16p_not.so tools/bin/lab_f16p_nogather.so tools/bin/lab_f16p_cvtonly.so tools/bin/lab_f16p_noarith.so >> $O 2>&1
done
tools/pmc_quick.sh f16p rgba16f "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" -- python3 tools/ab_libs.py --format rgba16f --ring 128 --per-launch 128 --rounds 1 --steps 5 --tries 1 shipped > gpurun_out/r05_15_pmc.txt 2>&1
cat $O gpurun_out/r05_15_pmc.txt
