#!/bin/bash
# Kernel times + three counter passes of ONE command (run on the GPU box from the repo root):
#   bash tools/prof_cmd.sh <tag> <command...>        e.g.  bash tools/prof_cmd.sh kpbwd python tools/bench_kp_bwd.py --child --pairs 2048
# -> gpurun_out/kt_<tag>/summary.json (rocprofv3 --kernel-trace --stats) and gpurun_out/pmc_<tag>_{a,b,c}/summary.json
# (counters only with --kernel-trace, one pass per counter group: tools/pmc_pass.sh); text summaries on stdout.
TAG=$1; shift
bash tools/kernel_times.sh $TAG "$@"
bash tools/pmc_pass.sh ${TAG}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "$@"
bash tools/pmc_pass.sh ${TAG}_b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "$@"
bash tools/pmc_pass.sh ${TAG}_c "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_WAVES_LT_64 SQ_ACTIVE_INST_FLAT" "$@"
