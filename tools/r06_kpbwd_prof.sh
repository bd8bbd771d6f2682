#!/bin/bash
# kernel times + counters of the TK backward at 2,048 pairs (run on the GPU box from the repo root)
CMD="python tools/bench_kp_bwd.py --child --pairs ${1:-2048}"
bash tools/kernel_times.sh kpbwd $CMD
bash tools/pmc_pass.sh kpbwd_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" $CMD
bash tools/pmc_pass.sh kpbwd_b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" $CMD
bash tools/pmc_pass.sh kpbwd_c "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_WAVES_LT_64 SQ_ACTIVE_INST_FLAT" $CMD
