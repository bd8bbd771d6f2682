#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c2; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 600 python -m pytest tests/test_kernel_pool_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t_kp.log; echo "kp tests $(( $(date +%s)-t0 ))s"; tail -25 $O/t_kp.log | cut -c1-300
timeout 600 python -m pytest tests/test_rank_order_gpu.py -q -m gpu -k "tk_split" -s 2>&1 | grep -E "rank parity|passed|failed|Error|assert" | cut -c1-700 > $O/t_rank.log; echo "rank $(( $(date +%s)-t0 ))s"; tail -6 $O/t_rank.log
for v in 0; do
  echo "== TK no_wg=$v"
  MM_KP_NO_WG=$v timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 10 2>&1 | tail -1
  MM_KP_NO_WG=$v timeout 300 python tools/bench_kernel_pool.py --queries 64 --steps 10 --qlen config1 2>&1 | tail -1
  MM_KP_NO_WG=$v timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 10 --qlen 8 2>&1 | tail -1
done
for d in 0 1 2 3 4 7; do
  echo "== TKL stage-1 removal dbg=$d"
  MM_KP_DBG=$d timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120
done
echo "== kernel times TK WG"; timeout 400 bash tools/kernel_times.sh r3c2_tk python tools/bench_kernel_pool.py --full --queries 64 --steps 5 2>&1 | grep "mm::"
echo "== PMC TK WG"; timeout 400 bash tools/pmc_pass.sh r3c2_tk "SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python tools/bench_kernel_pool.py --full --queries 64 --steps 5 2>&1 | tail -12
echo "total $(( $(date +%s)-t0 ))s"
