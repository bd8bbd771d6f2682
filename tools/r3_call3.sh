#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c3; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 600 python -m pytest tests/test_tkl_gpu.py tests/test_kernel_pool_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t_tkl.log; echo "tkl+kp tests $(( $(date +%s)-t0 ))s"; tail -4 $O/t_tkl.log | cut -c1-300
timeout 600 python -m pytest tests/test_rank_order_gpu.py tests/test_torch_ops_gpu.py -q -m gpu -k "tkl_split or capturable or tkl" -s 2>&1 | grep -E "rank parity|passed|failed|Error|assert" | cut -c1-500 > $O/t_rank.log; echo "rank $(( $(date +%s)-t0 ))s"; tail -6 $O/t_rank.log
echo "== TKL"; timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1 | cut -c1-140; timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-140
echo "== TKL Q=30"; timeout 300 python tools/bench_tkl.py --steps 10 --Q 30 2>&1 | tail -1 | cut -c1-140
for d in 1 2 3; do echo "== TKL dbg=$d"; MM_KP_DBG=$d timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120; done
echo "== kernel times TKL full"; timeout 400 bash tools/kernel_times.sh r3c3_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL ragged"; timeout 400 bash tools/kernel_times.sh r3c3_tklr python tools/bench_tkl.py --steps 5 2>&1 | grep "mm::"
echo "== PMC TK old kernel"; MM_KP_NO_WG=1 timeout 400 bash tools/pmc_pass.sh r3c3_tk "SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python tools/bench_kernel_pool.py --full --queries 64 --steps 5 2>&1 | tail -8
echo "== TK clock over time (20 steps then 100 steps)"; MM_KP_NO_WG=1 timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 3 2>&1 | tail -1 | cut -c1-100; MM_KP_NO_WG=1 timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 100 2>&1 | tail -1 | cut -c1-100
echo "total $(( $(date +%s)-t0 ))s"
