#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c4; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 600 python -m pytest tests/test_tkl_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t_tkl.log; echo "tkl tests $(( $(date +%s)-t0 ))s"; tail -4 $O/t_tkl.log | cut -c1-300
timeout 600 python -m pytest tests/test_rank_order_gpu.py tests/test_torch_ops_gpu.py tests/test_fuzz_gpu.py -q -m gpu -k "tkl_split or capturable or tkl" 2>&1 | tail -3 | cut -c1-300
for d in 0 8 16 1; do echo "== TKL dbg=$d"; MM_KP_DBG=$d timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120; MM_KP_DBG=$d timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1 | cut -c1-120; done
echo "== TKL Q=30"; timeout 300 python tools/bench_tkl.py --steps 10 --Q 30 2>&1 | tail -1 | cut -c1-140
echo "== kernel times TKL full"; timeout 400 bash tools/kernel_times.sh r3c4_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL full wb stores"; MM_KP_DBG=8 timeout 400 bash tools/kernel_times.sh r3c4_tkl8 python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL ragged"; timeout 400 bash tools/kernel_times.sh r3c4_tklr python tools/bench_tkl.py --steps 5 2>&1 | grep "mm::"
echo "total $(( $(date +%s)-t0 ))s"
