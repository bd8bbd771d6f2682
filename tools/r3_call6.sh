#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c6; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_tkl_gpu.py tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -4 $O/t.log | cut -c1-300
echo "== TKL"; timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120; timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1 | cut -c1-120
echo "== TK"; timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 20 2>&1 | tail -1 | cut -c1-120
timeout 300 python tools/bench_kernel_pool.py --queries 64 --steps 20 --qlen config1 2>&1 | tail -1 | cut -c1-120
echo "== kernel times TKL full"; timeout 400 bash tools/kernel_times.sh r3c6_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL ragged"; timeout 400 bash tools/kernel_times.sh r3c6_tklr python tools/bench_tkl.py --steps 5 2>&1 | grep "mm::"
echo "total $(( $(date +%s)-t0 ))s"
