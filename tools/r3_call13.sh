#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c13; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_tkl_gpu.py tests/test_fuzz_gpu.py tests/test_rank_order_gpu.py tests/test_torch_ops_gpu.py tests/test_variants_gpu.py tests/test_dot_topk_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "tkl or capturable or idcm or dot" 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -4 $O/t.log | cut -c1-300
echo "== TKL"; timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120; timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1 | cut -c1-120; timeout 300 python tools/bench_tkl.py --steps 10 --Q 30 2>&1 | tail -1 | cut -c1-120
echo "== kernel times TKL full"; timeout 400 bash tools/kernel_times.sh r3c13_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL ragged"; timeout 400 bash tools/kernel_times.sh r3c13_tklr python tools/bench_tkl.py --steps 5 2>&1 | grep "mm::"
echo "total $(( $(date +%s)-t0 ))s"
echo "== dot"; timeout 400 python tools/bench_dot_topk.py --steps 3 2>&1 | tail -1 | cut -c1-200
echo "== kernel times dot"; timeout 400 bash tools/kernel_times.sh r3c13_dot python tools/bench_dot_topk.py --steps 2 2>&1 | grep "mm::"
