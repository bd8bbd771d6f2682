#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c11; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_tkl_gpu.py tests/test_fuzz_gpu.py tests/test_rank_order_gpu.py tests/test_torch_ops_gpu.py -x -q -m gpu -k "tkl or capturable" 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -6 $O/t.log | cut -c1-300
for v in 0 1; do echo "== TKL no_wg=$v"; MM_TKL_NO_WG=$v timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120; MM_TKL_NO_WG=$v timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1 | cut -c1-120; MM_TKL_NO_WG=$v timeout 300 python tools/bench_tkl.py --steps 10 --docs 1024 2>&1 | tail -1 | cut -c1-120; done
echo "== kernel times TKL full"; timeout 400 bash tools/kernel_times.sh r3c11_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL ragged"; timeout 400 bash tools/kernel_times.sh r3c11_tklr python tools/bench_tkl.py --steps 5 2>&1 | grep "mm::"
echo "total $(( $(date +%s)-t0 ))s"
