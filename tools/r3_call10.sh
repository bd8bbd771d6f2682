#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c10; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_dot_topk_gpu.py tests/test_fullsize_gpu.py tests/test_maxsim_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -8 $O/t.log | cut -c1-300
echo "== dot"; timeout 400 python tools/bench_dot_topk.py --steps 3 2>&1 | tail -2 | cut -c1-300
echo "== kernel times dot"; timeout 400 bash tools/kernel_times.sh r3c10_dot python tools/bench_dot_topk.py --steps 2 2>&1 | grep "mm::"
echo "total $(( $(date +%s)-t0 ))s"
