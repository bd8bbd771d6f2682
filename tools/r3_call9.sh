#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c9; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_maxsim_gpu.py tests/test_kernel_pool_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -8 $O/t.log | cut -c1-300
for v in 0 1; do echo "== all-pairs nowg=$v"; for sz in "256 256" "1024 1024" "512 4096" "64 1024"; do MM_MAXSIM_INB_NOWG=$v timeout 300 python tools/bench_inbatch.py $sz 2>&1 | tail -1 | cut -c1-200; done; done
echo "total $(( $(date +%s)-t0 ))s"
