#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c5; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_tkl_gpu.py tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py tests/test_maxsim_gpu.py tests/test_colbert_dropin_gpu.py tests/test_dot_topk_gpu.py tests/test_torch_ops_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -4 $O/t.log | cut -c1-300
echo "== TKL"; timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1 | cut -c1-120; timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1 | cut -c1-120
echo "== TKL Q=30"; timeout 300 python tools/bench_tkl.py --steps 10 --Q 30 2>&1 | tail -1 | cut -c1-140
echo "== TK"; timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 20 2>&1 | tail -1 | cut -c1-120
echo "== kernel times TKL full"; timeout 400 bash tools/kernel_times.sh r3c5_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | grep "mm::"
echo "== kernel times TKL ragged"; timeout 400 bash tools/kernel_times.sh r3c5_tklr python tools/bench_tkl.py --steps 5 2>&1 | grep "mm::"
echo "== eval_batch"; timeout 400 python bench.py --only eval_batch --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())['result']['shapes']
for k, v in j.items(): print(k, {x: round(v[x], 1) for x in ('us_per_call_device', 'us_per_call_completed', 'us_per_call_host_issue')}, round(v['roofline']['frac'], 3))"
echo "total $(( $(date +%s)-t0 ))s"
