#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c12; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py tests/test_fuzz_gpu.py tests/test_torch_ops_gpu.py tests/test_tkl_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/t.log; echo "tests $(( $(date +%s)-t0 ))s"; tail -6 $O/t.log | cut -c1-300
echo "== eval_batch"; timeout 400 python bench.py --only eval_batch --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())['result']['shapes']
for k, v in j.items(): print(k, {x: round(v[x], 1) for x in ('us_per_call_device', 'us_per_call_completed', 'us_per_call_host_issue')}, round(v['roofline']['frac'], 3))"
echo "total $(( $(date +%s)-t0 ))s"
