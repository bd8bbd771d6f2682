#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
{
timeout 900 python -m pytest tests/test_colbert_dropin_gpu.py tests/test_fp16_flow_gpu.py tests/test_maxsim_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -2
for e in 0 1; do MM_MAXSIM_PY_AUTOGRAD=$e python bench.py --only eval_batch --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']['shapes']
print('python_path=$e', {k:(round(v['us_per_call_completed'],2), round(v['us_per_call_host_issue'],2), round(v['roofline']['frac'],3)) for k,v in r.items()})"; done
} > gpurun_out/r05_ab9.txt 2>&1
cat gpurun_out/r05_ab9.txt
