#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
{
timeout 600 python -m pytest tests/test_colbert_dropin_gpu.py tests/test_maxsim_ragged_bwd_gpu.py tests/test_torch_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "--- C++ node"; python tools/host_step_profile.py 3000 2>&1 | grep "^=="
echo "--- Python node (MM_MAXSIM_PY_AUTOGRAD=1)"; MM_MAXSIM_PY_AUTOGRAD=1 python tools/host_step_profile.py 3000 2>&1 | grep "^=="
for e in 0 1; do MM_MAXSIM_PY_AUTOGRAD=$e python bench.py --only train_step --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']['colbert_fp16_autocast_q32_d180_e128']
print('py_node=$e', {k:(round(v['step_us'],1), round(v['kernel_us'],1), round(v['host_us'],1)) for k,v in r.items()})"; done
} > gpurun_out/r05_ab8.txt 2>&1
cat gpurun_out/r05_ab8.txt
