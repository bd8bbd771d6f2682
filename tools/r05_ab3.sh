#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "--- fp32 MaxSim leg (kernel_pool_split128_kernel<MX>), one / two wavefronts per SIMD"
for r in 1 2; do for o in 1 2; do MM_KP128_OCC=$o python bench.py --only maxsim_fp32 --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read())['result']; print('occ$o maxsim_fp32', round(r['ms'],4), round(r['roofline']['frac'],4))"; done; done
MM_KP128_OCC=2 timeout 600 python -m pytest tests/test_maxsim_gpu.py -x -q -m gpu -k "fp32 or f32 or float32" 2>&1 | tail -2
} > gpurun_out/r05_ab3.txt 2>&1
cat gpurun_out/r05_ab3.txt
