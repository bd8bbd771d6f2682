#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py -x -q -m gpu 2>&1 | tail -3
MM_KP_MULTI_WG=0 timeout 600 python -m pytest tests/test_kernel_pool_gpu.py -x -q -m gpu -k "multi" 2>&1 | tail -2
echo "--- Conv-KNRM 3x3: wavefront-per-query-tensor workgroups (default) | flat independent workgroups | 2-D grid, one wavefront per SIMD"
v() { python bench.py --only variants --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']
print('$1', {k:(round(v['ms'],4), round(v['roofline']['frac'],3)) for k,v in r.items() if isinstance(v,dict) and 'ms' in v and ('conv' in k)})"; }
for r in 1 2; do
v default_wg
MM_KP_MULTI_WG=0 v flat
MM_KP_MULTI_2D=1 MM_KP128_OCC=1 v grid2d_occ1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_conv -o c -- python bench.py --only variants --lean --no-cpu-baseline > /dev/null 2>&1
python tools/summarize_rocprof.py gpurun_out/pmc_conv gpurun_out/pmc_conv_summary.json "mm::" > /dev/null
python - <<'P'
import json
j=json.load(open('gpurun_out/pmc_conv_summary.json'))
for k,v in j['pmc'].items():
    if 'split128' in k: print(k[:100], {c:(x['dispatches'], round(x['avg_per_dispatch'])) for c,x in v.items() if isinstance(x,dict) and 'avg_per_dispatch' in x})
P
find gpurun_out/pmc_conv -name "*.db" -delete
} > gpurun_out/r05_ab4.txt 2>&1
cat gpurun_out/r05_ab4.txt
