#!/bin/bash
# round 5, GPU call 3: flat XCD-grouped multi launch + OCC=2, standalone region kernel as the default
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py tests/test_tkl_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "--- folded epilogue forced: the poisoned-workspace test"
MM_TKL_FOLD_REGIONS=1 timeout 300 python -m pytest tests/test_tkl_gpu.py -x -q -m gpu -k "poisoned or config3" 2>&1 | tail -2
echo "--- variants leg: default | 2-D grid | flat with one wavefront per SIMD | 2-D with two"
v() { python bench.py --only variants --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']
print('$1', {k:(round(v['ms'],4), round(v['roofline']['frac'],3)) for k,v in r.items() if isinstance(v,dict) and 'ms' in v and ('conv' in k or 'small' in k)})"; }
for r in 1 2; do
v default
MM_KP_MULTI_2D=1 MM_KP128_OCC=1 v grid2d_occ1
MM_KP128_OCC=1 v flat_occ1
MM_KP_MULTI_2D=1 v grid2d_occ2
done
echo "--- tkl leg, default (standalone region kernel)"
for r in 1 2; do python bench.py --only tkl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read())['result']; print('tkl', round(r['ms'],4), round(r['roofline']['frac'],4), 'b1024', r.get('batch_1024_documents',{}).get('ms'), r.get('batch_1024_documents',{}).get('frac'))"; done
} > gpurun_out/r05_ab2.txt 2>&1
cat gpurun_out/r05_ab2.txt
