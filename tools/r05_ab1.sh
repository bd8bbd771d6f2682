#!/bin/bash
# round 5, GPU call 2: new tests + A/B of the TKL epilogue orderings and of the two-wavefront pooling form
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_fp16_flow_gpu.py tests/test_colbert_dropin_gpu.py tests/test_variants_gpu.py "tests/test_tkl_gpu.py::test_folded_region_epilogue_is_bit_equal_to_the_standalone_region_kernel" "tests/test_tkl_gpu.py::test_folded_region_epilogue_is_stable_under_repetition_concurrency_and_a_poisoned_workspace" -x -q -m gpu 2>&1 | tail -4
echo "--- pooling tests with MM_KP128_OCC=2 forced"
MM_KP128_OCC=2 timeout 600 python -m pytest tests/test_variants_gpu.py tests/test_kernel_pool_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "--- TKL epilogue orderings (ms, frac)"
LEG=tkl VARIANTS="default ep0 ep2" bash tools/ab_library_variants.sh
for r in 1 2; do MM_TKL_REGION_KERNEL=1 python bench.py --only tkl --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read())['result']; print('region_kernel tkl', round(r['ms'],4), round(r['roofline']['frac'],4))"; done
echo "--- pooling variants leg, MM_KP128_OCC = 1 / 2"
for r in 1 2; do for o in 1 2; do MM_KP128_OCC=$o python bench.py --only variants --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']
print('occ$o', {k:(round(v['ms'],4), round(v['roofline']['frac'],3)) for k,v in r.items() if isinstance(v,dict) and 'ms' in v})"; done; done
} > gpurun_out/r05_ab1.txt 2>&1
cat gpurun_out/r05_ab1.txt
