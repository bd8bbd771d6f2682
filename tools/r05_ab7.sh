#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
{
LEG=tkl VARIANTS="default t3n3 t3n4 t4n2" BENCH_ARGS="" bash tools/ab_library_variants.sh
for v in default t3n4; do
  lib=$PWD/variants/libmm_native_$v.so; [ "$v" = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
  MM_NATIVE_LIB=$lib python bench.py --only tkl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read())['result']; print('$v tkl', round(r['ms'],4), round(r['roofline']['frac'],4), 'b1024', r.get('batch_1024_documents',{}).get('ms'), r.get('batch_1024_documents',{}).get('frac'))"
done
MM_NATIVE_LIB=$PWD/variants/libmm_native_t3n4.so timeout 300 python -m pytest tests/test_tkl_gpu.py -x -q -m gpu 2>&1 | tail -1
} > gpurun_out/r05_ab7.txt 2>&1
cat gpurun_out/r05_ab7.txt
