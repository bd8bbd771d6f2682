#!/bin/bash
# what limits TKL's stage 1 / the TK kernel: wavefronts per CU of the E = 100n streaming launch (4 = one per SIMD, the default)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
LEG=tkl VARIANTS="default wpc3 wpc2" bash tools/ab_library_variants.sh
LEG=tk VARIANTS="default wpc3 wpc2" bash tools/ab_library_variants.sh
for v in default wpc2; do
  lib=$PWD/variants/libmm_native_$v.so; [ "$v" = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
  rm -rf gpurun_out/tr_$v; MM_NATIVE_LIB=$lib rocprofv3 --kernel-trace --stats -d gpurun_out/tr_$v/trace -o t -- python bench.py --only tkl --lean --no-cpu-baseline > /dev/null 2>&1
  python tools/summarize_rocprof.py gpurun_out/tr_$v gpurun_out/tr_$v.json "mm::" > /dev/null
  python -c "
import json; j=json.load(open('gpurun_out/tr_$v.json'))
print('$v', [(k['name'][10:40], k['calls'], round(k.get('median_us') or k['avg_us'],1)) for k in j['kernel_trace'][:4]])"
  find gpurun_out/tr_$v -name '*.db' -delete
done
} > gpurun_out/r05_ab5.txt 2>&1
cat gpurun_out/r05_ab5.txt
