#!/bin/bash
# Conv-KNRM loop kernel: query-tile prefetch distance (none / head of the pair / last block of the pair): ms per launch and FETCH_SIZE per launch
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for v in default touch0 touch2; do
    lib=$PWD/variants/libmm_native_$v.so; [ $v = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
    MM_NATIVE_LIB=$lib python tools/bench_conv_knrm_multi.py 10 2>/dev/null | tail -1 | sed "s/^/$v: /"
  done
done
for v in default touch0 touch2; do
  lib=$PWD/variants/libmm_native_$v.so; [ $v = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
  rm -rf gpurun_out/pmc_touch_$v
  MM_NATIVE_LIB=$lib rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_touch_$v/pmc_fetch -o c -- python tools/bench_conv_knrm_multi.py 3 > /dev/null 2>&1
  MM_PROF_COMMAND="tools/bench_conv_knrm_multi.py ($v)" python tools/summarize_rocprof.py gpurun_out/pmc_touch_$v gpurun_out/pmc_touch_$v.json "kernel_pool" > /dev/null
  python - $v <<'P'
import json, sys
j = json.load(open(f"gpurun_out/pmc_touch_{sys.argv[1]}.json"))
for k, v in j["pmc"].items():
    f = v.get("FETCH_SIZE")
    if f and "multi128" in k: print(f"{sys.argv[1]}: FETCH_SIZE x 2 = {f['avg_per_dispatch'] * 2048 / 1e9:.2f} GB per launch, {f['avg_dispatch_ns'] / 1e6:.3f} ms")
P
  find gpurun_out/pmc_touch_$v -name "*.db" -delete
done
