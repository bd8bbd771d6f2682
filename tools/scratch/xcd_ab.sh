#!/bin/bash
# Conv-KNRM loop kernel: XCD-grouped wavefront ids (default) vs consecutive ids (MM_KP_MULTI_2D=1): ms per launch and FETCH_SIZE per launch
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for v in 0 1; do MM_KP_MULTI_2D=$v python tools/bench_conv_knrm_multi.py 10 2>/dev/null | tail -1 | sed "s/^/consecutive_ids=$v: /"; done
done
for v in 0 1; do
  rm -rf gpurun_out/pmc_xcd_$v
  MM_KP_MULTI_2D=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_xcd_$v/pmc_fetch -o c -- python tools/bench_conv_knrm_multi.py 3 > /dev/null 2>&1
  MM_PROF_COMMAND="MM_KP_MULTI_2D=$v tools/bench_conv_knrm_multi.py" python tools/summarize_rocprof.py gpurun_out/pmc_xcd_$v gpurun_out/pmc_xcd_$v.json "kernel_pool" > /dev/null
  python - $v <<'P'
import json, sys
j = json.load(open(f"gpurun_out/pmc_xcd_{sys.argv[1]}.json"))
for k, v in j["pmc"].items():
    f = v.get("FETCH_SIZE")
    if f and "multi128" in k: print(f"consecutive_ids={sys.argv[1]}: FETCH_SIZE x 2 = {f['avg_per_dispatch'] * 2048 / 1e9:.2f} GB per launch, {f['avg_dispatch_ns'] / 1e6:.3f} ms")
P
  find gpurun_out/pmc_xcd_$v -name "*.db" -delete
done
timeout 600 python -m pytest tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py -x -q -m gpu -k "multi or conv" 2>&1 | tail -2
