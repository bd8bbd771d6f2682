#!/bin/bash
# Conv-KNRM 3 x 3 multi launch: the per-combination form (default) vs one wavefront per document tensor looping over the query
# tensors (MM_KP_MULTI_LOOP=1): golden tests, time per launch round-robin, FETCH_SIZE per launch.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_variants_gpu.py -x -q -m gpu 2>&1 | tail -3
MM_KP_MULTI_LOOP=1 timeout 600 python -m pytest tests/test_variants_gpu.py -x -q -m gpu -k conv 2>&1 | tail -2
for v in 1 0; do MM_KP_MULTI_LOOP=$v python tools/bench_conv_knrm_multi.py 10 2>/dev/null | tail -1 | sed "s/^/loop=$v: /"; done
for v in 1 0; do MM_KP_MULTI_LOOP=$v python tools/bench_conv_knrm_multi.py 10 2>/dev/null | tail -1 | sed "s/^/loop=$v: /"; done
for v in 1 0; do
  rm -rf gpurun_out/pmc_conv_$v
  MM_KP_MULTI_LOOP=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_conv_$v/pmc_fetch -o c -- python tools/bench_conv_knrm_multi.py 3 > /dev/null 2>&1
  MM_PROF_COMMAND="MM_KP_MULTI_LOOP=$v tools/bench_conv_knrm_multi.py" python tools/summarize_rocprof.py gpurun_out/pmc_conv_$v gpurun_out/pmc_conv_$v.json "kernel_pool" > /dev/null
  python - $v <<'P'
import json, sys
j = json.load(open(f"gpurun_out/pmc_conv_{sys.argv[1]}.json"))
for k, v in j["pmc"].items():
    f = v.get("FETCH_SIZE")
    if f: print(f"loop={sys.argv[1]}: {k[:70]} dispatches {f['dispatches']} FETCH_SIZE x 2 = {f['avg_per_dispatch'] * 2048 / 1e9:.2f} GB per launch, {f['avg_dispatch_ns'] / 1e6:.3f} ms")
P
  find gpurun_out/pmc_conv_$v -name "*.db" -delete
done
