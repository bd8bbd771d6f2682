#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
{
run() {  # run <tag> <env...>
  local tag=$1; shift
  rm -rf gpurun_out/pmc_conv_$tag
  env "$@" python tools/bench_conv_knrm_multi.py 5 2>/dev/null | tail -1 | sed "s/^/$tag: /"
  env "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_conv_$tag/pmc_fetch -o c -- python tools/bench_conv_knrm_multi.py 3 > /dev/null 2>&1
  MM_PROF_COMMAND="tools/bench_conv_knrm_multi.py $*" python tools/summarize_rocprof.py gpurun_out/pmc_conv_$tag gpurun_out/pmc_conv_$tag.json "kernel_pool_split128" > /dev/null
  python - "$tag" <<'P'
import json, sys
j = json.load(open(f"gpurun_out/pmc_conv_{sys.argv[1]}.json"))
for k, v in j["pmc"].items():
    f = v.get("FETCH_SIZE")
    if f: print(f"{sys.argv[1]}: {k[:70]}  dispatches {f['dispatches']}  FETCH_SIZE x 2 (gfx950 half-count) = {f['avg_per_dispatch'] * 2048 / 1e9:.2f} GB per launch, {f['avg_dispatch_ns'] / 1e6:.3f} ms")
P
  find gpurun_out/pmc_conv_$tag -name "*.db" -delete
}
run grid2d_one_wavefront MM_KP_MULTI_2D=1 MM_KP128_OCC=1
run flat_two_wavefronts X=1
run grid2d_two_wavefronts MM_KP_MULTI_2D=1
run workgroup_per_range MM_KP_MULTI_WG=1
} > gpurun_out/r05_ab11.txt 2>&1
cat gpurun_out/r05_ab11.txt
