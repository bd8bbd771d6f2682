#!/bin/bash
# A/B of library builds on ONE box: LEG=<bench leg> VARIANTS="a b c" -> variants/libmm_native_<variant>.so (tools/build_variant.sh)
# selected in turn with MM_NATIVE_LIB, two round-robin passes; "default" = the shipped matchmaker_amd/csrc/libmm_native.so
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
for rnd in 1 2; do
for v in ${VARIANTS:-default}; do
  lib=$PWD/variants/libmm_native_$v.so
  [ "$v" = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
  MM_NATIVE_LIB=$lib python bench.py --only ${LEG:-tk} --lean --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab/${LEG:-tk}_$v.log 2>&1
  python - <<P
import json
for ln in reversed(open("gpurun_out/ab/${LEG:-tk}_$v.log").read().splitlines()):
    if ln.startswith("{"):
        j = json.loads(ln); r = j.get("result", j)
        print("$v ${LEG:-tk}", round(r["ms"], 4), round(r["roofline"]["frac"], 4), flush=True)
        break
else:
    print("$v ${LEG:-tk}: no result line", flush=True)
P
done
done
