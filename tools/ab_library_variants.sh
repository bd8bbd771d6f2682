#!/bin/bash
# A/B of library builds on ONE box: LEG=<bench leg> VARIANTS="a b c" -> csrc/libmm_native_<variant>.so swapped in turn, two round-robin passes
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/stepc
L=matchmaker_amd/csrc
cp $L/libmm_native.so $L/libmm_native_keep.so
for rnd in 1 2; do
for v in ${VARIANTS:-prev base one s6 s10}; do
  cp $L/libmm_native_$v.so $L/libmm_native.so
  python bench.py --only ${LEG:-tk} --lean --no-cpu-baseline > gpurun_out/stepc/${LEG:-tk}_$v.log 2>&1
  python - <<P
import json
for ln in reversed(open("gpurun_out/stepc/${LEG:-tk}_$v.log").read().splitlines()):
    if ln.startswith("{"):
        j = json.loads(ln); r = j.get("result", j)
        print("$v ${LEG:-tk}", round(r["ms"], 4), round(r["roofline"]["frac"], 4))
        break
P
done
done
cp $L/libmm_native_keep.so $L/libmm_native.so
