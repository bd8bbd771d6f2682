#!/bin/bash
# Same-box A/B runs of a bench leg, round-robin (two passes), one line per run (ms, roofline fraction).  Replaces the thirteen
# one-shot tools/r05_ab*.sh scripts of round 5.
#
#   bash tools/ab.sh env <leg> <tag>:<ENV=V,ENV=V...> [<tag>:<...> ...]     environment switches (MM_* knobs read once per process)
#       bash tools/ab.sh env train_step split: f32:MM_KP_BWD_F32=1
#       bash tools/ab.sh env variants occ1:MM_KP128_OCC=1 occ2:MM_KP128_OCC=2
#   bash tools/ab.sh lib <leg> <variant> [<variant> ...]                    library builds (tools/build_variant.sh -> variants/libmm_native_<variant>.so;
#       bash tools/ab.sh lib tkl default ep0 ep2                            "default" = the shipped library), selected with MM_NATIVE_LIB
#   BENCH_ARGS="..." adds arguments to `python bench.py --only <leg> --lean --no-cpu-baseline`; KEYS="a.b c.d" prints those
#   fields of the leg's result instead of ms / roofline.frac.
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
mode=$1; leg=$2; shift 2
show() {  # show <tag> <log>
  KEYS="${KEYS:-}" python - "$1" "$2" <<'P'
import json, os, sys
tag, log = sys.argv[1], sys.argv[2]
for ln in reversed(open(log).read().splitlines()):
    if ln.startswith("{"):
        j = json.loads(ln); r = j.get("result", j)
        keys = os.environ.get("KEYS", "").split()
        if keys:
            def get(o, path):
                for p in path.split("."):
                    o = o[p]
                return round(o, 4) if isinstance(o, float) else o
            print(tag, {k: get(r, k) for k in keys}, flush=True)
        else:
            print(tag, round(r["ms"], 4), round(r["roofline"]["frac"], 4), flush=True)
        break
else:
    print(tag + ": no result line", flush=True)
P
}
for rnd in 1 2; do
  for spec in "$@"; do
    if [ "$mode" = lib ]; then
      tag=$spec; lib=$PWD/variants/libmm_native_$spec.so
      [ "$spec" = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
      MM_NATIVE_LIB=$lib python bench.py --only $leg --lean --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab/${leg}_$tag.log 2>&1
    else
      tag=${spec%%:*}; envs=${spec#*:}
      env $(echo "$envs" | tr ',' ' ') python bench.py --only $leg --lean --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab/${leg}_$tag.log 2>&1
    fi
    show "$tag $leg" gpurun_out/ab/${leg}_$tag.log
  done
done
