#!/bin/bash
# The dot top-k part of tools/evidence.sh alone: bash tools/evidence_dot.sh <rNN>  -> gpurun_out/profiles_<rNN>/<rNN>_dot_topk_{trace,pmc}.json, _phases.txt
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/profiles_$R; mkdir -p $O
bash tools/profile_legs.sh ${R}d "dot_topk" "" > gpurun_out/legs_${R}d.log 2>&1
cp gpurun_out/legs_${R}d/dot_topk.json $O/${R}_dot_topk_trace.json
bash tools/prof_cmd.sh dot python tools/bench_dot_topk.py --steps 2 > gpurun_out/dot_$R.log 2>&1
python - "$O/${R}_dot_topk_pmc.json" <<'P'
import json, sys
out = {"command": "python tools/bench_dot_topk.py --steps 2 (tools/prof_cmd.sh: three counter passes)", "pmc": {}}
for t in "abc":
    try:
        j = json.load(open(f"gpurun_out/pmc_dot_{t}/summary.json"))
    except OSError:
        continue
    for k, v in j.get("pmc", {}).items():
        if "dot_stream" in k or "topk_rows" in k or "sample_tau" in k:
            out["pmc"].setdefault(k, {}).update(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
P
MM_DOT_PROF=1 python tools/bench_dot_topk.py --steps 1 2>&1 | grep MM_DOT_PROF | tail -1 > $O/${R}_dot_topk_phases.txt
ls -la $O; cat $O/${R}_dot_topk_phases.txt; tail -3 gpurun_out/legs_${R}d.log
