#!/bin/bash
# One rocprofv3 PMC pass (counters only with --kernel-trace, as the guide prescribes) + text summary.
#   bash tools/pmc_pass.sh <tag> "<counters...>" <command...>
set -u
TAG=$1; CNT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc $CNT --kernel-trace -d $O/pmc -o p -- "$@" > $O/cmd.log 2>&1
python tools/summarize_rocprof.py $O $O/summary.json "mm::" > /dev/null
python - "$O/summary.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in j["pmc"].items():
    print(k[:100])
    for c, d in sorted(v.items()):
        if isinstance(d, dict) and "avg_per_dispatch" in d:
            print(f'   {c:32s} {d["avg_per_dispatch"]:16.0f}   ({d["dispatches"]} dispatches, {d["avg_dispatch_ns"]/1e3:.1f} us)')
PY
find $O -name "*.db" -delete
