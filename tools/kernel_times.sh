#!/bin/bash
# Per-kernel average durations of one command (rocprofv3 --kernel-trace --stats), printed as text.
#   bash tools/kernel_times.sh <tag> <command...>     (run on the GPU box from the repo root)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kt_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $O/trace -o kt -- "$@" > $O/cmd.log 2>&1
python tools/summarize_rocprof.py $O $O/summary.json "" > /dev/null
python - "$O/summary.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k in j["kernel_trace"]:
    if k["pct"] > 0.3:
        print(f'{k["avg_us"]:10.1f} us x{k["calls"]:<5d} {k["pct"]:5.1f}%  {k["name"][:110]}')
PY
find $O -name "*.db" -delete
tail -2 $O/cmd.log
