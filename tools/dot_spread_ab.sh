#!/bin/bash
# dot top-k filter: LDS-DMA spread over the K loop — parity + A/B on one box
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/stepb
timeout 900 python -m pytest tests -q -m gpu -x -k "dot or topk or flat or retrieval" 2>&1 | tail -4
for v in 1 0 1 0; do
  MM_DOT_NO_SPREAD=$v python bench.py --only dot_topk --lean --no-cpu-baseline > gpurun_out/stepb/dot_nospread$v.log 2>&1
  python - <<P
import json
for ln in reversed(open("gpurun_out/stepb/dot_nospread$v.log").read().splitlines()):
    if ln.startswith("{"):
        j = json.loads(ln); r = j.get("result", j)
        print("no_spread=$v", round(r["ms"], 3), round(r["roofline"]["frac"], 4))
        break
P
done
