#!/bin/bash
# round 5: per-leg rocprofv3 evidence (kernel traces of bench.py's own legs + PMC passes), tools/profile_legs.sh
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_legs.sh r05 "headline tk tkl maxsim_fp32 variants train_step ragged_aggregate dot_topk" "headline tkl variants maxsim_fp32" > gpurun_out/legs_r05.log 2>&1
tail -30 gpurun_out/legs_r05.log
ls gpurun_out/legs_r05/*.json
