#!/bin/bash
# rocprofv3 evidence for the brute-force top-k kernels (run on the GPU box from the repo root):
#   bash tools/profile_dot.sh <tag>  -> gpurun_out/prof_<tag>_dot/summary.json
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/prof_${TAG}_dot; mkdir -p $O
CMD="python tools/bench_dot_topk.py --steps 2"
rocprofv3 --kernel-trace --stats -d $O/trace -o dot -- $CMD > $O/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o dot -- $CMD > $O/bench_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o dot -- $CMD > $O/bench_sq.log 2>&1
MM_PROF_COMMAND="$CMD" python tools/summarize_rocprof.py $O $O/summary.json "mm::" > /dev/null
find $O -name "*.db" -delete
tail -1 $O/bench_trace.log
