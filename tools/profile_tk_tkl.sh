#!/bin/bash
# rocprofv3 evidence for the TK / TKL kernels (run on the GPU box from the repo root):
#   bash tools/profile_tk_tkl.sh <tag> [tk tkl idcm tksparse ...]  -> gpurun_out/prof_<tag>_<workload>/summary.json
set -u
TAG=${1:-r01}
shift
WORKLOADS=${@:-tk tkl}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for W in $WORKLOADS; do
  O=$R/gpurun_out/prof_${TAG}_$W; mkdir -p $O
  case $W in
    tk) CMD="python tools/bench_kernel_pool.py --full --queries 64 --steps 5";;
    tksparse) CMD="python tools/bench_kernel_pool.py --full --queries 64 --steps 5 --gate";;
    idcm) CMD="python tools/bench_kernel_pool.py --shape 30,64,768 --clamp 1e-4 --queries 64 --cands 1000 --full --steps 5";;
    convknrm) CMD="python tools/bench_kernel_pool.py --shape 30,180,128 --queries 64 --cands 1000 --full --steps 5";;
    *) CMD="python tools/bench_tkl.py --full --steps 5";;
  esac
  rocprofv3 --kernel-trace --stats -d $O/trace -o $W -- $CMD > $O/bench_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o $W -- $CMD > $O/bench_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o $W -- $CMD > $O/bench_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o $W -- $CMD > $O/bench_sq.log 2>&1
  MM_PROF_COMMAND="$CMD" python tools/summarize_rocprof.py $O $O/summary.json "mm::" > /dev/null
  find $O -name "*.db" -delete      # raw traces are tens of MB; the summary is what gets committed
  tail -1 $O/bench_trace.log
done
