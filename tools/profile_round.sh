#!/bin/bash
# All rocprofv3 evidence of a round in one go (run on the GPU box from the repo root):
#   bash tools/profile_round.sh <tag> ["workloads"]   -> gpurun_out/prof_<tag>_{maxsim,dropin,tk,tkl,tklragged,dot,allpairs,published}/summary.json
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip tracing).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
prof() {   # prof <name> <command...>
  local W=$1; shift
  local O=$R/gpurun_out/prof_${TAG}_$W; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/trace -o $W -- "$@" > $O/bench_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o $W -- "$@" > $O/bench_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o $W -- "$@" > $O/bench_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace -d $O/pmc_sq -o $W -- "$@" > $O/bench_sq.log 2>&1
  if [ "$W" = maxsim ]; then
    python tools/summarize_rocprof.py $O $O/summary.json "mm::" > /dev/null
  else
    MM_PROF_COMMAND="$*" python tools/summarize_rocprof.py $O $O/summary.json "mm::" > /dev/null
  fi
  find $O -name "*.db" -delete      # raw traces are tens of MB; the summary is what gets committed
  echo "== $W"; tail -1 $O/bench_trace.log | cut -c1-400
}
WL=${2:-"maxsim dropin tk tkl tklragged dot"}
for W in $WL; do
  case $W in
    maxsim) prof maxsim python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2;;
    dropin) NQ=256 prof dropin python tools/bench_maxsim_variants.py dropin;;
    tk) prof tk python tools/bench_kernel_pool.py --full --queries 64 --steps 5;;
    tkl) prof tkl python tools/bench_tkl.py --full --steps 5;;
    tklragged) prof tklragged python tools/bench_tkl.py --steps 5;;
    dot) prof dot python tools/bench_dot_topk.py --steps 2;;
    allpairs) prof allpairs python tools/bench_inbatch.py 1024 1024;;
    published) prof published python tools/bench_maxsim_variants.py published768;;
  esac
done
