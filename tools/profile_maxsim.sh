#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box from the repo root):
#   bash tools/profile_maxsim.sh <tag>      -> gpurun_out/prof_<tag>/{trace,pmc_*}/maxsim_results.db
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip tracing).
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/trace -o maxsim -- $B --steps 20 --warmup 3 > $O/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o maxsim -- $B --steps 5 --warmup 1 > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o maxsim -- $B --steps 5 --warmup 1 > $O/bench_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/pmc_tcc -o maxsim -- $B --steps 5 --warmup 1 > $O/bench_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o maxsim -- $B --steps 5 --warmup 1 > $O/bench_sq.log 2>&1
python tools/summarize_rocprof.py $O $O/summary.json "mm::" > /dev/null
find $O -name "*.db" -delete      # raw traces are tens of MB; the summary is what gets committed
tail -1 $O/bench_trace.log
