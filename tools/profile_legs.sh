#!/bin/bash
# Profiles of bench.py's OWN legs (the judge's "every extra fraction reproducible from profiles/"):
#   bash tools/profile_legs.sh <tag> ["legs"] ["pmc legs"]
# per leg:  rocprofv3 --kernel-trace --stats -- python bench.py --only <leg> --no-cpu-baseline
#           -> gpurun_out/legs_<tag>/<leg>.json = {bench leg's own line, kernel trace of the same process, un-traced line}
# per pmc leg additionally FETCH_SIZE / WRITE_SIZE / SQ passes (own runs, --kernel-trace only) -> <leg>_pmc.json
set -u
TAG=${1:-r03}
LEGS=${2-"headline dropin_forward published_checkpoint maxsim_fp32 all_pairs tk tkl dot_topk eval_batch"}
PMCL=${3-"headline tk tkl dot_topk all_pairs"}   # "" = no counter passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/legs_$TAG; mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for L in $LEGS; do
  CMD="python bench.py --only $L --lean --no-cpu-baseline --steps 20 --warmup 3"
  $CMD > $O/${L}_plain.log 2>&1
  rm -rf $O/$L; mkdir -p $O/$L
  rocprofv3 --kernel-trace --stats -d $O/$L/trace -o $L -- $CMD > $O/${L}_traced.log 2>&1
  MM_PROF_COMMAND="$CMD" python tools/summarize_rocprof.py $O/$L $O/${L}_trace_summary.json "mm::" > /dev/null
  python tools/merge_leg_profile.py $L $O/${L}_plain.log $O/${L}_traced.log $O/${L}_trace_summary.json $O/$L.json
  find $O/$L -name "*.db" -delete
  echo "== $L: $(python -c "import json;j=json.load(open('$O/$L.json'));print(j.get('check'))")"
done
for L in $PMCL; do
  CMD="python bench.py --only $L --lean --no-cpu-baseline --steps 10 --warmup 2"
  rm -rf $O/pmc_$L; mkdir -p $O/pmc_$L
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_$L/pmc_fetch -o $L -- $CMD > $O/pmc_$L/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_$L/pmc_write -o $L -- $CMD > $O/pmc_$L/write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace -d $O/pmc_$L/pmc_sq -o $L -- $CMD > $O/pmc_$L/sq.log 2>&1
  if [ "$L" = headline ]; then   # bench.py matches this file by its workload keys (queries / cands / lengths)
    python tools/summarize_rocprof.py $O/pmc_$L $O/${L}_pmc.json "mm::" > /dev/null
  else
    MM_PROF_COMMAND="$CMD" python tools/summarize_rocprof.py $O/pmc_$L $O/${L}_pmc.json "mm::" > /dev/null
  fi
  find $O/pmc_$L -name "*.db" -delete
  echo "== pmc $L done"
done

# TKL on full 2,048-token documents (the bench leg runs config 3's own lengths)
if [ -n "$PMCL" ] && [ -z "${SKIP_TKLFULL:-}" ]; then
CMD="python tools/bench_tkl.py --full --steps 5"
rm -rf $O/pmc_tklfull; mkdir -p $O/pmc_tklfull
rocprofv3 --kernel-trace --stats -d $O/pmc_tklfull/trace -o t -- $CMD > $O/pmc_tklfull/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_tklfull/pmc_fetch -o t -- $CMD > $O/pmc_tklfull/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_tklfull/pmc_write -o t -- $CMD > $O/pmc_tklfull/write.log 2>&1
MM_PROF_COMMAND="$CMD" python tools/summarize_rocprof.py $O/pmc_tklfull $O/tklfull_pmc.json "mm::" > /dev/null
find $O/pmc_tklfull -name "*.db" -delete
echo "== tklfull done"
fi
