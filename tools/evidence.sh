#!/bin/bash
# The rocprofv3 evidence of a round in one GPU call (run on the GPU box from the repo root):
#   bash tools/evidence.sh <rNN>
# -> gpurun_out/legs_<rNN>/*.json (tools/profile_legs.sh: every bench leg profiled as the process that prints it, counter passes for
#    the HBM-bound legs), the pooling-variant counters split by process (Conv-KNRM's multi launch and IDCM's ck-small sampler share
#    one kernel instantiation), the TK backward alone (kernel trace + three counter passes + phase clocks), all copied under
#    gpurun_out/profiles_<rNN>/ with the names profiles/README.md lists: copy that directory's content into profiles/.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/profiles_$R; mkdir -p $O
bash tools/profile_legs.sh $R "headline tk tkl maxsim_fp32 train_step ragged_aggregate eval_batch dot_topk" "headline tk tkl maxsim_fp32" > gpurun_out/legs_$R.log 2>&1
for L in headline tk tkl maxsim_fp32 train_step ragged_aggregate eval_batch dot_topk; do cp gpurun_out/legs_$R/$L.json $O/${R}_${L}_trace.json 2>/dev/null; done
cp gpurun_out/legs_$R/headline_pmc.json $O/${R}_maxsim_headline_pmc.json 2>/dev/null
for L in tk tkl maxsim_fp32 tklfull; do cp gpurun_out/legs_$R/${L}_pmc.json $O/${R}_${L}_pmc.json 2>/dev/null; done
# pooling variants: the leg as a whole (trace), then the counters per process subset
MM_BENCH_VARIANTS= bash tools/profile_legs.sh ${R}v "variants" "" >> gpurun_out/legs_$R.log 2>&1
cp gpurun_out/legs_${R}v/variants.json $O/${R}_variants_trace.json 2>/dev/null
SKIP_TKLFULL=1 MM_BENCH_VARIANTS=knrm,tk_sparse,idcm_sampler_ck,idcm_sampler_ck_small bash tools/profile_legs.sh ${R}va "" "variants" >> gpurun_out/legs_$R.log 2>&1
cp gpurun_out/legs_${R}va/variants_pmc.json $O/${R}_variants_pmc.json 2>/dev/null
SKIP_TKLFULL=1 MM_BENCH_VARIANTS=conv_knrm_3x3 bash tools/profile_legs.sh ${R}vc "" "variants" >> gpurun_out/legs_$R.log 2>&1
cp gpurun_out/legs_${R}vc/variants_pmc.json $O/${R}_conv_knrm_pmc.json 2>/dev/null
# the TK backward alone: 2,048 and 32,768 pairs with the forward's pooled sums (what the training step runs)
bash tools/prof_cmd.sh kpbwd python tools/bench_kp_bwd.py --child --pairs 2048,32768 > gpurun_out/kpbwd_$R.log 2>&1
cp gpurun_out/kt_kpbwd/summary.json $O/${R}_tk_bwd_trace.json 2>/dev/null
python - "$O/${R}_tk_bwd_pmc.json" <<'P'
import json, sys
out = {"command": "python tools/bench_kp_bwd.py --child --pairs 2048,32768 (tools/prof_cmd.sh: three counter passes)", "pmc": {}}
for t in "abc":
    try:
        j = json.load(open(f"gpurun_out/pmc_kpbwd_{t}/summary.json"))
    except OSError:
        continue
    for k, v in j.get("pmc", {}).items():
        out["pmc"].setdefault(k, {}).update(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
P
# the dot top-k call alone: three counter passes (matrix-pipe busy cycles, LDS, waits) + the in-kernel phase clocks of the one-set form
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
if [ -f variants/libmm_native_phases.so ]; then
  MM_NATIVE_LIB=variants/libmm_native_phases.so python tools/bench_kp_bwd.py --child --phases --pairs 2048,32768 2>&1 | grep PHASES > $O/${R}_tk_bwd_phases.txt
fi
ls -la $O
tail -5 gpurun_out/legs_$R.log
