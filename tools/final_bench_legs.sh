#!/bin/bash
# bench line + per-leg profiles (the second half of tools/final_evidence.sh, without the test-suite)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3final; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log > $O/r03_bench_n1.json
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3final/r03_bench_n1.json'))
print('headline', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('self_check', {}).get('ok'))
for k,v in j['extra'].items():
    if 'roofline' in v: print(k, round(v.get('ms',0),4), round(v['roofline']['frac'],4))
    elif k=='eval_batch':
        for n,s in v['shapes'].items(): print(' ', n, round(s['us_per_call_completed'],1), round(s['us_per_call_host_issue'],1), round(s['roofline']['frac'],3))
print('tkl exact', j['extra']['tkl'].get('exact_f32_mfma', {}).get('ms'), 'tkl 1024', j['extra']['tkl'].get('batch_1024_documents'))
PY
bash tools/profile_legs.sh r03 > gpurun_out/legs_r03.log 2>&1; grep "^==" gpurun_out/legs_r03.log | cut -c1-330
