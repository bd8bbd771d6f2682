#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r4final; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 > $O/gpu_tests.log; echo "gpu tests $(( $(date +%s)-t0 ))s"; tail -24 $O/gpu_tests.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log > $O/r04_bench_n1.json; python - <<'PY'
import json
j=json.load(open('gpurun_out/r4final/r04_bench_n1.json'))
print('headline', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('self_check'))
for k,v in j['extra'].items():
    if 'roofline' in v: print(k, round(v.get('ms',0),4), round(v['roofline']['frac'],4))
    elif k=='eval_batch':
        for n,s in v['shapes'].items(): print(' ', n, round(s['us_per_call_completed'],1), round(s['us_per_call_host_issue'],1), round(s['roofline']['frac'],3))
    else: print(k, str(v)[:200])
print('tk exact', j['extra']['tk'].get('exact_f32_mfma'))
print('vendor gemm (dot)', j['extra']['dot_topk'].get('vendor_gemm'))
print('vendor gemm (all pairs)', j['extra']['all_pairs'].get('vendor_gemm'))
print('tkl exact', j['extra']['tkl'].get('exact_f32_mfma'))
print('tkl 1024', j['extra']['tkl'].get('batch_1024_documents'))
PY
echo "total $(( $(date +%s)-t0 ))s"
