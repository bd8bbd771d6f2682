#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
{
for r in 1 2; do for nb in 2 3; do MM_MAXSIM_NBUF=$nb python bench.py --only eval_batch --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']['shapes']
print('nbuf=$nb', {k:(round(v['us_per_call_completed'],2), round(v['us_per_call_device'],2), round(v['roofline']['frac'],3)) for k,v in r.items() if 'colbert' in k})"; done; done
for nb in 2 3; do MM_MAXSIM_NBUF=$nb python bench.py --only dropin_forward --lean --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())['result']; print('nbuf=$nb dropin_forward', round(r['ms'],4), round(r['roofline']['frac'],4))"; done
} > gpurun_out/r05_ab10.txt 2>&1
cat gpurun_out/r05_ab10.txt
