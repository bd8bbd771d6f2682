#!/bin/bash
# eval.py-sized calls: parity of the in-kernel-mask path, bench.py eval_batch with and without two wavefronts per pair
# (MM_MAXSIM_NO_WPP2=1; MM_MAXSIM_NO_INLINE_MASKS=1 restores the packing launch), host-path profile
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/stepa
timeout 600 python -m pytest tests/test_maxsim_gpu.py -q -m gpu -x -k "long_queries or longer_than_one_tile or pair_per_row" 2>&1 | tail -5
MM_MAXSIM_NO_WPP2=1 python bench.py --only eval_batch --lean --no-cpu-baseline > gpurun_out/stepa/eval_packed.log 2>&1
python bench.py --only eval_batch --lean --no-cpu-baseline > gpurun_out/stepa/eval_inline.log 2>&1
python - <<'P'
import json
for n in ("packed", "inline"):  # "packed" = MM_MAXSIM_NO_WPP2=1 in this run
    for ln in reversed(open(f"gpurun_out/stepa/eval_{n}.log").read().splitlines()):
        if ln.startswith("{"):
            j = json.loads(ln); r = j.get("result", j)
            for k, v in r["shapes"].items():
                print(n, k, round(v["us_per_call_device"], 1), round(v["us_per_call_completed"], 1), round(v["us_per_call_host_issue"], 1), round(v["roofline"]["frac"], 3))
            break
P
python tools/host_path_profile.py 3000 2>&1 | tee gpurun_out/stepa/host_profile.txt | cut -c1-200
