#!/bin/bash
# What the driver runs at round end, on one box (tools/round.sh <rNN>): the whole -m gpu suite (-x), smoke(), the default bench.
#   -> gpurun_out/<rNN>_gpu_tests.txt, gpurun_out/<rNN>_bench_stdout.txt (+ gpurun_out/bench_full.json written by bench.py)
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1700 python -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/${R}_gpu_tests.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - t0 ))s" >> gpurun_out/${R}_gpu_tests.txt
tail -22 gpurun_out/${R}_gpu_tests.txt
python -c "from __graft_entry__ import smoke; smoke()" 2>&1 | tail -2
t0=$(date +%s)
python bench.py > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench_stderr.txt
echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s; stdout bytes $(wc -c < gpurun_out/${R}_bench_stdout.txt); last line bytes $(tail -1 gpurun_out/${R}_bench_stdout.txt | wc -c)"
tail -1 gpurun_out/${R}_bench_stdout.txt
