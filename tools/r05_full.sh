#!/bin/bash
# round 5: the whole -m gpu suite as the driver runs it (-x), then the default bench run as the driver runs it
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/r05_gpu_tests.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - t0 ))s" >> gpurun_out/r05_gpu_tests.txt
tail -22 gpurun_out/r05_gpu_tests.txt
python -c "from __graft_entry__ import smoke; smoke()" 2>&1 | tail -2
t0=$(date +%s)
python bench.py > gpurun_out/r05_bench_stdout.txt 2> gpurun_out/r05_bench_stderr.txt
echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s; stdout bytes $(wc -c < gpurun_out/r05_bench_stdout.txt); last line bytes $(tail -1 gpurun_out/r05_bench_stdout.txt | wc -c)"
tail -1 gpurun_out/r05_bench_stdout.txt
