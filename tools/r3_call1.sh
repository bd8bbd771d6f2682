#!/bin/bash
# round-3 GPU call 1: parity of the new TKL data path + new rank tests + RCCL tests, then A/B timings
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c1; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 600 python -m pytest tests/test_tkl_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/t_tkl.log; echo "tkl tests rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/t_tkl.log
timeout 600 python -m pytest tests/test_kernel_pool_gpu.py tests/test_variants_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/t_kp.log; echo "kp tests $(( $(date +%s)-t0 ))s"; tail -3 $O/t_kp.log
timeout 900 python -m pytest tests/test_rank_order_gpu.py -q -m gpu -k "tk" -s 2>&1 | grep -E "rank parity|passed|failed|Error|assert" | cut -c1-900 > $O/t_rank.log; echo "rank tests $(( $(date +%s)-t0 ))s"; tail -12 $O/t_rank.log
timeout 600 python -m pytest tests/test_rccl_gpu.py -q -m gpu -s -rs 2>&1 | tail -12 | cut -c1-1200 > $O/t_rccl.log; echo "rccl tests $(( $(date +%s)-t0 ))s"; cat $O/t_rccl.log
for v in 0 1; do
  echo "== TKL pairsums=$v"
  MM_TKL_PAIRSUMS=$v timeout 300 python tools/bench_tkl.py --steps 10 2>&1 | tail -1
  MM_TKL_PAIRSUMS=$v timeout 300 python tools/bench_tkl.py --steps 10 --full 2>&1 | tail -1
done
echo "== TK"; timeout 300 python tools/bench_kernel_pool.py --full --queries 64 --steps 10 2>&1 | tail -1
timeout 300 python tools/bench_kernel_pool.py --queries 64 --steps 10 --qlen config1 2>&1 | tail -1
echo "== kernel times TKL (cos)"; timeout 400 bash tools/kernel_times.sh r3c1_tkl python tools/bench_tkl.py --steps 5 --full 2>&1 | tail -8
echo "== kernel times TKL ragged (cos)"; timeout 400 bash tools/kernel_times.sh r3c1_tklr python tools/bench_tkl.py --steps 5 2>&1 | tail -8
echo "== eval_batch"; timeout 400 python bench.py --only eval_batch --no-cpu-baseline 2>&1 | tail -1 | cut -c1-3000
echo "total $(( $(date +%s)-t0 ))s"
