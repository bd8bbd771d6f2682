cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3final
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r3final/gpu_tests_head.log; cat gpurun_out/r3final/gpu_tests_head.log
bash tools/profile_legs.sh r03c "published_checkpoint eval_batch" "" 2>&1 | grep "^==" | cut -c1-400
