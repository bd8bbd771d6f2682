#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
{
for v in default wpc3 wpc2; do
  lib=$PWD/variants/libmm_native_$v.so; [ "$v" = default ] && lib=$PWD/matchmaker_amd/csrc/libmm_native.so
  echo "== $v"; MM_NATIVE_LIB=$lib python tools/exp_tkl_two_streams.py 2>&1 | grep -v Warn | tail -5
done
} > gpurun_out/r05_ab6.txt 2>&1
cat gpurun_out/r05_ab6.txt
