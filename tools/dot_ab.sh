#!/bin/bash
# Same-box A/B of the dot top-k call (DOT_AB_FLAGS=--raw: the bare native call, status ignored — by-removal builds): bash tools/dot_ab.sh <variant name> ...   ("" = the default library; round-robin, two passes)
for rnd in 1 2; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset MM_NATIVE_LIB; else export MM_NATIVE_LIB=$PWD/variants/libmm_native_$v.so; fi
    echo -n "pass $rnd $v: "; python tools/bench_dot_topk.py --steps 8 $DOT_AB_FLAGS 2>&1 | tail -1 | cut -c1-62
  done
done
