#!/bin/bash
# Same-box A/B of per-kernel times (rocprofv3 --kernel-trace --stats), round-robin, two passes:
#   bash tools/ab_kernels.sh <kernel grep> <leg> <tag>:<ENV=V,...> [<tag>:<...> ...]     (environment switches; "lib=<path>" selects MM_NATIVE_LIB)
#   e.g. bash tools/ab_kernels.sh stage1 tkl rows: slices:MM_TKL_STAGE1_SLICES=1
pat=$1; leg=$2; shift 2
for rnd in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    (
      IFS=,; for kv in $envs; do [ -n "$kv" ] && { if [ "${kv%%=*}" = lib ]; then export MM_NATIVE_LIB=${kv#*=}; else export "$kv"; fi; }; done
      echo -n "pass $rnd $tag: "
      bash tools/kernel_times.sh ab_$tag python bench.py --only $leg --lean --no-cpu-baseline 2>&1 | grep -E "$pat" | awk '{printf "%s us  ", $1}'; echo
    )
  done
done
