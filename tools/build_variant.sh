#!/bin/bash
# A/B library variant: tools/build_variant.sh <name> <translation unit> <-D flags...>
#   -> variants/libmm_native_<name>.so (git-ignored, outside the package: A/B libraries do not ship in matchmaker_amd/) = the current objects with <translation unit> recompiled under the flags.
# Select it for one process with MM_NATIVE_LIB=<path> (matchmaker_amd/_lib.py).  Build the default library first.
set -e
cd "$(dirname "$0")/../matchmaker_amd/csrc"
name=$1; tu=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c -o build/${tu}_${name}.o ${tu}.hip
objs=""
for o in common maxsim maxsim_pair kernel_pool kernel_pool128 kernel_pool_bwd kernel_pool_bwd_split tkl tkl_stage1_rows tkl_stage1_ksplit tkl_bwd dot_topk; do
  if [ "$o" = "$tu" ]; then objs="$objs build/${tu}_${name}.o"; else objs="$objs build/$o.o"; fi
done
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libmm_native_${name}.so $objs
echo "$(cd ../../variants && pwd)/libmm_native_${name}.so"
