#!/bin/bash
# tools/kres.sh <file.hip> [filter]: per-kernel VGPR / spill / scratch report of one translation unit (hipcc -Rpass-analysis)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Rpass-analysis=kernel-resource-usage "$@" -c -o build/$(basename ${f%.hip}).o $f 2>&1 | grep -E "error|Function Name|TotalSGPRs|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - - - | sed -e 's/Function Name: //' | c++filt | awk -F'\t' '{print $1" | "$3" | "$4" | "$5" | "$6" | "$7" | "$8}'
