#!/bin/bash
# usage: tools/kres.sh file.hip   -> per-kernel VGPR/AGPR/scratch/occupancy/LDS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: _ZN12_GLOBAL__N_1[0-9]*//' | cut -c1-200
