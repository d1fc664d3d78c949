#!/bin/bash
# usage: tools/ksize.sh file.hip [extra hipcc flags]  -> per kernel: VGPRs, spills, code bytes (device ELF symbol sizes)
F="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include "$@" --cuda-device-only -c "$F" -o /tmp/ksize.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|    VGPRs:|VGPRs Spill|ScratchSize" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - | sed -E 's/Function Name: _ZN12_GLOBAL__N_1[0-9]*//' | cut -c1-160
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/ksize.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/ksize.dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf -s --wide /tmp/ksize.dev.o | awk '$4=="FUNC"{print $3, $8}' | sed -E 's/_ZN12_GLOBAL__N_1[0-9]*//' | cut -c1-100
