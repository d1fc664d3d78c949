#!/bin/bash
# usage: tools/build_variant.sh <tag> <file.hip> "<extra hipcc flags>" [<file2.hip> "<flags2>" ...]
# Builds tools/scratch/variants/libpnnp_<tag>.so = the current library with the named source files recompiled with extra flags.
# Run HERE (the .so files travel to the GPU box); on the box select one with PNNP_LIB=tools/scratch/variants/libpnnp_<tag>.so.
set -e
cd "$(dirname "$0")/.."
TAG="$1"; shift
python tools/build.py > /dev/null
V=tools/scratch/variants; mkdir -p $V/obj_$TAG
OBJS=$(ls pnnp_amd/csrc/_build/*.o)
while [ $# -gt 0 ]; do
  F="$1"; FL="$2"; shift 2
  # the file's own flags: tools/build.py's EXTRA table (ONE source: round 6 found this script adding -fno-slp-vectorize to a file the product built without it)
  EXTRA=$(python -c "import sys; sys.path.insert(0, 'tools'); import build; print(' '.join(build.EXTRA.get('$F', [])))")
  O=$V/obj_$TAG/${F%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I include $EXTRA $FL -c pnnp_amd/csrc/$F -o $O
  OBJS=$(echo "$OBJS" | grep -v "/${F%.hip}.o"); OBJS="$OBJS $O"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libpnnp_$TAG.so $OBJS
echo built $V/libpnnp_$TAG.so
