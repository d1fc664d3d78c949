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
  EXTRA=""
  case "$F" in conv_x3.hip|conv_x3s.hip|conv_h2s.hip|wgrad_h2s.hip|gemm_h2s.hip|wgrad_h2g.hip|wgrad_x3.hip|wgrad_x3s.hip|wgrad_x3g.hip|gemm_x3.hip|gemm_x3s.hip|conv_igemm.hip|wino.hip) EXTRA="-fno-slp-vectorize";; esac
  O=$V/obj_$TAG/${F%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I include $EXTRA $FL -c pnnp_amd/csrc/$F -o $O
  OBJS=$(echo "$OBJS" | grep -v "/${F%.hip}.o"); OBJS="$OBJS $O"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libpnnp_$TAG.so $OBJS
echo built $V/libpnnp_$TAG.so
