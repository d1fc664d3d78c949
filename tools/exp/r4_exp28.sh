#!/bin/bash
# gemm_x3 cycle stamps on the layers of tools/pointwise_bench.py
export PNNP_LIB=tools/scratch/variants/libpnnp_gxst.so
O=gpurun_out/e28_gx_stamps.txt; : > $O
for a in "convt 32 512 256" "convt 128 128 64" "convt 256 64 32" "s2 512 32 64 12" "s2 128 128 256 12" "pw 64 512 256 12" "pw 512 64 32 12"; do
  python tools/gx_stamps.py $a 2>&1 | grep -v "^/opt" >> $O
done
cat $O
