#!/bin/bash
export PNNP_LIB=tools/scratch/variants/libpnnp_gxsst.so GXS=1
O=gpurun_out/e34_gxs_stamps.txt; : > $O
for a in "convt 32 512 256" "convt 128 128 64" "convt 256 64 32" "s2 128 128 256 12" "pw 64 512 256 12"; do
  python tools/gx_stamps.py $a 2>&1 | grep -v "^/opt" >> $O
done
cat $O
