#!/bin/bash
export PNNP_LIB=tools/scratch/variants/libpnnp_x3sst.so
O=gpurun_out/e40_x3s_stamps.txt; : > $O
for a in "64 256 256 fwd" "64 256 256 dgrad" "256 64 64 fwd" "256 64 64 dgrad" "512 32 32 fwd" "512 32 32 dgrad"; do
  python tools/x3s_stamps.py $a 2>&1 | grep -v "^/opt" >> $O
done
cat $O
