#!/bin/bash
export PNNP_LIB=tools/scratch/variants/libpnnp_epioobst.so
O=gpurun_out/e44_epioob_stamps.txt; : > $O
for a in "256 64 64 fwd" "256 64 64 dgrad" "512 32 32 fwd"; do python tools/x3s_stamps.py $a 2>&1 | grep -v "^/opt" >> $O; done
cat $O
