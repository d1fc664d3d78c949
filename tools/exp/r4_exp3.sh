#!/bin/bash
# round 4, experiment 3: the 32-column kernel (BN = 32): where does a tile's time go?
O=gpurun_out/r4e3; mkdir -p $O
V=tools/scratch/variants
for shp in "512 32 32" "512 64 32" "256 64 64"; do
  echo "== stamps $shp" >> $O/stamps.txt
  PNNP_LIB=$V/libpnnp_stamps0.so python tools/x3_stamps.py $shp >> $O/stamps.txt 2>&1
done
L32="--layers conv1_2,conv9_1,conv2_1,conv2_2"
for r in 1 2; do
  for t in fm0 skipepi; do
    echo "== $t" >> $O/layers32.txt
    PNNP_LIB=$V/libpnnp_$t.so python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 $L32 2>/dev/null | grep -v "^layer" >> $O/layers32.txt
  done
done
echo done > $O/done.txt
