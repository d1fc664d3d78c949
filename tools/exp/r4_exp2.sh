#!/bin/bash
# round 4, experiment 2: baseline per-layer table; bounds for the 32-column kernel; priority alternation
O=gpurun_out/r4e2; mkdir -p $O
V=tools/scratch/variants
python tools/layer_bench.py --x3 --reps 5 > $O/layers_all.txt 2>&1
L32="--layers conv1_2,conv9_1,conv9_2,conv2_1"
L64="--layers conv2_2,conv3_2,conv4_2,conv5_2,conv7_1"
for r in 1 2; do
  for t in fm0 nobar skipst nobar_skipst prio1 prio2; do
    echo "== $t" >> $O/layers32.txt
    PNNP_LIB=$V/libpnnp_$t.so python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 $L32 2>/dev/null | grep -v "^layer" >> $O/layers32.txt
  done
  for t in fm0 prio1 prio2; do
    echo "== $t" >> $O/layers64.txt
    PNNP_LIB=$V/libpnnp_$t.so python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 $L64 2>/dev/null | grep -v "^layer" >> $O/layers64.txt
  done
done
echo "== stampsp1" >> $O/stamps.txt
PNNP_LIB=$V/libpnnp_stampsp1.so python tools/x3_stamps.py 64 256 256 >> $O/stamps.txt 2>&1
echo done > $O/done.txt
