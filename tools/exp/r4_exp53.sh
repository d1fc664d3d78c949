#!/bin/bash
# conv_x3s: producers touch the pages of the tile's output three items before the epilogue stores to them (X3S_TLBWARM)
O=gpurun_out/r4e53; mkdir -p $O
(export PNNP_LIB=tools/scratch/variants/libpnnp_tlbwarmst.so
for a in "256 64 64 fwd" "256 64 64 dgrad" "64 256 256 fwd"; do python tools/x3s_stamps.py $a 2>&1 | grep -v "^/opt" >> $O/stamps.txt; done)
for r in 1 2; do
for v in new tlbwarm; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -E "conv2_2|conv4_2|conv8_1|total" >> $O/layers.txt
done; done
unset PNNP_LIB
cat $O/stamps.txt $O/layers.txt
