#!/bin/bash
O=gpurun_out/r4e45; mkdir -p $O
(export PNNP_LIB=tools/scratch/variants/libpnnp_skew32bst.so
for a in "256 64 64 fwd" "256 64 64 dgrad"; do python tools/x3s_stamps.py $a 2>&1 | grep -v "^/opt" >> $O/stamps.txt; done)
for v in new skew32a skew32b; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -v "^/opt" >> $O/layers.txt
done
unset PNNP_LIB
cat $O/stamps.txt; grep -E "==|conv2_2|conv4_2|conv8_1|total" $O/layers.txt
