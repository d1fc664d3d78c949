#!/bin/bash
# conv_x3s: cache-policy bits on the epilogue's stores (FWD / BWD bodies): aux 1 = sc0, 2 = nt, 3 = sc0 + nt
O=gpurun_out/r4e55; mkdir -p $O
for r in 1 2; do
for v in new aux1 aux2 aux3; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -E "conv1_2|conv2_2|conv4_2|conv8_1|total" >> $O/layers.txt
done; done
unset PNNP_LIB
cat $O/layers.txt
