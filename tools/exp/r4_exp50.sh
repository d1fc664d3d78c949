#!/bin/bash
# conv_x3s: the favoured consumer wave of each SIMD idles N cycles behind every MFMA (X3S_YIELD)
O=gpurun_out/r4e50; mkdir -p $O
for v in new yield1 yield2 yield1n8 yield2n8; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -E "conv2_2|conv4_2|conv8_1|total" >> $O/layers.txt
done
unset PNNP_LIB
cat $O/layers.txt
