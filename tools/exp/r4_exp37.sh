#!/bin/bash
# producers at s_setprio 3 (variant pprio) vs the library
O=gpurun_out/r4e37; mkdir -p $O
for r in 1 2; do
for v in new pprio; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -E "total" >> $O/layers.txt
  python tools/pointwise_bench.py 2>&1 | grep -E "^total" >> $O/layers.txt
done; done
cat $O/layers.txt
