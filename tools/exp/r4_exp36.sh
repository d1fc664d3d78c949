#!/bin/bash
# conv_x3s: consumer waves of a SIMD alternate their priority every 1 / 3 MFMAs
O=gpurun_out/r4e36; mkdir -p $O
for r in 1 2; do
for v in new prio1 prio3; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -E "total" >> $O/layers.txt
done; done
cat $O/layers.txt
