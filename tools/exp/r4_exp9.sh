#!/bin/bash
# round 4, experiment 9: start-up skew between workgroups (de-phased epilogues)
O=gpurun_out/r4e9; mkdir -p $O
V=tools/scratch/variants
for r in 1 2; do
  for t in new skew16 skew32 skew64 skew127; do
    echo "== $t" >> $O/layers.txt
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
echo done > $O/done.txt
