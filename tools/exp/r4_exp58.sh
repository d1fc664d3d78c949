#!/bin/bash
# non-temporal hint on conv_x3s's mask requests (maskaux2) / halo loads (haloaux2) vs the library: config 3, three alternating runs
O=gpurun_out/r4e58; mkdir -p $O
for r in 1 2 3; do
for v in new maskaux2 haloaux2; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "[$v config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done; done
cat $O/bench_ab.txt
