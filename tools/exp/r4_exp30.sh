#!/bin/bash
O=gpurun_out/r4e30; mkdir -p $O
python tools/scratch/dbg_pool.py 2>&1 | grep -v "^/opt" > $O/dbg_pool.txt
for v in new direct0; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  python bench.py --no-cpu-baseline > $O/bench_$v.json 2>/dev/null
done
unset PNNP_LIB
cat $O/dbg_pool.txt
