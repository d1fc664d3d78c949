#!/bin/bash
O=gpurun_out/r4e26; mkdir -p $O
V=tools/scratch/variants
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" > $O/pytest.txt
for r in 1 2 3; do
  for t in base new; do
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    echo "[$t] $(python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  done
done
echo done > $O/done.txt
