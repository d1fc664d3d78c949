#!/bin/bash
# same-box comparison: round 3's tree (commit 3465271, built in tools/scratch/r3tree) vs HEAD, config 3 and config 5, interleaved
O=gpurun_out/r4e49; mkdir -p $O
for r in 1 2 3; do
  for t in r3 r4; do
    if [ $t = r3 ]; then D=tools/scratch/r3tree; else D=.; fi
    echo "[$t config3] $(cd $D && python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  done
done
for r in 1 2; do
  for t in r3 r4; do
    if [ $t = r3 ]; then D=tools/scratch/r3tree; else D=.; fi
    echo "[$t config5] $(cd $D && python bench.py --no-kernel-events --no-cpu-baseline --arch resunet --noise noiseflow --batch 12 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  done
done
echo done > $O/done.txt
