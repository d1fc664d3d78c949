#!/bin/bash
O=gpurun_out/r4e15; mkdir -p $O
V=tools/scratch/variants
timeout 1500 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py tests/test_gpu_conv.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|rror" > $O/pytest.txt
for v in stampsn stampsnd; do
for shp in "512 32 32" "256 64 64"; do
  echo "== $v $shp" >> $O/stamps.txt
  PNNP_LIB=$V/libpnnp_$v.so python tools/x3_stamps.py $shp 2>&1 | grep "per item" >> $O/stamps.txt
done; done
for r in 1 2 3; do
  for t in base nodefer32 new; do
    echo "== $t" >> $O/layers.txt
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
unset PNNP_LIB
for r in 1 2; do
  for t in base nodefer32 new; do
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    echo "[$t] $(python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  done
done
echo done > $O/done.txt
