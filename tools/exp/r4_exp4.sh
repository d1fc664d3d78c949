#!/bin/bash
# round 4, experiment 4: bias preload + division-free tile stepping + pipelined epilogue round trips (new default) vs the old build (fm0);
# BN = 32 staging spread over two rows; the real SKIP_STORE bound
O=gpurun_out/r4e4; mkdir -p $O
V=tools/scratch/variants
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py tests/test_gpu_conv.py -x -q 2>&1 | tail -3 > $O/pytest_default.txt
PNNP_LIB=$V/libpnnp_spread.so timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py tests/test_gpu_resunet.py -x -q 2>&1 | tail -3 > $O/pytest_spread.txt
for shp in "512 32 32" "512 64 32" "256 64 64"; do
  echo "== stamps $shp" >> $O/stamps.txt
  PNNP_LIB=$V/libpnnp_stampsn.so python tools/x3_stamps.py $shp >> $O/stamps.txt 2>&1
done
for r in 1 2; do
  for t in fm0 new spread skipst; do
    echo "== $t" >> $O/layers.txt
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
unset PNNP_LIB
for r in 1 2; do
  for t in fm0 new spread; do
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    echo "[$t] $(python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  done
done
echo done > $O/done.txt
