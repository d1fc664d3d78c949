#!/bin/bash
O=gpurun_out/r4e25; mkdir -p $O
V=tools/scratch/variants
PNNP_LIB=$V/libpnnp_wwide.so timeout 1500 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head > $O/pytest.txt
for r in 1 2 3; do
  for t in wwide new; do
    echo "== $t" >> $O/layers.txt
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    python tools/layer_bench.py --x3 --only wgrad --reps 7 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
echo done > $O/done.txt
