#!/bin/bash
O=gpurun_out/r4e13; mkdir -p $O
V=tools/scratch/variants
PNNP_LIB=$V/libpnnp_adv.so timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py -x -q 2>&1 | grep -E "passed|failed|rror" > $O/pytest.txt
for r in 1 2 3; do
  for t in base adv; do
    echo "== $t" >> $O/layers.txt
    PNNP_LIB=$V/libpnnp_$t.so python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
echo done > $O/done.txt
