#!/bin/bash
# wgrad_x3s on 16x16x32 (library) vs 32x32x16 (variant wxs32) vs wgrad_x3 (variant wspec0): parity, layers, step
O=gpurun_out/r4e51; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 > $O/pytest.txt
for r in 1 2; do
for v in new wspec0; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only wgrad 2>&1 | grep -v "^/opt" >> $O/layers.txt
  echo "[$v config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done; done
unset PNNP_LIB
cat $O/pytest.txt $O/bench_ab.txt; grep -E "==|total|conv1_2|conv2_1|conv9_1|conv4_2" $O/layers.txt
