#!/bin/bash
# gemm_x3s with item PAIRS on the 256-pixel tiles (library) vs one item per barrier (variant si1 = -DGXS_PAIRS=0): parity, pointwise table, config 3 / 5
O=gpurun_out/r4e52; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_unet.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py tests/test_gpu_limits.py tests/test_gpu_eval_pipeline.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/pytest.txt
for r in 1 2; do
for v in new si1; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/pointwise.txt
  timeout 300 python tools/pointwise_bench.py 2>&1 | grep -v "^/opt" >> $O/pointwise.txt
  echo "[$v config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  echo "[$v config5] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline --arch resunet --noise noiseflow --batch 12 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done; done
unset PNNP_LIB
cat $O/pytest.txt $O/bench_ab.txt; grep -E "==|^total" $O/pointwise.txt
