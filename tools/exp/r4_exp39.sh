#!/bin/bash
# conv_x3s: act' masks touched three rows ahead of the epilogue (library) vs not (variant nowarm)
O=gpurun_out/r4e39; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3 > $O/pytest.txt
for r in 1 2; do
for v in new nowarm; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only dgrad 2>&1 | grep -v "^/opt" >> $O/layers.txt
  echo "[$v config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done; done
unset PNNP_LIB
cat $O/pytest.txt $O/bench_ab.txt; grep -E "==|total" $O/layers.txt
