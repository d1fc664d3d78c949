#!/bin/bash
O=gpurun_out/r4e35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_unet.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py tests/test_gpu_limits.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/pytest.txt
(export PNNP_LIB=tools/scratch/variants/libpnnp_gxsst.so GXS=1
for a in "convt 32 512 256" "convt 256 64 32" "s2 128 128 256 12"; do python tools/gx_stamps.py $a 2>&1 | grep -v "^/opt" >> $O/stamps.txt; done)
for v in new gxspec0; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/pointwise.txt
  timeout 300 python tools/pointwise_bench.py 2>&1 | grep -v "^/opt" >> $O/pointwise.txt
  echo "[$v config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
  echo "[$v config5] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline --arch resunet --noise noiseflow --batch 12 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done
unset PNNP_LIB
cat $O/pytest.txt $O/bench_ab.txt $O/stamps.txt
