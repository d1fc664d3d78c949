#!/bin/bash
# conv_x3s with the argument cache + bias in LDS: parity, stamps, layers, step (vs variant direct0 = round 3's kernel, for the box's level)
O=gpurun_out/r4e41; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_unet.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py tests/test_gpu_eval_pipeline.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3 > $O/pytest.txt
(export PNNP_LIB=tools/scratch/variants/libpnnp_x3sst.so
for a in "64 256 256 fwd" "256 64 64 fwd" "256 64 64 dgrad" "512 32 32 fwd" "512 32 32 dgrad"; do python tools/x3s_stamps.py $a 2>&1 | grep -v "^/opt" >> $O/stamps.txt; done)
python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -v "^/opt" > $O/layers.txt
for r in 1 2; do
echo "[new config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
echo "[direct0 config3] $(PNNP_LIB=tools/scratch/variants/libpnnp_direct0.so timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done
cat $O/pytest.txt $O/bench_ab.txt $O/stamps.txt; tail -3 $O/layers.txt
