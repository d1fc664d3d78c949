#!/bin/bash
O=gpurun_out/r4e11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "bwd_weight or convt_wgrad" 2>&1 | grep -E "passed|failed|rror|assert|Mismatch|x3 " | head -30 > $O/pytest.txt
python tools/convt_wgrad_bench.py > $O/bench.txt 2>&1
echo done > $O/done.txt
