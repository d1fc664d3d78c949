#!/bin/bash
O=gpurun_out/r4e12; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err
python bench.py --no-cpu-baseline --arch resunet --noise noiseflow --batch 12 > $O/bench5.json 2> $O/bench5.err
echo done > $O/done.txt
