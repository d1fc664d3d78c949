#!/bin/bash
# samples GPU clock / power while bench.py runs (read-only rocm-smi queries)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|Socket" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_watch.txt &
W=$!
python bench.py --no-kernel-events --steps 200 --warmup 5 2>/dev/null | cut -c1-160
kill $W 2>/dev/null
sort gpurun_out/power_watch.txt | uniq -c | sort -rn | head -12
