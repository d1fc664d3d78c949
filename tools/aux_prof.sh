#!/bin/bash
# on the GPU box, from the repo root: rocprofv3 kernel stats of tools/aux_bench.py joined with its byte counts
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/auxp
rocprofv3 --kernel-trace --stats -d /tmp/auxp -o aux --output-format csv -- python3 /root/repo/tools/aux_bench.py --json /root/repo/gpurun_out/aux_rows.json > /root/repo/gpurun_out/aux_host.txt 2>&1
cd /root/repo
CSV=$(find /tmp/auxp -name "*kernel_stats.csv" | head -1)
cp $CSV gpurun_out/aux_kernel_stats.csv
python tools/aux_join.py $CSV gpurun_out/aux_rows.json gpurun_out/aux_kernels.json
