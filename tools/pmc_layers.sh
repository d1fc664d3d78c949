#!/bin/bash
# on the GPU box: PMC pass over tools/layer_bench.py  ->  gpurun_out/pmc_layers_<tag>.csv (per-dispatch counters of the conv kernels)
# usage: bash tools/pmc_layers.sh <tag> "<counters>" <layer_bench args...>
TAG=$1; CTRS=$2; shift 2
mkdir -p /root/repo/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcl
rocprofv3 --pmc $CTRS --kernel-trace -d /tmp/pmcl -o p --output-format csv -- python3 /root/repo/tools/layer_bench.py "$@" > /root/repo/gpurun_out/pmc_layers_$TAG.txt 2>&1
cd /root/repo
python - "$(find /tmp/pmcl -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_layers_$TAG.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Dispatch_Id'], r['Kernel_Name'][:60], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
    agg.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], 'w', newline=''))
w.writerow(['dispatch', 'kernel', 'grid', 'lds'] + names)
for k, v in agg.items():
    if 'igemm' in k[1] or 'wino' in k[1] or 'wgrad' in k[1] or 'gemm_x3' in k[1]:
        w.writerow(list(k) + [v.get(n, '') for n in names])
PY
tail -5 gpurun_out/pmc_layers_$TAG.txt
