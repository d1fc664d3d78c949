#!/bin/bash
# on the GPU box: PMC pass over any python tool  ->  gpurun_out/pmc_<tag>.csv (per-dispatch counters of the kernels whose name matches <filter>)
# usage: bash tools/pmc_prog.sh <tag> "<counters>" <kernel-name filter> tools/<script>.py [args...]
TAG=$1; CTRS=$2; FILT=$3; PROG=$4; shift 4
mkdir -p /root/repo/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcp
rocprofv3 --pmc $CTRS --kernel-trace -d /tmp/pmcp -o p --output-format csv -- python3 /root/repo/$PROG "$@" > /root/repo/gpurun_out/pmc_$TAG.txt 2>&1
cd /root/repo
python - "$(find /tmp/pmcp -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_$TAG.csv "$FILT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Dispatch_Id'], r['Kernel_Name'][:70], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
    agg.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], 'w', newline=''))
w.writerow(['dispatch', 'kernel', 'grid', 'lds'] + names)
for k, v in agg.items():
    if sys.argv[3] in k[1]:
        w.writerow(list(k) + [v.get(n, '') for n in names])
PY
tail -3 gpurun_out/pmc_$TAG.txt
