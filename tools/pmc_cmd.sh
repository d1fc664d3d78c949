#!/bin/bash
# on the GPU box: one PMC pass over any python tool -> gpurun_out/pmc_<tag>.csv (per kernel: mean of each counter over its dispatches)
# usage: bash tools/pmc_cmd.sh <tag> "<counters>" <kernel substring> tools/<script>.py [args...]
TAG=$1; CTRS=$2; MATCH=$3; shift 3
mkdir -p /root/repo/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcc
rocprofv3 --pmc $CTRS --kernel-trace -d /tmp/pmcc -o p --output-format csv -- python3 "/root/repo/$1" "${@:2}" > /root/repo/gpurun_out/pmc_$TAG.txt 2>&1
cd /root/repo
python - "$(find /tmp/pmcc -name '*counter_collection.csv' | head -1)" "$MATCH" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if sys.argv[2] not in r['Kernel_Name']:
        continue
    k = (r['Kernel_Name'][:70], r.get('Grid_Size', ''))
    d = agg.setdefault(k, collections.defaultdict(lambda: [0.0, 0]))
    d[r['Counter_Name']][0] += float(r['Counter_Value']); d[r['Counter_Name']][1] += 1
for k, d in agg.items():
    print(k[0], 'grid', k[1], ' '.join(f'{n}={v[0]/v[1]:.4g}' for n, v in sorted(d.items())), 'n=%d' % max(v[1] for v in d.values()))
PY
