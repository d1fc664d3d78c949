#!/bin/bash
# on the GPU box, from the repo root: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py -> profiles-style CSVs + traffic.json
# usage: bash tools/pmc_traffic.sh [commit]   (the commit is recorded in traffic.json next to the kernel-source hash)
mkdir -p /root/repo/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
cd /root/repo
F=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmc_w -name "*counter_collection.csv" | head -1)
python - "$F" "$W" <<'PY'
import csv, sys
# keep only the conv kernels' rows (the full files are tens of MB)
for src, dst in ((sys.argv[1], 'gpurun_out/pmc_fetch_size.csv'), (sys.argv[2], 'gpurun_out/pmc_write_size.csv')):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if any(k in r['Kernel_Name'] for k in ('wino_kernel', 'wino_wgrad_kernel', 'igemm_kernel<9', 'wgrad_kernel<9', 'igemm_x3_kernel', 'igemm_x3s_kernel', 'igemm_h2s_kernel', 'wgrad_h2s_kernel', 'wgrad_x3s_kernel'))]
    w = csv.DictWriter(open(dst, 'w', newline=''), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
python tools/traffic_from_pmc.py gpurun_out/pmc_fetch_size.csv gpurun_out/pmc_write_size.csv gpurun_out/traffic.json "$1"
