#!/usr/bin/env python3
"""Join tools/aux_bench.py's rows with the rocprofv3 kernel stats of the same run: GPU time per operator call = total duration of the
matching kernels / invocations -> achieved algorithmic GB/s and fraction of the 8 TB/s HBM peak.
    python tools/aux_join.py kernel_stats.csv rows.json [out.json]"""
import csv, json, sys

stats = list(csv.DictReader(open(sys.argv[1])))
doc = json.load(open(sys.argv[2]))
out = []
for r in doc['rows']:
    if r['match'] == ['__skip__']:
        continue
    tot = sum(float(k['TotalDurationNs']) for k in stats if any(m in k['Name'].replace(' ', '') for m in r['match']))
    if tot == 0:
        continue
    us = tot / r['invocations'] / 1e3
    gbps = r['alg_MB'] * 1e6 / (us * 1e-6) / 1e9
    out.append(dict(kernel=r['kernel'], workload=r['workload'], alg_MB=round(r['alg_MB'], 1), gpu_us=round(us, 1), GBps=round(gbps),
                    frac_of_hbm_peak=round(gbps / doc['peak_GBps'], 3), host_us=round(r['host_us'], 1)))
    print(f"{r['kernel']:40s} {r['alg_MB']:9.1f} MB  gpu {us:8.1f} us  {gbps:7.0f} GB/s  {100*gbps/doc['peak_GBps']:5.1f}% of HBM peak   (host-timed {r['host_us']:.1f} us)")
if len(sys.argv) > 3:
    json.dump(dict(peak_GBps=doc['peak_GBps'], rows=out), open(sys.argv[3], 'w'), indent=1)
