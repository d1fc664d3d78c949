#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/traffic.json:
HBM bytes per launch of the dominant kernel classes (wino_kernel, igemm_kernel<9,...>, the weight-gradient kernels), corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads: x2;
both counters are in KiB)."""
import collections, csv, json, re, sys
fetch_csv, write_csv, out = sys.argv[1:4]
def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != name:
            continue
        m = re.search(r'(\w+_kernel(?:<[^>]*>)?)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name']
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    return agg
f = load(fetch_csv, 'FETCH_SIZE'); w = load(write_csv, 'WRITE_SIZE')
res = {}
for cls, pred in (('igemm9', lambda k: k.startswith('igemm_kernel<9')), ('wgrad9', lambda k: k.startswith('wgrad_kernel<9')),
                  ('wino', lambda k: k.startswith('wino_kernel')), ('wino_wgrad', lambda k: k.startswith('wino_wgrad_kernel'))):
    n = sum(v[0] for k, v in f.items() if pred(k))
    if not n:
        continue
    fs = sum(v[1] for k, v in f.items() if pred(k)); ws = sum(v[1] for k, v in w.items() if pred(k))
    res[cls] = {'launches': n, 'fetch_kib_per_launch_raw': fs / n, 'write_kib_per_launch': ws / n,
                'hbm_bytes_per_launch': (2 * fs + ws) * 1024 / n}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
