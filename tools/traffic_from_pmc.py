#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/traffic.json:
HBM bytes per launch of the dominant kernel classes (wino_kernel, igemm_kernel<9,...>, the weight-gradient kernels), corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads: x2;
both counters are in KiB)."""
import collections, csv, json, os, re, sys
fetch_csv, write_csv, out = sys.argv[1:4]
commit = sys.argv[4] if len(sys.argv) > 4 else None       # the commit the passes ran on (the GPU box has no .git: pass it in)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
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
for cls, pred in (('x3', lambda k: k.startswith('igemm_x3_kernel') or k.startswith('igemm_x3s_kernel')), ('h2', lambda k: k.startswith('igemm_h2s_kernel')),
                  ('wgrad_h2', lambda k: k.startswith('wgrad_h2s_kernel')), ('wgrad_x3', lambda k: k.startswith('wgrad_x3s_kernel')), ('igemm9', lambda k: k.startswith('igemm_kernel<9')), ('wgrad9', lambda k: k.startswith('wgrad_kernel<9')),
                  ('wino', lambda k: k.startswith('wino_kernel')), ('wino_wgrad', lambda k: k.startswith('wino_wgrad_kernel'))):
    n = sum(v[0] for k, v in f.items() if pred(k))
    if not n:
        continue
    fs = sum(v[1] for k, v in f.items() if pred(k)); ws = sum(v[1] for k, v in w.items() if pred(k))
    res[cls] = {'launches': n, 'fetch_kib_per_launch_raw': fs / n, 'write_kib_per_launch': ws / n,
                'hbm_bytes_per_launch': (2 * fs + ws) * 1024 / n}
# bench.py attaches these numbers to its `roofline.traffic` only while the kernel sources still hash to csrc_sha
res['csrc_sha'] = bench.csrc_sha()
res['commit'] = commit
res['command'] = 'python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
