#!/usr/bin/env python3
"""From a `rocprofv3 --kernel-trace` CSV of `bench.py --force-reducer` (or a multi-rank run): how much of the RCCL kernels' time
on the side stream runs concurrently with the backward pass's compute kernels on the main stream.

    python tools/overlap_from_trace.py <..._kernel_trace.csv> out.json

Writes {rccl_launches, rccl_ms, overlapped_ms, overlap_frac, per_bucket: [{start_us, dur_us, overlapped_us, with: [kernel, ...]}]}
(timestamps relative to the first RCCL launch of the analysed step: the LAST complete step in the trace)."""
import csv
import json
import re
import sys


def short(name):
    m = re.search(r'(\w+_kernel)', name)
    return m.group(1) if m else name.split('(')[0][-60:]


def main():
    src, out = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(src)))
    ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Stream_Id', '')) for r in rows]
    ks.sort()
    is_rccl = lambda n: 'nccl' in n.lower() or 'rccl' in n.lower()
    rccl = [k for k in ks if is_rccl(k[2])]
    comp = [k for k in ks if not is_rccl(k[2])]
    if not rccl:
        json.dump({'rccl_launches': 0}, open(out, 'w')); print('no RCCL kernels in the trace'); return
    # steps are delimited by adam_kernel launches: take the RCCL launches between the last two
    adams = [k[0] for k in comp if 'adam_kernel' in k[2]]
    lo, hi = (adams[-2], adams[-1]) if len(adams) >= 2 else (0, 1 << 62)
    step = [k for k in rccl if lo < k[0] < hi] or rccl
    t0 = step[0][0]
    per, tot, ov = [], 0, 0
    for s, e, n, q, st in step:
        o, names = 0, []
        for cs, ce, cn, cq, cst in comp:
            if ce <= s or cs >= e:
                continue
            o += min(e, ce) - max(s, cs); names.append(short(cn))
        o = min(o, e - s)
        per.append({'start_us': (s - t0) / 1e3, 'dur_us': (e - s) / 1e3, 'overlapped_us': o / 1e3, 'queue': q, 'stream': st,
                    'with': sorted(set(names))})
        tot += e - s; ov += o
    res = {'rccl_launches': len(step), 'rccl_ms': tot / 1e6, 'overlapped_ms': ov / 1e6, 'overlap_frac': ov / tot if tot else None,
           'compute_queues': sorted({k[3] for k in comp if lo < k[0] < hi}), 'rccl_queues': sorted({k[3] for k in step}),
           'per_bucket': per}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != 'per_bucket'}))


if __name__ == '__main__':
    main()
